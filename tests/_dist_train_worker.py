"""Worker of tests/test_dist_gpu.py: one rank of a data-parallel TecoGAN training run.

  python tests/_dist_train_worker.py RANK WORLD PORT OUTFILE BACKEND

Every rank constructs the model under its OWN seed (codes/utils/base_utils.py:46: seed +
rank), takes its shard of a fixed global batch, runs two VSRGANModel.train() iterations and
writes its weights / logs to OUTFILE.  BACKEND gloo lets two ranks share one GPU (RCCL
refuses duplicate devices): device tensors are then staged through the host by
utils/dist_utils -- the exchanged VALUES and every HIP kernel are the ones an RCCL run uses.
WORLD = 1 is the single-process reference on the whole batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

CROP, T, SCALE, GLOBAL_N = int(os.environ.get('TG_TEST_CROP', '32')), 4, 4, 2


def make_opt():
    return {
        'scale': SCALE, 'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1, 'is_train': True,
        'manual_seed': 0,
        'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5}, 'train': {'crop_size': CROP}},
        'model': {'name': 'TecoGAN',
                  'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10,
                                'load_path': None},
                  'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3, 'load_path': None}},
        'train': {'tempo_extent': T, 'ckpt_dir': '/tmp',
                  'generator': {'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'discriminator': {'update_policy': 'adaptive', 'update_threshold': 0.4,
                                    'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'},
                  'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
        'logger': {'decay': 0.99},
    }


def main():
    rank, world, port, outfile, backend = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    from procedural_weights import smooth_clip
    from tecogan_pytorch_amd.models import define_model
    from tecogan_pytorch_amd.utils import dist_utils
    opt = make_opt()
    if world > 1 and backend != 'nccl':
        # the ranks share cuda:0: chained launches assume the process owns the device;
        # this test is about the exchange, so the shared-GPU ranks run one launch per layer
        os.environ['TG_WINO_CHAIN'] = '0'
        from tecogan_pytorch_amd.models.networks.tecogan_nets import SRNet
        SRNet.chain_body = False
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank),
                          WORLD_SIZE=str(world), LOCAL_RANK=str(rank if backend == 'nccl' else 0))
        dist_utils.init_dist(opt, rank if backend == 'nccl' else 0, backend=backend, device='cuda')
    torch.manual_seed(opt['manual_seed'] + opt['rank'])          # base_utils.py:46
    m = define_model(opt)
    after_init = {('G.' + k): v.detach().cpu().clone() for k, v in m.net_G.state_dict().items()}
    after_init.update({('D.' + k): v.detach().cpu().clone() for k, v in m.net_D.state_dict().items()})
    logs = []
    per = GLOBAL_N // world
    for it in range(2):
        gt = torch.stack([smooth_clip(T, 3, CROP + 8, CROP + 8, seed=300 + 10 * it + i, shift=1.0)
                          for i in range(GLOBAL_N)])
        m.prepare_training_data({'gt': gt[rank * per:(rank + 1) * per]})
        m.train()
        local = dict(m.log_dict)
        m.update_running_log()                                   # reduce to rank 0 (mean)
        logs.append({'local': local, 'reduced': dict(m.log_dict)})
    final = {('G.' + k): v.detach().cpu().clone() for k, v in m.net_G.state_dict().items()}
    final.update({('D.' + k): v.detach().cpu().clone() for k, v in m.net_D.state_dict().items()})
    from tecogan_pytorch_amd.models import train_graph as TG
    torch.save({'after_init': after_init, 'final': final, 'logs': logs,
                'chained_launches': int(TG._ChainState.epoch), 'chain_disabled': bool(TG._ChainState.disabled)}, outfile)
    if world > 1:
        torch.distributed.barrier()
        dist_utils.destroy_c_comm()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
