"""Driver with the reference's three modes and CLI flags (codes/main.py,
codes/utils/base_utils.py:14-30): train | test | profile.

`train` reads the reference's LMDB training sets when `dataset.train.seq_dir` names one
(tecogan_pytorch_amd/data: the decoded frames live in HBM, batches are cut out by a HIP
kernel with the reference's augmentation); without it, and for `test` (PNG folders are out of
scope), clips come from a synthetic source that honours the loader's output contract
(unpaired_lmdb_dataset.py:89-93, paired_folder_dataset.py:57-63); any iterable of such dicts
can be passed to `train()` / `test()`.

  python -m tecogan_pytorch_amd.main --mode profile --lr_size 3x134x320 --test_speed
  python -m tecogan_pytorch_amd.main --mode train --opt my_train.yml --gpu_ids 0
  torchrun --nproc-per-node 8 -m tecogan_pytorch_amd.main --mode train ...   (DDP over RCCL)
"""
import argparse
import os
import random
import time

import numpy as np
import torch
import yaml

from .metrics.psnr import compute_psnr, compute_psnr_device
from .models import define_model
from .models.networks import define_generator
from .utils import dist_utils


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--exp_dir', type=str, default='.')
    p.add_argument('--mode', type=str, required=True, help='train|test|profile')
    p.add_argument('--opt', type=str, default=None, help='yaml config (reference schema)')
    p.add_argument('--gpu_ids', type=str, default='0')
    p.add_argument('--lr_size', type=str, default='3x256x256')
    p.add_argument('--test_speed', action='store_true')
    p.add_argument('--local_rank', default=-1, type=int)
    p.add_argument('--iters', type=int, default=None,
                   help='train iterations to run.  Default: with an LMDB training set, up to '
                        'train.total_iter of the yml (the reference\'s loop, main.py:60-66); with the '
                        'synthetic source 20')
    p.add_argument('--resume', type=int, default=0,
                   help='iteration to resume from: loads {G,D}_iter<k>.pth and state_iter<k>.pth from '
                        'train.ckpt_dir and continues at k + 1 (the reference leaves this as a TODO, '
                        'base_model.py:220-222)')
    return p.parse_args(argv)


def default_opt():
    """The keys the hot path reads, with the shipped TecoGAN 4xSR BD values
    (experiments_BD/TecoGAN/TecoGAN_VimeoTecoGAN_4xSR_2GPU/train.yml) minus the VGG loss."""
    return {
        'scale': 4, 'manual_seed': 0,
        'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5},
                    'train': {'crop_size': 128, 'batch_size_per_gpu': 2, 'tempo_extent': 10}},
        'model': {'name': 'TecoGAN',
                  'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10},
                  'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3}},
        'train': {'tempo_extent': 10, 'total_iter': 20,
                  'generator': {'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'discriminator': {'update_policy': 'adaptive', 'update_threshold': 0.4,
                                    'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'},
                  'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
        'test': {'padding_mode': 'reflect', 'num_pad_front': 5},
        'logger': {'log_freq': 1, 'decay': 0.99, 'ckpt_freq': 0},
    }


def setup(args):
    if args.opt:
        with open(os.path.join(args.exp_dir, args.opt)) as f:
            opt = yaml.load(f.read(), Loader=yaml.FullLoader)
    else:
        opt = default_opt()
    opt['is_train'] = args.mode == 'train'
    local_rank = int(os.environ.get('LOCAL_RANK', args.local_rank))
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        dist_utils.init_dist(opt, max(local_rank, 0))
    else:
        if not torch.cuda.is_available():
            raise RuntimeError('an MI355X is required: the path has no CPU fallback')
        torch.cuda.set_device(int(args.gpu_ids.split(',')[0]))
        opt.update({'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1})
    seed_everything(opt.get('manual_seed', 2021) + opt['rank'])          # base_utils.py:46
    if opt['is_train']:      # the reference's test.yml files have no `train` section
        opt['train'].setdefault('ckpt_dir', os.path.join(args.exp_dir, 'train', 'ckpt'))
    return opt


def seed_everything(seed):
    """setup_random_seed, base_utils.py:78-83: the LMDB data set draws its crop / flip / rotation /
    moving-first-frame geometry from Python's `random`, the degradation from numpy / torch."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def synthetic_train_batches(opt, n_iter, seed):
    """{'gt': n x t x 3 x (S+2b) x (S+2b) float32 in [0,1]} (BD; b = int(3 sigma))."""
    g = torch.Generator().manual_seed(seed)
    n = opt['dataset']['train'].get('batch_size_per_gpu', 2)
    t = opt['train']['tempo_extent']
    s = opt['dataset']['train']['crop_size']
    b = int(opt['dataset']['degradation'].get('sigma', 1.5) * 3.0)
    for _ in range(n_iter):
        yield {'gt': torch.rand(n, t, 3, s + 2 * b, s + 2 * b, generator=g)}


def resume(model, opt, it):
    """Weights of iteration `it` + optimiser moments, schedule position and adaptive-D counter.
    The learning rate comes from the restored schedule position; every other hyper-parameter
    from the CURRENT yml."""
    ck = opt['train']['ckpt_dir']
    model.load_network(model.net_G, os.path.join(ck, f'G_iter{it}.pth'))
    if hasattr(model, 'net_D'):
        model.load_network(model.net_D, os.path.join(ck, f'D_iter{it}.pth'))
    got = model.resume_training_state(os.path.join(ck, f'state_iter{it}.pth'))
    if got != it:
        raise ValueError(f'state_iter{it}.pth was written at iteration {got}')
    return it


def train(opt, batches, start_iter=0):
    """codes/main.py:14-129 call sequence (logging to stdout on rank 0).  start_iter > 0:
    continue a checkpointed run (iterations are numbered from start_iter + 1, so checkpoints
    are not overwritten and the schedules continue where they stopped)."""
    model = define_model(opt)
    if start_iter:
        resume(model, opt, start_iter)
    for it, data in enumerate(batches, start_iter + 1):
        model.prepare_training_data(data)
        model.train()
        model.update_running_log()
        model.update_learning_rate()
        if opt['rank'] == 0 and it % opt['logger'].get('log_freq', 100) == 0:
            print(model.get_format_msg(0, it), flush=True)
        ck = opt['logger'].get('ckpt_freq', 0)
        if ck and it % ck == 0:
            os.makedirs(opt['train']['ckpt_dir'], exist_ok=True)
            model.save(it)
            model.save_training_state(it)      # optimiser moments etc.: restartable run
    return model


def test(opt, sequences):
    """codes/main.py:132-207: sequences sharded round-robin over ranks, PSNR-Y per sequence.
    `sequences`: list of {'gt': thwc uint8, 'lr': thwc float32, 'seq_idx': str}."""
    model = define_model(opt)
    rank, world = dist_utils.get_dist_info()
    vals = [0.0] * len(sequences)
    for idx in dist_utils.shard_indices(len(sequences)):
        data = sequences[idx]
        model.prepare_inference_data(data)
        # output frames stay on the GPU; the GT clip goes up as raw uint8; the squared-error
        # sums come from the HIP kernel (metric_calculator.py:228-244 without the host trip)
        hr_seq = model.infer(device_output=True).contiguous()
        gt = data['gt'].to(hr_seq.device).contiguous()
        vals[idx] = float(np.mean(compute_psnr_device(gt, hr_seq)))      # (synchronises)
        model.net_G.check_faults()      # fail-safe of chained launches: the clip above is complete
    red = dist_utils.reduce_sum_to_master(vals, device=opt['device'] if opt['dist'] else 'cpu')
    if rank == 0:
        for d, v in zip(sequences, red.tolist()):
            print(f"{d['seq_idx']}: PSNR-Y {v:.3f} dB")
    return red


def profile(opt, lr_size, test_speed=False):
    """codes/main.py:210-264 protocol: FLOPs/params, then FPS of step() over 30 fresh random
    inputs with a device sync per frame."""
    device = torch.device('cuda')
    net_G = define_generator(opt).to(device)
    gflops, params = net_G.profile(lr_size, device)
    for k in gflops:
        print(f'{k}: {gflops[k]:.3f} GFLOPs, {params[k] / 1e6:.3f} M params')
    print(f'total: {sum(gflops.values()):.3f} GFLOPs, {sum(params.values()) / 1e6:.3f} M params')
    if not test_speed:
        return None
    net_G.eval()
    n_test, tot = 30, 0.0
    with torch.no_grad():
        for _ in range(n_test):
            dummy = net_G.generate_dummy_data(lr_size, device)
            torch.cuda.synchronize()
            t0 = time.time()
            net_G.step(*dummy)
            torch.cuda.synchronize()
            tot += time.time() - t0
    fps = n_test / tot
    print(f'Speed: {fps:.2f} FPS ({1e3 * tot / n_test:.3f} ms/frame)')
    return fps


def main(argv=None):
    args = parse_args(argv)
    opt = setup(args)
    if args.mode == 'train' and opt['dataset'].get('train', {}).get('seq_dir') and \
            os.path.exists(opt['dataset']['train']['seq_dir']):
        from .data import TrainSource
        src = TrainSource(opt)

        total = int(opt['train'].get('total_iter', 0))
        n_iter = args.iters if args.iters is not None else max(0, total - args.resume)
        per_epoch = max(1, len(src))
        # a resumed run continues in the epoch (and at the batch) iteration `resume` had reached:
        # the sampler order of an epoch is a function of (seed, epoch) alone
        ep0, skip = divmod(args.resume, per_epoch)
        if ep0 > 0:
            src.replay(ep0)      # the geometry draws of the finished epochs (random streams aligned with an uninterrupted run)

        def epochs():
            ep, left, sk = ep0, n_iter, skip
            while left > 0:
                for i, b in enumerate(src.epoch(ep, first_batch=sk)):
                    if left <= 0:
                        return
                    left -= 1
                    yield b
                ep, sk = ep + 1, 0
        train(opt, epochs(), start_iter=args.resume)
    elif args.mode == 'train':
        train(opt, synthetic_train_batches(opt, 20 if args.iters is None else args.iters,
                                           100 + opt['rank'] + 7919 * args.resume),
              start_iter=args.resume)
    elif args.mode == 'test':
        g = torch.Generator().manual_seed(7)
        seqs = []
        for i in range(4):
            lr = torch.rand(8, 32, 48, 3, generator=g)
            gt = (torch.rand(8, 32 * opt['scale'], 48 * opt['scale'], 3, generator=g) * 255).to(torch.uint8)
            seqs.append({'gt': gt, 'lr': lr, 'seq_idx': f'synthetic_{i:03d}'})
        test(opt, seqs)
    elif args.mode == 'profile':
        profile(opt, tuple(int(v) for v in args.lr_size.split('x')), args.test_speed)
    else:
        raise ValueError(f'Unrecognized mode: {args.mode} (train|test|profile)')


if __name__ == '__main__':
    main()
