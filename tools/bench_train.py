#!/usr/bin/env python
"""Training-step timing (BASELINE configs[2]/[3]; not the headline metric).
  python tools/bench_train.py [--crop 256] [--steps 10] [--model TecoGAN]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      tools/bench_train.py --crop 128          # BASELINE config 4: DDP over RCCL, one rank per GPU
One JSON line (rank 0): steps/s, HR frames/s (= world * n * 19 / step), per-step ms (MAX over ranks).
Under torch.distributed.run the gradients of G (and of D when it updates) are averaged over the
ranks through one flat-bucket all-reduce each, BatchNorm statistics are global (SyncBatchNorm
halves) and the adaptive-D decision uses one fused 2-float all-reduce; seeds are 0 + rank."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--crop', type=int, default=256)
    ap.add_argument('--tempo', type=int, default=10)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--log-every', type=int, default=0, help='wait for the scalars every N iterations (0: only at the end)')
    ap.add_argument('--model', default='TecoGAN')
    ap.add_argument('--force-d', action='store_true', help='update D every step (threshold = +inf)')
    ap.add_argument('--feature-crit', action='store_true',
                    help='add the VGG19 perceptual loss of the shipped TecoGAN yml (weight 0.2, layers '
                         '8/17/26/35; default-initialised VGG weights: the timing does not depend on them)')
    ap.add_argument('--fm-crit', action='store_true', help='add the discriminator feature-matching loss (CB)')
    a = ap.parse_args()
    from tecogan_pytorch_amd.models import define_model
    from tecogan_pytorch_amd.utils import dist_utils
    world = int(os.environ.get('WORLD_SIZE', '1'))
    opt = {
        'scale': 4, 'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1, 'is_train': True,
        'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5}, 'train': {'crop_size': a.crop}},
        'model': {'name': a.model,
                  'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10},
                  'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3}},
        'train': {'tempo_extent': a.tempo, 'ckpt_dir': '/tmp',
                  'generator': {'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'discriminator': {'update_policy': 'adaptive', 'update_threshold': 1e9 if a.force_d else 0.4,
                                    'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'},
                  'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
        'logger': {'decay': 0.99},
    }
    if a.model == 'FRVSR':
        del opt['train']['pingpong_crit'], opt['train']['gan_crit']
    if a.feature_crit:
        opt['train']['feature_crit'] = {'type': 'CosineSimilarity', 'weight': 0.2, 'reduction': 'mean',
                                        'feature_layers': [8, 17, 26, 35], 'init': 'default'}
    if a.fm_crit:
        opt['train']['feature_matching_crit'] = {'type': 'CB', 'weight': 1.0, 'reduction': 'mean'}
    if world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ):
        dist_utils.init_dist(opt, int(os.environ.get('LOCAL_RANK', '0')))
    rank = opt['rank']
    torch.manual_seed(0 + rank)                   # base_utils.py:46
    m = define_model(opt)
    if opt['dist']:                               # identical initial weights on every rank
        import torch.distributed as dist
        for net in (m.net_G, getattr(m, 'net_D', None)):
            if net is not None:
                for p in net.parameters():
                    dist.broadcast(p.data, 0)
                    from tecogan_pytorch_amd import ops as _ops
                    _ops.bump_version(p)
    gen = torch.Generator().manual_seed(1 + rank)
    # (resident in HBM like bench.py's training leg and like the LMDB front end's batches: data/__init__.py)
    data = [{'gt': torch.rand(a.batch, a.tempo, 3, a.crop + 8, a.crop + 8, generator=gen).cuda()}
            for _ in range(2)]
    for i in range(a.warmup):
        m.prepare_training_data(data[i % 2]); m.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nupd0 = getattr(m, 'cnt_upd_D', 0.0)
    for i in range(a.steps):
        m.prepare_training_data(data[i % 2]); m.train()
        if a.log_every and (i + 1) % a.log_every == 0:
            m.sync_log()                 # what reading the losses every `log_every` iterations costs (1: the reference's loop)
    m.sync_log()
    torch.cuda.synchronize()
    nupd = int(getattr(m, 'cnt_upd_D', 0.0) - nupd0)
    dt = (time.perf_counter() - t0) / a.steps
    dt = dist_utils.max_over_ranks(dt, device='cuda') if opt['dist'] else dt
    tt = 2 * a.tempo - 1 if a.model != 'FRVSR' else a.tempo
    if rank != 0:
        return
    print(json.dumps({'model': a.model, 'world_size': opt['world_size'], 'crop': a.crop, 'batch': a.batch, 'tempo_extent': a.tempo,
                      'feature_crit': a.feature_crit, 'fm_crit': a.fm_crit, 'ms_per_step': 1e3 * dt, 'steps_per_s': 1 / dt,
                      'hr_frames_per_s': opt['world_size'] * a.batch * tt / dt, 'd_updates': nupd, 'steps': a.steps,
                      'last_log': {k: float(v) for k, v in m.log_dict.items()},
                      'peak_mem_gb': torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == '__main__':
    main()
