#!/usr/bin/env python
"""Golden vectors added in round 3, produced by the upstream reference (authoring container only):
  * VSRGANModel.train() with `degradation: BI` at 2x: the loader hands over {'gt', 'lr'}
    (base_model.py:51-53), the critic is sized by `gt_crop_size` (networks/__init__.py:25-28) and
    every up-sampling on the path is bilinear -- two iterations, log dicts / gradient and parameter
    digests / BatchNorm statistics;
  * SpatialDiscriminator (tecogan_nets.py:480-534) BACKWARD: gradients of sum(logit * r) w.r.t.
    the input clip and the parameters, with and without the bicubic condition, from autograd.
Output: tests/golden/r3_extra.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
from procedural_weights import generator_state_dict, discriminator_state_dict, smooth_clip, _conv  # noqa: E402

GT, T, N, SCALE = 32, 4, 2, 2
WATCH_G = ['fnet.encoder1.0.weight', 'fnet.flow.2.weight', 'srnet.conv_in.0.weight',
           'srnet.resblocks.4.conv.2.weight', 'srnet.conv_up.0.weight', 'srnet.conv_out.bias']
WATCH_D = ['conv_in.0.weight', 'discriminator_block.block2.0.weight', 'discriminator_block.block3.1.weight',
           'dense.weight', 'dense.bias']
SNET_WATCH = ['conv_in.0.weight', 'conv_in.0.bias', 'discriminator_block.block1.0.weight',
              'discriminator_block.block2.1.weight', 'discriminator_block.block4.1.bias', 'dense.weight']


def bi_opt(device='cpu'):
    return {
        'scale': SCALE, 'dist': False, 'device': device, 'rank': 0, 'world_size': 1, 'is_train': True,
        'dataset': {'degradation': {'type': 'BI'}, 'train': {'gt_crop_size': GT}},
        'model': {'name': 'TecoGAN',
                  'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10, 'load_path': None},
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


def bi_batch(seed):
    """paired loader contract (paired_lmdb_dataset.py): {'gt': n,t,3,S,S, 'lr': n,t,3,S/2,S/2}.  The
    LR frames are the exact 2x2 box average of the GT frames (reproducible bit for bit anywhere)."""
    gt = torch.stack([smooth_clip(T, 3, GT, GT, seed=seed + i, shift=1.0) for i in range(N)])
    lr = torch.nn.functional.avg_pool2d(gt.view(N * T, 3, GT, GT), SCALE).view(N, T, 3, GT // SCALE, GT // SCALE)
    return {'gt': gt, 'lr': lr}


def digest(v):
    v = v.detach().double().reshape(-1)
    return np.array([v.norm().item(), v.sum().item(), v[0].item(), v[v.numel() // 2].item(), v[-1].item()])


def snet_sd(in_ch, spatial):
    sd = discriminator_state_dict(spatial_size=spatial)
    sd.pop('upsample_func.kernels')
    sd.pop('conv_in.0.weight'); sd.pop('conv_in.0.bias')
    _conv(sd, 'conv_in.0', 64, in_ch, 3, 0, gain=1.4)
    return sd


def main():
    ref = _ref_import.import_reference()
    import models
    import logging
    logging.getLogger('base').setLevel(logging.ERROR)
    torch.set_num_threads(8)
    d = {}
    # ---------------- TecoGAN, BI 2x ------------------------------------------------------------
    torch.manual_seed(0)
    m = models.define_model(bi_opt())
    m.net_G.load_state_dict(generator_state_dict(scale=SCALE, degradation='BI'), strict=True)
    m.net_D.load_state_dict(discriminator_state_dict(spatial_size=GT, scale=SCALE, degradation='BI'), strict=True)
    keys = ['l_gan_D', 'p_real_D', 'p_fake_D', 'distance', 'n_upd_D', 'l_pix_G', 'l_warp_G', 'l_pp_G',
            'l_gan_G', 'p_fake_G']
    for it in range(2):
        m.prepare_training_data(bi_batch(300 + 10 * it))
        assert tuple(m.lr_data.shape) == (N, T, 3, GT // SCALE, GT // SCALE)
        m.train()
        d[f'bi_log{it}'] = np.array([m.log_dict[k] for k in keys])
        if it == 0:
            for k, p in m.net_G.named_parameters():
                if k in WATCH_G:
                    d['bi_gradG_' + k] = digest(p.grad)
            for k, p in m.net_D.named_parameters():
                if k in WATCH_D:
                    d['bi_gradD_' + k] = digest(p.grad)
        for k, p in m.net_D.named_parameters():
            if k in WATCH_D:
                d[f'bi_paramD{it}_' + k] = digest(p)
        bn = m.net_D.state_dict()
        d[f'bi_bn{it}_rm'] = bn['discriminator_block.block1.1.running_mean'].numpy().copy()
        d[f'bi_bn{it}_rv'] = bn['discriminator_block.block4.1.running_var'].numpy().copy()
    d['bi_log_keys'] = np.array(keys)
    # ---------------- SpatialDiscriminator backward ---------------------------------------------
    S, n, t = 32, 2, 3
    NETS = ref.nets
    r = torch.from_numpy(np.random.RandomState(11).uniform(-1, 1, (n * t, 1)).astype(np.float32))
    d['snet_r'] = r.numpy()
    for cond in (False, True):
        net = NETS.SpatialDiscriminator(3, S, cond).train()
        net.load_state_dict(snet_sd(6 if cond else 3, S), strict=True)
        data = torch.stack([smooth_clip(t, 3, S, S, seed=50 + i) for i in range(n)]).requires_grad_(True)
        bi = torch.stack([smooth_clip(t, 3, S, S, seed=60 + i) for i in range(n)])
        (logit, feats), _ = net(data, {'bi_data': bi})
        (logit * r).sum().backward()
        tag = f'snet_cond{int(cond)}'
        d[tag + '_ddata'] = data.grad.numpy().copy()
        for k, p in net.named_parameters():
            if k in SNET_WATCH:
                d[f'{tag}_grad_{k}'] = digest(p.grad)
    path = os.path.join(HERE, 'r3_extra.npz')
    np.savez_compressed(path, **d)
    print('r3_extra.npz', os.path.getsize(path) // 1024, 'KiB')
    for k in ('bi_log0', 'bi_log1'):
        print(k, d[k])


if __name__ == '__main__':
    main()
