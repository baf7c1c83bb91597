#!/usr/bin/env python
"""Golden vectors for the TRAINING rows, produced by the upstream reference's
own VSRModel.train() / VSRGANModel.train() on CPU (authoring container only).
Output: tests/golden/train_small.npz (log dicts, gradient and parameter
digests).  Inputs/weights are regenerated procedurally by the tests."""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402
from procedural_weights import (generator_state_dict, discriminator_state_dict,
                                smooth_clip)  # noqa: E402

CROP, T, N, SCALE = 32, 4, 2, 4


def make_opt(model_name):
    return {
        'scale': SCALE, 'dist': False, 'device': 'cpu', 'rank': 0, 'world_size': 1,
        'is_train': True,
        'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5},
                    'train': {'crop_size': CROP}},
        'model': {'name': model_name,
                  'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10,
                                'load_path': None},
                  'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3,
                                    'load_path': None}},
        'train': {'tempo_extent': T, 'ckpt_dir': '/tmp',
                  'generator': {'lr': 1e-4 if model_name == 'FRVSR' else 5e-5,
                                'betas': [0.9, 0.999]},
                  'discriminator': {'update_policy': 'adaptive', 'update_threshold': 0.4,
                                    'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'},
                  'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
        'logger': {'decay': 0.99},
    }


def train_batch(seed):
    """loader contract (unpaired_lmdb_dataset.py:89-93): {'gt': n,t,3,S+8,S+8 in [0,1]}"""
    clips = [smooth_clip(T, 3, CROP + 8, CROP + 8, seed=seed + i, shift=1.0) for i in range(N)]
    return torch.stack(clips)


def digest(named):
    out = {}
    for k, v in named.items():
        v = v.detach().double().reshape(-1)
        out[k] = np.array([v.norm().item(), v.sum().item(), v[0].item(), v[v.numel() // 2].item(),
                           v[-1].item()])
    return out


WATCH_G = ['fnet.encoder1.0.weight', 'fnet.decoder1.2.bias', 'fnet.flow.2.weight',
           'srnet.conv_in.0.weight', 'srnet.resblocks.4.conv.2.weight', 'srnet.conv_up.2.weight',
           'srnet.conv_out.bias']
WATCH_D = ['conv_in.0.weight', 'discriminator_block.block2.0.weight',
           'discriminator_block.block3.1.weight', 'discriminator_block.block4.1.bias',
           'dense.weight', 'dense.bias']


def main():
    ref = _ref_import.import_reference()
    import models
    import logging
    logging.getLogger('base').setLevel(logging.ERROR)
    torch.set_num_threads(8)
    d = {}

    # ---------------- FRVSR (VSRModel.train) -----------------------------------
    opt = make_opt('FRVSR')
    del opt['train']['pingpong_crit'], opt['train']['gan_crit']
    torch.manual_seed(0)
    m = models.define_model(opt)
    m.net_G.load_state_dict(generator_state_dict(scale=SCALE, degradation='BD'), strict=True)
    for it in range(2):
        m.prepare_training_data({'gt': train_batch(100 + 10 * it)})
        if it == 0:
            d['frvsr_lr_data'] = m.lr_data.numpy().copy()
            d['frvsr_gt_data'] = m.gt_data.numpy().copy()
        m.train()
        d[f'frvsr_log{it}'] = np.array([m.log_dict['l_pix_G'], m.log_dict['l_warp_G']])
        if it == 0:
            g = {k: p.grad for k, p in m.net_G.named_parameters()}
            for k, v in digest({k: g[k] for k in WATCH_G}).items():
                d['frvsr_grad_' + k] = v
        for k, v in digest({k: dict(m.net_G.named_parameters())[k] for k in WATCH_G}).items():
            d[f'frvsr_param{it}_' + k] = v

    # ---------------- TecoGAN (VSRGANModel.train) ------------------------------
    for tag, thr in (('gan', 0.4), ('gan_noD', -1e9)):
        opt = make_opt('TecoGAN')
        opt['train']['discriminator']['update_threshold'] = thr
        torch.manual_seed(0)
        m = models.define_model(opt)
        m.net_G.load_state_dict(generator_state_dict(scale=SCALE, degradation='BD'), strict=True)
        m.net_D.load_state_dict(discriminator_state_dict(spatial_size=CROP, scale=SCALE,
                                                         degradation='BD'), strict=True)
        for it in range(2):
            m.prepare_training_data({'gt': train_batch(200 + 10 * it)})
            m.train()
            keys = ['l_gan_D', 'p_real_D', 'p_fake_D', 'distance', 'n_upd_D', 'l_pix_G',
                    'l_warp_G', 'l_pp_G', 'l_gan_G', 'p_fake_G']
            d[f'{tag}_log{it}'] = np.array([m.log_dict[k] for k in keys])
            if it == 0:
                gG = {k: p.grad for k, p in m.net_G.named_parameters()}
                for k, v in digest({k: gG[k] for k in WATCH_G}).items():
                    d[f'{tag}_gradG_' + k] = v
                if thr > 0:
                    # D grads were produced by loss_D.backward() (before the frozen pass)
                    gD = {k: p.grad for k, p in m.net_D.named_parameters()}
                    for k, v in digest({k: gD[k] for k in WATCH_D}).items():
                        d[f'{tag}_gradD_' + k] = v
            for k, v in digest({k: dict(m.net_D.named_parameters())[k] for k in WATCH_D}).items():
                d[f'{tag}_paramD{it}_' + k] = v
            bn = m.net_D.state_dict()
            d[f'{tag}_bn{it}_rm'] = bn['discriminator_block.block1.1.running_mean'].numpy().copy()
            d[f'{tag}_bn{it}_rv'] = bn['discriminator_block.block4.1.running_var'].numpy().copy()
            d[f'{tag}_bn{it}_nbt'] = np.array(int(bn['discriminator_block.block1.1.num_batches_tracked']))
        d[f'{tag}_log_keys'] = np.array(keys)

    path = os.path.join(HERE, 'train_small.npz')
    np.savez_compressed(path, **d)
    print('train_small.npz', os.path.getsize(path) // 1024, 'KiB')
    for k in ('frvsr_log0', 'frvsr_log1', 'gan_log0', 'gan_log1', 'gan_noD_log0', 'gan_noD_log1'):
        print(k, d[k])


if __name__ == '__main__':
    main()
