"""Perceptual (VGG19 cosine) and feature-matching losses of the training oracle against the
reference's own VSRGANModel.train() (tests/golden/make_golden_feat.py -> train_feat.npz).
CPU only."""
import numpy as np
import pytest
import torch

from oracle import tecogan_oracle as O
from procedural_weights import (generator_state_dict, discriminator_state_dict, vgg19_state_dict)
from procedural_weights import smooth_clip

CROP, T, N, SCALE = 32, 4, 2, 4
WATCH_G = ['fnet.encoder1.0.weight', 'fnet.decoder1.2.bias', 'fnet.flow.2.weight',
           'srnet.conv_in.0.weight', 'srnet.resblocks.4.conv.2.weight', 'srnet.conv_up.2.weight',
           'srnet.conv_out.bias']


def batch(seed):
    return torch.stack([smooth_clip(T, 3, CROP + 8, CROP + 8, seed=seed + i, shift=1.0)
                        for i in range(N)])


def digest(v):
    v = v.detach().double().reshape(-1)
    return np.array([v.norm().item(), v.sum().item(), v[0].item(), v[v.numel() // 2].item(),
                     v[-1].item()])


def close(a, b, rtol, atol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.all(np.abs(a - b) <= atol + rtol * np.abs(b)), (what, a, b)

VARIANTS = {
    'feat': dict(feat=True, fm=None),
    'featfm': dict(feat=True, fm=dict(kind='CB', weight=0.3, reduction='mean')),
    'fm_l1': dict(feat=False, fm=dict(kind='L1', weight=0.5, reduction='mean',
                                      layer_norm=[10.0, 12.0, 20.0, 80.0])),
}


def test_vgg_features_and_cosine_loss_vs_reference(golden):
    g = golden('train_feat')
    sdF = vgg19_state_dict()
    x = torch.from_numpy(g['op_x']).requires_grad_(True)
    y = torch.from_numpy(g['op_y'])
    fx = O.vgg19_features(sdF, x)
    fy = O.vgg19_features(sdF, y)
    for i, f in enumerate(fx):
        v = f.detach().double()
        close([v.norm().item(), v.sum().item(), v.max().item()], g[f'op_feat{i}_stats'], 1e-5, 1e-6,
              f'feat{i}')
    assert np.abs(fx[3].detach().numpy() - g['op_feat3']).max() <= 1e-5
    loss = sum(O.cosine_similarity_loss(a, b.detach()) for a, b in zip(fx, fy))
    loss.backward()
    close(loss.item(), g['op_loss'], 1e-6, 1e-7, 'cosine loss')
    assert np.abs(x.grad.numpy() - g['op_grad_x']).max() <= 1e-6 + 1e-4 * np.abs(g['op_grad_x']).max()


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_tecogan_feature_losses_two_iterations(golden, tag):
    g = golden('train_feat')
    v = VARIANTS[tag]
    sdG = generator_state_dict(scale=SCALE, degradation='BD')
    sdD = discriminator_state_dict(spatial_size=CROP, scale=SCALE, degradation='BD')
    sdF = vgg19_state_dict() if v['feat'] else None
    aG, aD, st = {}, {}, {}
    keys = list(g[f'{tag}_log_keys'])
    for it in range(2):
        lr, gt = O.prepare_training_data(batch(300 + 10 * it), SCALE, 'BD')
        log, gG, _ = O.vsrgan_train_step(sdG, sdD, aG, aD, st, lr, gt, SCALE, 'BD', CROP, T,
                                         sd_F=sdF, feat_w=0.2, fm=v['fm'])
        ref = dict(zip(keys, g[f'{tag}_log{it}']))
        assert set(keys) == set(log), (keys, list(log))
        for k in keys:
            close(log[k], ref[k], 2e-4, 2e-6, f'{tag} it{it} {k}')
        if it == 0:
            for k in WATCH_G:
                close(digest(gG[k]), g[f'{tag}_gradG_' + k], 5e-3, 2e-7, 'gradG ' + k)
        for k in WATCH_G:
            close(digest(sdG[k]), g[f'{tag}_paramG{it}_' + k], 1e-5, 2e-3, f'paramG{it} ' + k)
