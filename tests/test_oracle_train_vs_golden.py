"""Training oracle (oracle/tecogan_oracle.py: vsr_train_step / vsrgan_train_step,
discriminator, losses, Adam, BD data prep) against vectors from the reference's own
VSRModel.train() / VSRGANModel.train() (tests/golden/make_golden_train.py). CPU only."""
import numpy as np
import pytest
import torch

from oracle import tecogan_oracle as O
from procedural_weights import generator_state_dict, discriminator_state_dict, smooth_clip

CROP, T, N, SCALE = 32, 4, 2, 4
WATCH_G = ['fnet.encoder1.0.weight', 'fnet.decoder1.2.bias', 'fnet.flow.2.weight',
           'srnet.conv_in.0.weight', 'srnet.resblocks.4.conv.2.weight', 'srnet.conv_up.2.weight',
           'srnet.conv_out.bias']
WATCH_D = ['conv_in.0.weight', 'discriminator_block.block2.0.weight',
           'discriminator_block.block3.1.weight', 'discriminator_block.block4.1.bias',
           'dense.weight', 'dense.bias']


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


def test_prepare_training_data(golden):
    g = golden('train_small')
    lr, gt = O.prepare_training_data(batch(100), SCALE, 'BD')
    assert np.abs(lr.numpy() - g['frvsr_lr_data']).max() <= 1e-6
    assert np.array_equal(gt.numpy(), g['frvsr_gt_data'])


def test_frvsr_two_iterations(golden):
    g = golden('train_small')
    sd = generator_state_dict(scale=SCALE, degradation='BD')
    adam = {}
    for it in range(2):
        lr, gt = O.prepare_training_data(batch(100 + 10 * it), SCALE, 'BD')
        log, grads = O.vsr_train_step(sd, adam, lr, gt, SCALE, 'BD', lr=1e-4)
        close([log['l_pix_G'], log['l_warp_G']], g[f'frvsr_log{it}'], 2e-5, 1e-7, f'log{it}')
        if it == 0:
            for k in WATCH_G:
                close(digest(grads[k]), g['frvsr_grad_' + k], 2e-3, 1e-7, 'grad ' + k)
        for k in WATCH_G:
            # Adam's first steps are ~lr*sign(g): a near-zero gradient whose sign flips under
            # fp32 summation-order noise moves one weight by 2*lr, so the digest tolerance is
            # a few such flips, not fp32 round-off
            close(digest(sd[k]), g[f'frvsr_param{it}_' + k], 1e-5, 2e-3, f'param{it} ' + k)


@pytest.mark.parametrize('tag,thr', [('gan', 0.4), ('gan_noD', -1e9)])
def test_tecogan_two_iterations(golden, tag, thr):
    g = golden('train_small')
    sdG = generator_state_dict(scale=SCALE, degradation='BD')
    sdD = discriminator_state_dict(spatial_size=CROP, scale=SCALE, degradation='BD')
    aG, aD, st = {}, {}, {}
    keys = list(g[f'{tag}_log_keys'])
    for it in range(2):
        lr, gt = O.prepare_training_data(batch(200 + 10 * it), SCALE, 'BD')
        log, gG, gD = O.vsrgan_train_step(sdG, sdD, aG, aD, st, lr, gt, SCALE, 'BD', CROP, T,
                                          update_threshold=thr)
        ref = dict(zip(keys, g[f'{tag}_log{it}']))
        for k in keys:
            close(log[k], ref[k], 2e-4, 2e-6, f'{tag} it{it} {k}')
        if it == 0:
            for k in WATCH_G:
                close(digest(gG[k]), g[f'{tag}_gradG_' + k], 5e-3, 2e-7, 'gradG ' + k)
            if thr > 0:
                for k in WATCH_D:
                    close(digest(gD[k]), g[f'{tag}_gradD_' + k], 5e-3, 2e-7, 'gradD ' + k)
        for k in WATCH_D:
            close(digest(sdD[k]), g[f'{tag}_paramD{it}_' + k], 1e-5, 2e-3, f'paramD{it} ' + k)
        close(sdD['discriminator_block.block1.1.running_mean'].numpy(), g[f'{tag}_bn{it}_rm'],
              1e-4, 1e-6, 'bn rm')
        close(sdD['discriminator_block.block4.1.running_var'].numpy(), g[f'{tag}_bn{it}_rv'],
              1e-4, 1e-6, 'bn rv')
        # BN statistics advance 3x per iteration even when D is not updated
        assert int(sdD['discriminator_block.block1.1.num_batches_tracked']) == int(g[f'{tag}_bn{it}_nbt']) == 3 * (it + 1)
