"""Round 4: the criteria no shipped yml selects but the reference's define_criterion accepts -- `gan_crit:
LSGAN` (optim/losses.py:17-28) and `feature_crit` of type L1 / MSE / CB (vsrgan_model.py:226-241) -- in the
training oracle against the reference's own VSRGANModel.train() (tests/golden/make_golden_r4.py ->
r4_extra.npz).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import tecogan_oracle as O
from procedural_weights import generator_state_dict, discriminator_state_dict, vgg19_state_dict
from tests.test_oracle_feat_vs_golden import CROP, T, SCALE, WATCH_G, batch, digest, close

VARIANTS = {
    'lsgan': dict(gan_type='LSGAN'),
    'feat_l1': dict(feat_type='L1', feat_w=0.2),
    'feat_mse': dict(feat_type='MSE', feat_w=0.05),
    'feat_cb': dict(feat_type='CB', feat_w=0.2),
}


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_tecogan_lsgan_and_pointwise_feature_criteria_two_iterations(golden, tag):
    g = golden('r4_extra')
    v = dict(VARIANTS[tag])
    sdG = generator_state_dict(scale=SCALE, degradation='BD')
    sdD = discriminator_state_dict(spatial_size=CROP, scale=SCALE, degradation='BD')
    if 'feat_type' in v:
        v['sd_F'] = vgg19_state_dict()
    aG, aD, st = {}, {}, {}
    keys = list(g[f'{tag}_log_keys'])
    torch.set_num_threads(min(8, torch.get_num_threads()))
    for it in range(2):
        lr, gt = O.prepare_training_data(batch(500 + 10 * it), SCALE, 'BD')
        log, gG, _ = O.vsrgan_train_step(sdG, sdD, aG, aD, st, lr, gt, SCALE, 'BD', CROP, T, **v)
        ref = dict(zip(keys, g[f'{tag}_log{it}']))
        assert set(keys) == set(log), (keys, list(log))
        for k in keys:
            close(log[k], ref[k], 2e-4, 2e-6, f'{tag} it{it} {k}')
        if it == 0:
            for k in WATCH_G:
                close(digest(gG[k]), g[f'{tag}_gradG_' + k], 5e-3, 2e-7, 'gradG ' + k)
        for k in WATCH_G:
            close(digest(sdG[k]), g[f'{tag}_paramG{it}_' + k], 1e-5, 2e-3, f'paramG{it} ' + k)
