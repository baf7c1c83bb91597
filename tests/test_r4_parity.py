"""GPU, round 4: `gan_crit: LSGAN` (optim/losses.py:17-28) and `feature_crit` of type L1 / MSE / CB
(vsrgan_model.py:226-241; the shipped ymls use CosineSimilarity) through the HIP training step against the
reference's own VSRGANModel.train() (tests/golden/make_golden_r4.py -> r4_extra.npz): log dict of two
iterations (same entries, same order), watched generator gradient digests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from procedural_weights import generator_state_dict, discriminator_state_dict, vgg19_state_dict
from tests.test_hip_train import make_opt, batch, digest, close_digest, WATCH_G, CROP, SCALE

VARIANTS = {
    'lsgan': dict(gan_crit={'type': 'LSGAN', 'weight': 0.01, 'reduction': 'mean'}),
    'feat_l1': dict(feature_crit={'type': 'L1', 'weight': 0.2, 'reduction': 'mean', 'feature_layers': [8, 17, 26, 35], 'init': 'default'}),
    'feat_mse': dict(feature_crit={'type': 'MSE', 'weight': 0.05, 'reduction': 'mean', 'feature_layers': [8, 17, 26, 35], 'init': 'default'}),
    'feat_cb': dict(feature_crit={'type': 'CB', 'weight': 0.2, 'reduction': 'mean', 'feature_layers': [8, 17, 26, 35], 'init': 'default'}),
}


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_tecogan_lsgan_and_pointwise_feature_criteria_two_iterations(golden, tag):
    from tecogan_pytorch_amd.models import define_model
    g = golden('r4_extra')
    opt = make_opt('TecoGAN')
    opt['train'].update(VARIANTS[tag])
    m = define_model(opt)
    m.net_G.load_state_dict(generator_state_dict(scale=SCALE, degradation='BD'), strict=True)
    m.net_D.load_state_dict(discriminator_state_dict(spatial_size=CROP, scale=SCALE, degradation='BD'), strict=True)
    if 'feature_crit' in VARIANTS[tag]:
        m.net_F.load_vgg19_state_dict(vgg19_state_dict())
    keys = list(g[f'{tag}_log_keys'])
    for it in range(2):
        m.prepare_training_data({'gt': batch(500 + 10 * it)})
        m.train()
        assert list(m.log_dict.keys()) == keys          # same entries, same order as the reference
        ref = dict(zip(keys, g[f'{tag}_log{it}']))
        for k in keys:
            post_update = k in ('l_gan_G', 'p_fake_G') or it > 0
            rtol, atol = (1e-2, 5e-4) if post_update else (5e-4, 2e-5)
            assert abs(m.log_dict[k] - ref[k]) <= rtol * abs(ref[k]) + atol, (tag, it, k, m.log_dict[k], ref[k])
        if it == 0:
            pg = dict(m.net_G.named_parameters())
            for k in WATCH_G:
                close_digest(digest(pg[k].grad), g[f'{tag}_gradG_' + k], 2e-2, 'gradG ' + k)


def test_unknown_criteria_are_refused():
    from tecogan_pytorch_amd.models import define_model
    opt = make_opt('TecoGAN')
    opt['train']['gan_crit'] = {'type': 'WGAN', 'weight': 1}
    with pytest.raises(ValueError):
        define_model(opt)
