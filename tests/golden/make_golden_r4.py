#!/usr/bin/env python
"""Round-4 golden vectors, produced by the upstream reference on CPU (authoring container only):
VSRGANModel.train() for the criteria no shipped yml selects but define_criterion accepts
(codes/models/optim/__init__.py:5-35): `gan_crit: LSGAN` (losses.py:17-28) and `feature_crit` of type
L1 / MSE / CB instead of CosineSimilarity (vsrgan_model.py:226-241), two iterations each -- log dict,
watched generator gradient digests, generator parameter digests.  VGG19 carries the procedural weights
(make_golden_feat.py explains the torchvision stub).

Output: tests/golden/r4_extra.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402
from make_golden_feat import _VGG19  # noqa: E402
from make_golden_train import make_opt, train_batch, digest, WATCH_G, CROP, SCALE  # noqa: E402
from procedural_weights import generator_state_dict, discriminator_state_dict  # noqa: E402

WATCH_D = ['conv_in.0.weight', 'discriminator_block.block3.1.weight', 'dense.weight']

VARIANTS = {
    'lsgan': dict(gan_crit={'type': 'LSGAN', 'weight': 0.01, 'reduction': 'mean'}),
    'feat_l1': dict(feature_crit={'type': 'L1', 'weight': 0.2, 'reduction': 'mean', 'feature_layers': [8, 17, 26, 35]}),
    'feat_mse': dict(feature_crit={'type': 'MSE', 'weight': 0.05, 'reduction': 'mean', 'feature_layers': [8, 17, 26, 35]}),
    'feat_cb': dict(feature_crit={'type': 'CB', 'weight': 0.2, 'reduction': 'mean', 'feature_layers': [8, 17, 26, 35]}),
}


def main():
    _ref_import.import_reference()
    sys.modules['torchvision.models'].vgg19 = lambda pretrained=True: _VGG19()
    import models
    import logging
    logging.getLogger('base').setLevel(logging.ERROR)
    torch.set_num_threads(8)
    d = {}
    for tag, extra in VARIANTS.items():
        opt = make_opt('TecoGAN')
        opt['train'].update(extra)
        torch.manual_seed(0)
        m = models.define_model(opt)
        m.net_G.load_state_dict(generator_state_dict(scale=SCALE, degradation='BD'), strict=True)
        m.net_D.load_state_dict(discriminator_state_dict(spatial_size=CROP, scale=SCALE, degradation='BD'), strict=True)
        keys = None
        for it in range(2):
            m.prepare_training_data({'gt': train_batch(500 + 10 * it)})
            m.train()
            keys = list(m.log_dict.keys())
            d[f'{tag}_log{it}'] = np.array([m.log_dict[k] for k in keys])
            if it == 0:
                gG = {k: p.grad for k, p in m.net_G.named_parameters()}
                for k, v in digest({k: gG[k] for k in WATCH_G}).items():
                    d[f'{tag}_gradG_' + k] = v
                gD = {k: p.grad for k, p in m.net_D.named_parameters()}
                if gD[WATCH_D[0]] is not None:
                    for k, v in digest({k: gD[k] for k in WATCH_D}).items():
                        d[f'{tag}_gradD_' + k] = v
            for k, v in digest({k: dict(m.net_G.named_parameters())[k] for k in WATCH_G}).items():
                d[f'{tag}_paramG{it}_' + k] = v
        d[f'{tag}_log_keys'] = np.array(keys)
        print(tag, keys)
        print('  it0', d[f'{tag}_log0'])
        print('  it1', d[f'{tag}_log1'])
    path = os.path.join(HERE, 'r4_extra.npz')
    np.savez_compressed(path, **d)
    print('r4_extra.npz', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
