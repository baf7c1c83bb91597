#!/usr/bin/env python
"""Golden vectors for the perceptual (VGG19 cosine) and feature-matching losses of
VSRGANModel.train(), produced by the upstream reference on CPU (authoring container only).

The reference builds `torchvision.models.vgg19(pretrained=True).features`; torchvision is not
installed and the ImageNet weights are not obtainable offline, so the import stub's
`torchvision.models.vgg19` returns the published configuration-E layer stack carrying the
PROCEDURAL weights of procedural_weights.vgg19_state_dict -- the same weights the tests load
into the oracle and the HIP path.  Everything else (loss arithmetic, weighting, the order of
the D passes, logging) is the reference's own code.

Output: tests/golden/train_feat.npz"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402
from make_golden_train import make_opt, train_batch, digest, WATCH_G, CROP, SCALE  # noqa: E402
from procedural_weights import (generator_state_dict, discriminator_state_dict, vgg19_state_dict,
                                VGG19_CFG)  # noqa: E402


class _VGG19(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in VGG19_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.load_state_dict(vgg19_state_dict(), strict=True)


def main():
    _ref_import.import_reference()
    sys.modules['torchvision.models'].vgg19 = lambda pretrained=True: _VGG19()
    import models
    import logging
    logging.getLogger('base').setLevel(logging.ERROR)
    torch.set_num_threads(8)
    d = {}
    variants = {
        'feat': dict(feature_crit={'type': 'CosineSimilarity', 'weight': 0.2, 'reduction': 'mean',
                                   'feature_layers': [8, 17, 26, 35]}),
        'featfm': dict(feature_crit={'type': 'CosineSimilarity', 'weight': 0.2, 'reduction': 'mean',
                                     'feature_layers': [8, 17, 26, 35]},
                       feature_matching_crit={'type': 'CB', 'weight': 0.3, 'reduction': 'mean'}),
        'fm_l1': dict(feature_matching_crit={'type': 'L1', 'weight': 0.5, 'reduction': 'mean',
                                             'layer_norm': [10.0, 12.0, 20.0, 80.0]}),
    }
    for tag, extra in variants.items():
        opt = make_opt('TecoGAN')
        opt['train'].update(extra)
        torch.manual_seed(0)
        m = models.define_model(opt)
        m.net_G.load_state_dict(generator_state_dict(scale=SCALE, degradation='BD'), strict=True)
        m.net_D.load_state_dict(discriminator_state_dict(spatial_size=CROP, scale=SCALE,
                                                         degradation='BD'), strict=True)
        keys = None
        for it in range(2):
            m.prepare_training_data({'gt': train_batch(300 + 10 * it)})
            m.train()
            keys = list(m.log_dict.keys())
            d[f'{tag}_log{it}'] = np.array([m.log_dict[k] for k in keys])
            if it == 0:
                gG = {k: p.grad for k, p in m.net_G.named_parameters()}
                for k, v in digest({k: gG[k] for k in WATCH_G}).items():
                    d[f'{tag}_gradG_' + k] = v
            for k, v in digest({k: dict(m.net_G.named_parameters())[k] for k in WATCH_G}).items():
                d[f'{tag}_paramG{it}_' + k] = v
        d[f'{tag}_log_keys'] = np.array(keys)
        print(tag, keys)
        print('  it0', d[f'{tag}_log0'])
        print('  it1', d[f'{tag}_log1'])

    # op-level vectors: VGG features + cosine loss value / gradient on a small batch
    from models.networks.vgg_nets import VGGFeatureExtractor
    from models.optim.losses import CosineSimilarityLoss
    net_F = VGGFeatureExtractor([8, 17, 26, 35])
    x = train_batch(400)[0, :2, :, :32, :32].clone().requires_grad_(True)     # (2,3,32,32)
    y = train_batch(410)[0, :2, :, :32, :32].clone()
    fx, fy = net_F(x), net_F(y)
    crit = CosineSimilarityLoss()
    loss = sum(crit(a, b.detach()) for a, b in zip(fx, fy))
    loss.backward()
    d['op_x'], d['op_y'] = x.detach().numpy(), y.numpy()
    for i, f in enumerate(fx):
        v = f.detach().double()
        d[f'op_feat{i}_stats'] = np.array([v.norm().item(), v.sum().item(), v.max().item()])
    d['op_feat3'] = fx[3].detach().numpy()
    d['op_loss'] = np.array(loss.item())
    d['op_grad_x'] = x.grad.numpy()

    path = os.path.join(HERE, 'train_feat.npz')
    np.savez_compressed(path, **d)
    print('train_feat.npz', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
