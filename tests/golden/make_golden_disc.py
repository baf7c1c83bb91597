#!/usr/bin/env python
"""Golden vectors for the discriminator variants outside the shipped TecoGAN recipe:
SpatialDiscriminator (with / without condition) and the use_pp_crit=False flow
construction of SpatioTemporalDiscriminator -- outputs of the upstream reference
(authoring container only).  Output: tests/golden/disc_variants.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
from procedural_weights import generator_state_dict, discriminator_state_dict, smooth_clip  # noqa: E402


def spatial_sd(in_ch, spatial, seed=0):
    sd = discriminator_state_dict(spatial_size=spatial)
    sd.pop('upsample_func.kernels')
    from procedural_weights import _conv
    sd.pop('conv_in.0.weight'); sd.pop('conv_in.0.bias')
    _conv(sd, 'conv_in.0', 64, in_ch, 3, seed, gain=1.4)
    return sd


def main():
    ref = _ref_import.import_reference()
    NETS = ref.nets
    d = {}
    S, n, t = 32, 2, 3
    clips = torch.stack([smooth_clip(t, 3, S, S, seed=50 + i) for i in range(n)])
    bi = torch.stack([smooth_clip(t, 3, S, S, seed=60 + i) for i in range(n)])
    d['data'], d['bi'] = clips, bi
    for cond in (False, True):
        net = NETS.SpatialDiscriminator(3, S, cond).train()
        net.load_state_dict(spatial_sd(6 if cond else 3, S), strict=True)
        with torch.no_grad():
            (logit, feats), _ = net(clips, {'bi_data': bi})
        d[f'snet_cond{int(cond)}_logit'] = logit
        d[f'snet_cond{int(cond)}_feat3'] = feats[3]
    # STNet, use_pp_crit False: needs net_G.fnet
    G = NETS.FRNet(3, 3, 64, 10, 'BD', 4).eval()
    G.load_state_dict(generator_state_dict(), strict=True)
    D = NETS.SpatioTemporalDiscriminator(3, S, 3, 'BD', 4).train()
    D.load_state_dict(discriminator_state_dict(spatial_size=S), strict=True)
    lr = torch.stack([smooth_clip(6, 3, 8, 8, seed=70 + i) for i in range(n)])
    hr = torch.stack([smooth_clip(6, 3, S, S, seed=80 + i) for i in range(n)])
    bi6 = torch.stack([smooth_clip(6, 3, S, S, seed=90 + i) for i in range(n)])
    hr_flow = torch.from_numpy(np.random.RandomState(5).uniform(-2, 2, (n, 5, 2, S, S)).astype(np.float32))
    with torch.no_grad():
        (logit, feats), ret = D(hr, {'net_G': G, 'lr_data': lr, 'bi_data': bi6, 'hr_flow': hr_flow,
                                     'use_pp_crit': False, 'crop_border_ratio': 0.75})
    d['st_lr'], d['st_hr'], d['st_bi'], d['st_hr_flow'] = lr, hr, bi6, hr_flow
    d['st_nopp_logit'] = logit
    d['st_nopp_merge'] = ret['hr_flow_merge']
    np.savez_compressed(os.path.join(HERE, 'disc_variants.npz'), **{k: v.numpy() for k, v in d.items()})
    print('disc_variants.npz', os.path.getsize(os.path.join(HERE, 'disc_variants.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
