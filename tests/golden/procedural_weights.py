"""Deterministic, torch-RNG-independent weights for fixtures and parity tests.

A 10 MB weight blob is not committed; instead every tensor of the generator /
discriminator state dict is produced from numpy's frozen legacy RandomState
seeded by crc32(key name) -- identical in the authoring container (where the
reference is loaded with these weights to make the golden vectors) and on the
GPU box (where the HIP path and the oracle are loaded with them).

Magnitudes follow the fan-in rule PyTorch's default Conv2d init uses
(U(-1/sqrt(fan_in), 1/sqrt(fan_in))) times `gain`, so activations stay O(1)
through the 22-layer SRNet and the flow estimator produces non-trivial flow.
"""
import zlib

import numpy as np
import torch


def _tensor(name, shape, bound, seed):
    rs = np.random.RandomState((zlib.crc32(name.encode()) + seed) & 0x7FFFFFFF)
    return torch.from_numpy(rs.uniform(-bound, bound, size=shape).astype(np.float32))


def _conv(sd, key, co, ci, k, seed, gain=1.0, bias=True, transposed=False):
    fan_in = ci * k * k
    b = gain / np.sqrt(fan_in)
    shape = (ci, co, k, k) if transposed else (co, ci, k, k)
    sd[key + '.weight'] = _tensor(key + '.weight', shape, b * np.sqrt(3.0), seed)
    if bias:
        sd[key + '.bias'] = _tensor(key + '.bias', (co,), b, seed)


def generator_state_dict(in_nc=3, out_nc=3, nf=64, nb=10, scale=4,
                         degradation='BD', seed=0, flow_gain=1.0):
    """Keys exactly as the reference's FRNet.state_dict()."""
    sd = {}
    plan = [('encoder1', 2 * in_nc, 32), ('encoder2', 32, 64),
            ('encoder3', 64, 128), ('decoder1', 128, 256),
            ('decoder2', 256, 128), ('decoder3', 128, 64)]
    for name, ci, co in plan:
        _conv(sd, f'fnet.{name}.0', co, ci, 3, seed, gain=1.4)
        _conv(sd, f'fnet.{name}.2', co, co, 3, seed, gain=1.4)
    _conv(sd, 'fnet.flow.0', 32, 64, 3, seed, gain=1.4)
    _conv(sd, 'fnet.flow.2', 2, 32, 3, seed, gain=flow_gain)
    _conv(sd, 'srnet.conv_in.0', nf, (scale * scale + 1) * in_nc, 3, seed, gain=1.2)
    for b in range(nb):
        _conv(sd, f'srnet.resblocks.{b}.conv.0', nf, nf, 3, seed, gain=1.0)
        _conv(sd, f'srnet.resblocks.{b}.conv.2', nf, nf, 3, seed, gain=0.5)
    ups = [0, 2] if scale == 4 else [0]
    for u in ups:
        _conv(sd, f'srnet.conv_up.{u}', nf, nf, 3, seed, gain=1.2, transposed=True)
    _conv(sd, 'srnet.conv_out', out_nc, nf, 3, seed, gain=0.5)
    if degradation == 'BD':
        import sys, os
        here = os.path.dirname(os.path.abspath(__file__))
        root = os.path.dirname(os.path.dirname(here))
        if root not in sys.path:
            sys.path.insert(0, root)
        from oracle.tecogan_oracle import bicubic_kernels
        k = bicubic_kernels(scale)
        sd['upsample_func.kernels'] = k.clone()
        sd['srnet.upsample_func.kernels'] = k.clone()
    return sd


def discriminator_state_dict(in_nc=3, spatial_size=128, tempo_range=3,
                             scale=4, degradation='BD', seed=0):
    sd = {}
    _conv(sd, 'conv_in.0', 64, in_nc * tempo_range * 3, 3, seed, gain=1.4)
    chans = [(64, 64), (64, 64), (64, 128), (128, 256)]
    for i, (ci, co) in enumerate(chans, 1):
        p = f'discriminator_block.block{i}'
        _conv(sd, p + '.0', co, ci, 4, seed, gain=1.4, bias=False)
        sd[p + '.1.weight'] = 1.0 + _tensor(p + '.1.weight', (co,), 0.2, seed)
        sd[p + '.1.bias'] = _tensor(p + '.1.bias', (co,), 0.1, seed)
        sd[p + '.1.running_mean'] = torch.zeros(co)
        sd[p + '.1.running_var'] = torch.ones(co)
        sd[p + '.1.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    nfeat = 256 * (spatial_size // 16) ** 2
    sd['dense.weight'] = _tensor('dense.weight', (1, nfeat), np.sqrt(3.0 / nfeat), seed)
    sd['dense.bias'] = _tensor('dense.bias', (1,), 1.0 / np.sqrt(nfeat), seed)
    if degradation == 'BD':
        from oracle.tecogan_oracle import bicubic_kernels
        sd['upsample_func.kernels'] = bicubic_kernels(scale).clone()
    return sd


def smooth_clip(t, c, h, w, seed=0, shift=1.5):
    """A synthetic LR clip with real inter-frame motion (smooth blobs drifting
    by `shift` px/frame + mild noise) so FNet/warp are exercised on something
    flow-like rather than white noise.  Returns (t, c, h, w) fp32 in [0,1]."""
    rs = np.random.RandomState(1000 + seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32),
                         np.arange(w, dtype=np.float32), indexing='ij')
    frames = []
    nblob = 12
    cx = rs.uniform(0, w, nblob); cy = rs.uniform(0, h, nblob)
    rad = rs.uniform(2.0, max(3.0, min(h, w) / 4), nblob)
    col = rs.uniform(0.1, 1.0, (nblob, c))
    vx = rs.uniform(-shift, shift, nblob); vy = rs.uniform(-shift, shift, nblob)
    for i in range(t):
        img = np.full((c, h, w), 0.25, dtype=np.float32)
        for b in range(nblob):
            g = np.exp(-(((xx - cx[b] - vx[b] * i) ** 2 + (yy - cy[b] - vy[b] * i) ** 2)
                         / (2 * rad[b] ** 2)))
            img += (col[b][:, None, None] * g[None]).astype(np.float32) * 0.5
        img += rs.uniform(-0.02, 0.02, img.shape).astype(np.float32)
        frames.append(np.clip(img, 0, 1))
    return torch.from_numpy(np.stack(frames).astype(np.float32))


VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
             512, 512, 512, 512, 'M']


def vgg19_state_dict(seed=0):
    """`features.N.weight/bias` of torchvision's vgg19 layout (configuration E) with
    procedural He-scaled weights: the ImageNet weights cannot be obtained offline, and the
    parity of the perceptual-loss arithmetic does not depend on their values."""
    sd, cin, i = {}, 3, 0
    for v in VGG19_CFG:
        if v == 'M':
            i += 1
            continue
        # gain sqrt(2): keeps post-ReLU activations O(1) through 16 layers
        _conv(sd, f'features.{i}', v, cin, 3, seed, gain=float(np.sqrt(2.0)))
        sd[f'features.{i}.bias'] = sd[f'features.{i}.bias'] * 0.1 + 0.01
        cin = v
        i += 2
    return sd
