#!/usr/bin/env python
"""Generate the committed golden fixtures by running the UPSTREAM REFERENCE.

Runs only in the authoring container (needs /root/reference; see
_ref_import.py).  Output: tests/golden/*.npz -- inputs and the reference's
outputs, data only.  Weights are NOT stored: both sides rebuild them with
procedural_weights.py.  Re-run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402
from procedural_weights import (generator_state_dict, smooth_clip)  # noqa: E402


def rs_uniform(seed, shape, lo=0.0, hi=1.0):
    """Frozen legacy numpy stream; tests regenerate big inputs from the seed."""
    return torch.from_numpy(
        np.random.RandomState(seed).uniform(lo, hi, size=shape).astype(np.float32))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


def stats(x):
    """Size-independent digest of a big tensor: mean, L2, abs-max and a fixed
    lattice of sampled values."""
    x = x.detach().double()
    flat = x.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 257).long()
    return dict(mean=flat.mean().item(), l2=flat.norm().item(),
                amax=flat.abs().max().item(),
                samples=flat[idx].float().numpy(), sample_idx=idx.numpy())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = _ref_import.import_reference()
    NU, DU, NETS = ref.net_utils, ref.data_utils, ref.nets

    # ------------------------------------------------------------------ ops
    d = {}
    # backward_warp: generic, large out-of-range flow, exact-integer flow,
    # zero flow, flows that land exactly on the border.
    x = rs_uniform(1, (2, 3, 17, 23))
    fl = rs_uniform(2, (2, 2, 17, 23), -6, 6)
    d['warp_x'], d['warp_flow'] = x, fl
    d['warp_out'] = NU.backward_warp(x, fl)
    fl_big = rs_uniform(3, (2, 2, 17, 23), -40, 40)
    d['warp_flow_big'] = fl_big
    d['warp_out_big'] = NU.backward_warp(x, fl_big)
    fl_int = torch.round(rs_uniform(4, (2, 2, 17, 23), -3, 3))
    d['warp_flow_int'] = fl_int
    d['warp_out_int'] = NU.backward_warp(x, fl_int)
    d['warp_out_zero'] = NU.backward_warp(x, torch.zeros_like(fl))
    x1 = rs_uniform(5, (1, 3, 8, 1 + 8))  # small W
    fl1 = rs_uniform(6, (1, 2, 8, 9), -2, 2)
    d['warp_x_small'], d['warp_flow_small'] = x1, fl1
    d['warp_out_small'] = NU.backward_warp(x1, fl1)
    # space_to_depth
    for s in (2, 4):
        xs = rs_uniform(10 + s, (2, 3, 3 * s, 5 * s))
        d[f's2d{s}_x'] = xs
        d[f's2d{s}_out'] = NU.space_to_depth(xs, s)
    # upsamplers
    xu = rs_uniform(20, (2, 3, 9, 13), -1, 1)
    d['up_x'] = xu
    for s in (2, 4):
        d[f'bicubic{s}_out'] = NU.BicubicUpsampler(s)(xu)
        d[f'bicubic{s}_kernels'] = NU.BicubicUpsampler(s).kernels
        d[f'bilinear{s}_out'] = NU.get_upsampling_func(s, 'BI')(xu)
    # reflect pad bottom/right (tecogan_nets.py:239-241)
    import torch.nn.functional as F
    d['reflect_out'] = F.pad(xu, (0, 5, 0, 6), 'reflect')
    # float32_to_uint8: ties at .5/255, negatives, > 1
    q = np.concatenate([
        (np.arange(0, 256, dtype=np.float64) + 0.5) / 255.0,
        np.array([-0.3, -1e-8, 0.0, 1.0, 1.0 + 1e-7, 1.7, 0.49999 / 255, 0.50001 / 255]),
        np.random.RandomState(7).uniform(-0.1, 1.1, 400)]).astype(np.float32)
    d['quant_x'] = q
    d['quant_out'] = DU.float32_to_uint8(q)
    # BD blur+decimate: train mode (valid) and test mode (reflect pad)
    gt = rs_uniform(30, (2, 3, 40, 44))
    kern = DU.create_kernel(1.5)
    d['bd_gt'] = gt
    d['bd_kernel'] = kern
    d['bd_out_train'] = DU.downsample_bd(gt, kern, 4, pad_data=False)
    d['bd_out_test'] = DU.downsample_bd(gt, kern, 4, pad_data=True)
    d['bd_out_test_s2'] = DU.downsample_bd(gt, kern, 2, pad_data=True)
    # rgb->ycbcr
    img = np.random.RandomState(8).randint(0, 256, (9, 11, 3)).astype(np.uint8)
    d['ycbcr_x'] = img
    d['ycbcr_out'] = DU.rgb_to_ycbcr(img)
    save('ops', **d)

    # ------------------------------------------------------------- networks
    for deg, s in (('BD', 4), ('BI', 2), ('BD', 2)):
        tag = f'{deg}{s}'
        g = NETS.FRNet(3, 3, 64, 10, deg, s).eval()
        sd = generator_state_dict(scale=s, degradation=deg)
        g.load_state_dict(sd, strict=True)
        d = {}
        with torch.no_grad():
            # FNet on two sizes (one not a multiple of 8)
            for (h, w) in ((22, 40), (16, 24)):
                clip = smooth_clip(2, 3, h, w, seed=h)
                d[f'fnet_{h}x{w}_x1'], d[f'fnet_{h}x{w}_x2'] = clip[1:2], clip[0:1]
                d[f'fnet_{h}x{w}_out'] = g.fnet(clip[1:2], clip[0:1])
            # SRNet alone
            lr = rs_uniform(40, (2, 3, 12, 20))
            tr = rs_uniform(41, (2, 3 * s * s, 12, 20))
            d['srnet_lr'], d['srnet_tran'] = lr, tr
            d['srnet_out'] = g.srnet(lr, tr)
            # step, non-multiple-of-8 in both H and W
            for (h, w) in ((22, 40), (21, 37)):
                clip = smooth_clip(2, 3, h, w, seed=3)
                hp = rs_uniform(42, (1, 3, s * h, s * w))
                d[f'step_{h}x{w}_lr_curr'] = clip[1:2]
                d[f'step_{h}x{w}_lr_prev'] = clip[0:1]
                d[f'step_{h}x{w}_hr_prev'] = hp
                d[f'step_{h}x{w}_out'] = g.step(clip[1:2], clip[0:1], hp)
            # infer_sequence: 7 frames, uint8 thwc
            clip = smooth_clip(7, 3, 22, 40, seed=5)
            d['infer_lr'] = clip
            d['infer_out_u8'] = g.infer_sequence(clip, 'cpu')
        # forward_sequence (training unroll): n=2, t=4, 16x16
        g.train()
        with torch.no_grad():
            lr_seq = smooth_clip(8, 3, 16, 16, seed=9).view(2, 4, 3, 16, 16)
            fs = g.forward_sequence(lr_seq)
        g.eval()
        d['fseq_lr'] = lr_seq
        for k, v in fs.items():
            d['fseq_' + k] = v
        save(f'gen_{tag}', **d)

    # ----------------------------------------------- profile() + full size
    d = {}
    for deg, s, (h, w), tag in (('BD', 4, (134, 320), 'A'), ('BI', 2, (268, 640), 'E')):
        g = NETS.FRNet(3, 3, 64, 10, deg, s).eval()
        sd = generator_state_dict(scale=s, degradation=deg)
        g.load_state_dict(sd, strict=True)
        gf, pr = g.profile((3, h, w), 'cpu')
        d[f'profile_{tag}_gflops'] = np.array([gf['FNet'], gf['SRNet']])
        d[f'profile_{tag}_params'] = np.array([pr['FNet'], pr['SRNet']])
        # full-size frame digest: inputs regenerated in the test from seeds
        # (RandomState 100/101/102 uniform [0,1)), as main.py:249-262 does
        # with torch.rand.
        lc = rs_uniform(100, (1, 3, h, w))
        lp = rs_uniform(101, (1, 3, h, w))
        hp = rs_uniform(102, (1, 3, s * h, s * w))
        with torch.no_grad():
            out = g.step(lc, lp, hp)
        st = stats(out)
        for k, v in st.items():
            d[f'full_{tag}_{k}'] = v
        # and a smooth (flow-like) pair, where the warp is well conditioned
        clip = smooth_clip(2, 3, h, w, seed=11)
        with torch.no_grad():
            hp2 = g.step(clip[0:1], torch.zeros_like(clip[0:1]),
                         torch.zeros(1, 3, s * h, s * w))
            out2 = g.step(clip[1:2], clip[0:1], hp2)
        st = stats(out2)
        for k, v in st.items():
            d[f'fullsmooth_{tag}_{k}'] = v
    save('fullsize', **d)


if __name__ == '__main__':
    main()
