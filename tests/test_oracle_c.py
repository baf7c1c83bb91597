"""The plain-C restatement (oracle/tecogan_oracle.c) against the reference's
golden vectors -- a second oracle that shares no arithmetic library with the
torch-based one.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from procedural_weights import generator_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'oracle', '_build', 'liboracle_c.so')
FP = C.POINTER(C.c_float)


@pytest.fixture(scope='module')
def orc():
    if not os.path.isfile(SO):
        subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    return C.CDLL(SO)


def fp(a):
    return a.ctypes.data_as(FP)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def test_c_warp_s2d_upsample_quantise(orc, golden):
    g = golden('ops')
    x, fl = f32(g['warp_x']), f32(g['warp_flow_big'])
    y = np.empty_like(x)
    orc.orc_backward_warp(fp(x), fp(fl), fp(y), 2, 3, 17, 23)
    assert np.abs(y - g['warp_out_big']).max() <= 5e-6
    for s in (2, 4):
        xs = f32(g[f's2d{s}_x'])
        out = np.empty_like(g[f's2d{s}_out'])
        orc.orc_space_to_depth(fp(xs), fp(out), 2, 3, xs.shape[2], xs.shape[3], s)
        assert np.array_equal(out, g[f's2d{s}_out'])
        xu = f32(g['up_x'])
        for mode, key in ((1, f'bicubic{s}_out'), (2, f'bilinear{s}_out')):
            out = np.empty_like(g[key])
            orc.orc_upsample(fp(xu), fp(out), 6, 9, 13, s, mode, C.c_float(1.0))
            assert np.abs(out - g[key]).max() <= 2e-6
    q = f32(g['quant_x'])
    n = q.size // 3 * 3
    xq = f32(q[:n].reshape(3, 1, n // 3))
    out = np.empty((1, n // 3, 3), np.uint8)
    orc.orc_quantize_u8_hwc(fp(xq), out.ctypes.data_as(C.POINTER(C.c_uint8)), 3, 1, n // 3)
    assert np.array_equal(out, g['quant_out'][:n].reshape(3, 1, n // 3).transpose(1, 2, 0))
    pad = np.empty_like(g['reflect_out'])
    orc.orc_reflect_pad_br(fp(f32(g['up_x'])), fp(pad), 6, 9, 13, 6, 5)
    assert np.array_equal(pad, g['reflect_out'])


def layer_arrays(sd, nb, scale):
    keys = []
    for blk in ('encoder1', 'encoder2', 'encoder3', 'decoder1', 'decoder2', 'decoder3', 'flow'):
        keys += [f'fnet.{blk}.0', f'fnet.{blk}.2']
    keys.append('srnet.conv_in.0')
    for b in range(nb):
        keys += [f'srnet.resblocks.{b}.conv.0', f'srnet.resblocks.{b}.conv.2']
    keys += ['srnet.conv_up.0'] + (['srnet.conv_up.2'] if scale == 4 else [])
    keys.append('srnet.conv_out')
    ws = [f32(sd[k + '.weight'].numpy()) for k in keys]
    bs = [f32(sd[k + '.bias'].numpy()) for k in keys]
    return ws, bs


@pytest.mark.parametrize('deg,s', [('BD', 4), ('BI', 2)])
def test_c_frnet_step_vs_reference(orc, golden, deg, s):
    g = golden(f'gen_{deg}{s}')
    sd = generator_state_dict(scale=s, degradation=deg)
    ws, bs = layer_arrays(sd, 10, s)
    WT = (FP * len(ws))(*[fp(a) for a in ws])
    BS = (FP * len(bs))(*[fp(a) for a in bs])
    for hw in ('22x40', '21x37'):
        lc, lp, hp = (f32(g[f'step_{hw}_{k}']) for k in ('lr_curr', 'lr_prev', 'hr_prev'))
        ref = g[f'step_{hw}_out']
        out = np.empty_like(ref)
        h, w = lc.shape[2:]
        rc = orc.orc_frnet_step(WT, BS, 10, 64, s, 1 if deg == 'BD' else 2, fp(lc), fp(lp), fp(hp),
                                fp(out), 1, h, w)
        assert rc == 0
        assert np.abs(out - ref).max() <= 1e-4, (hw, np.abs(out - ref).max())
