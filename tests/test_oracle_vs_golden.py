"""Pins oracle/tecogan_oracle.py against vectors produced by the upstream
reference itself (tests/golden/make_golden.py).  CPU only.

Tolerances: ops the oracle restates from formulas are held to a few fp32 ulp
of the value range; whole-network outputs to 2e-5 abs (outputs are O(1); the
only differences are summation order inside the restated bicubic/bilinear/warp
ops, amplified through <= 24 conv layers)."""
import numpy as np
import pytest
import torch

from oracle import tecogan_oracle as O
from procedural_weights import generator_state_dict

T = torch.from_numpy


def close(a, b, atol, what=''):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    err = np.abs(a.astype(np.float64) - np.asarray(b, np.float64)).max()
    assert err <= atol, f'{what}: max abs err {err:.3e} > {atol:.1e}'


def test_linspace_matches_torch():
    for n in (2, 3, 9, 23, 134, 320, 536, 1280):
        assert np.array_equal(O.linspace_m1_p1(n), torch.linspace(-1.0, 1.0, n).numpy())


@pytest.mark.parametrize('case', ['', '_big', '_int', '_zero'])
def test_warp(golden, case):
    g = golden('ops')
    x = T(g['warp_x'])
    flow = torch.zeros(2, 2, 17, 23) if case == '_zero' else T(g['warp_flow' + case])
    close(O.backward_warp(x, flow), g['warp_out' + case], 3e-6, 'warp' + case)


def test_warp_small(golden):
    g = golden('ops')
    close(O.backward_warp(T(g['warp_x_small']), T(g['warp_flow_small'])),
          g['warp_out_small'], 3e-6)


@pytest.mark.parametrize('s', [2, 4])
def test_space_to_depth_exact(golden, s):
    g = golden('ops')
    assert np.array_equal(O.space_to_depth(T(g[f's2d{s}_x']), s).numpy(), g[f's2d{s}_out'])


@pytest.mark.parametrize('s', [2, 4])
def test_upsamplers(golden, s):
    g = golden('ops')
    x = T(g['up_x'])
    assert np.allclose(O.bicubic_kernels(s).numpy(), g[f'bicubic{s}_kernels'], atol=1e-7)
    close(O.bicubic_upsample(x, s), g[f'bicubic{s}_out'], 1e-6, 'bicubic')
    close(O.bilinear_upsample(x, s), g[f'bilinear{s}_out'], 1e-6, 'bilinear')


def test_reflect_pad_exact(golden):
    g = golden('ops')
    assert np.array_equal(O.reflect_pad_br(T(g['up_x']), 6, 5).numpy(), g['reflect_out'])


def test_quantise_exact(golden):
    g = golden('ops')
    assert np.array_equal(O.float32_to_uint8(g['quant_x']), g['quant_out'])


def test_bd_downsample(golden):
    g = golden('ops')
    k = O.gaussian_kernel2d(1.5)
    assert np.allclose(k, g['bd_kernel'][0, 0], atol=1e-8)
    gt = T(g['bd_gt'])
    close(O.downsample_bd(gt, 1.5, 4, False), g['bd_out_train'], 1e-6)
    close(O.downsample_bd(gt, 1.5, 4, True), g['bd_out_test'], 1e-6)
    close(O.downsample_bd(gt, 1.5, 2, True), g['bd_out_test_s2'], 1e-6)


def test_ycbcr_exact(golden):
    g = golden('ops')
    assert np.array_equal(O.rgb_to_ycbcr(g['ycbcr_x']), g['ycbcr_out'])


CFGS = [('BD', 4), ('BI', 2), ('BD', 2)]


@pytest.mark.parametrize('deg,s', CFGS)
def test_fnet(golden, deg, s):
    g = golden(f'gen_{deg}{s}')
    sd = O._sub(generator_state_dict(scale=s, degradation=deg), 'fnet.')
    for hw in ('22x40', '16x24'):
        out = O.fnet_forward(sd, T(g[f'fnet_{hw}_x1']), T(g[f'fnet_{hw}_x2']))
        assert out.shape[2] % 8 == 0 and out.shape[3] % 8 == 0
        # flow is tanh*24: a unit in the last place of the pre-activation is
        # worth up to 24x that on the output
        close(out, g[f'fnet_{hw}_out'], 1e-4, 'fnet ' + hw)


@pytest.mark.parametrize('deg,s', CFGS)
def test_srnet(golden, deg, s):
    g = golden(f'gen_{deg}{s}')
    sd = O._sub(generator_state_dict(scale=s, degradation=deg), 'srnet.')
    out = O.srnet_forward(sd, T(g['srnet_lr']), T(g['srnet_tran']), s, deg)
    close(out, g['srnet_out'], 2e-5, 'srnet')


@pytest.mark.parametrize('deg,s', CFGS)
def test_step(golden, deg, s):
    g = golden(f'gen_{deg}{s}')
    sd = generator_state_dict(scale=s, degradation=deg)
    for hw in ('22x40', '21x37'):
        out = O.frnet_step(sd, T(g[f'step_{hw}_lr_curr']), T(g[f'step_{hw}_lr_prev']),
                           T(g[f'step_{hw}_hr_prev']), s, deg)
        close(out, g[f'step_{hw}_out'], 5e-5, 'step ' + hw)


@pytest.mark.parametrize('deg,s', CFGS)
def test_infer_sequence_u8(golden, deg, s):
    g = golden(f'gen_{deg}{s}')
    sd = generator_state_dict(scale=s, degradation=deg)
    out = O.infer_sequence(sd, T(g['infer_lr']), s, deg)
    ref = g['infer_out_u8']
    assert out.shape == ref.shape and out.dtype == np.uint8
    diff = np.abs(out.astype(np.int16) - ref.astype(np.int16))
    # uint8 after a 7-frame recurrence: allow rare +-1 flips at rounding ties
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, ((diff != 0).mean(), diff.max())


@pytest.mark.parametrize('deg,s', CFGS)
def test_forward_sequence(golden, deg, s):
    g = golden(f'gen_{deg}{s}')
    sd = generator_state_dict(scale=s, degradation=deg)
    out = O.forward_sequence(sd, T(g['fseq_lr']), s, deg)
    for k, tol in (('hr_data', 1e-4), ('hr_flow', 5e-4), ('lr_flow', 1e-4),
                   ('lr_prev', 0), ('lr_curr', 0)):
        close(out[k], g['fseq_' + k], tol, 'fseq ' + k)


def test_profile_counts(golden):
    g = golden('fullsize')
    for tag, (s, h, w) in (('A', (4, 134, 320)), ('E', (2, 268, 640))):
        gf, pr = O.profile_counts(3, 3, 64, 10, s, h, w)
        assert np.allclose([gf['FNet'], gf['SRNet']], g[f'profile_{tag}_gflops'], rtol=1e-9)
        assert [pr['FNet'], pr['SRNet']] == list(g[f'profile_{tag}_params'])


def _digest_check(out, g, prefix, atol):
    flat = out.detach().double().reshape(-1)
    idx = T(g[prefix + 'sample_idx'])
    err = (flat[idx].float().numpy() - g[prefix + 'samples'])
    assert np.abs(err).max() <= atol, np.abs(err).max()
    assert abs(flat.mean().item() - float(g[prefix + 'mean'])) <= atol
    assert abs(flat.norm().item() - float(g[prefix + 'l2'])) <= atol * flat.numel() ** 0.5


@pytest.mark.parametrize('tag,deg,s,h,w', [('A', 'BD', 4, 134, 320)])
def test_fullsize_digest(golden, tag, deg, s, h, w):
    """BASELINE config 1 shape (profile.sh 3x134x320), digest of the reference's
    step() output on seeded inputs."""
    g = golden('fullsize')
    sd = generator_state_dict(scale=s, degradation=deg)
    rs = lambda seed, shape: T(np.random.RandomState(seed).uniform(0, 1, shape).astype(np.float32))
    with torch.no_grad():
        out = O.frnet_step(sd, rs(100, (1, 3, h, w)), rs(101, (1, 3, h, w)),
                           rs(102, (1, 3, s * h, s * w)), s, deg)
    _digest_check(out, g, f'full_{tag}_', 2e-4)
