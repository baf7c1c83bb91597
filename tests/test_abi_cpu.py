"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/tecogan_hip.h declares, argument validation returns error
codes (no crash, no exception across the boundary), and the host-side mirror
keeps the reference's state-dict layout and profile() numbers.  No kernel is
launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import tecogan_pytorch_amd  # noqa: F401
from tecogan_pytorch_amd import _lib as L
from tecogan_pytorch_amd.models.networks import FRNet, define_generator
from procedural_weights import generator_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'tecogan_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tg_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(L.LIB_PATH):
        pytest.fail(f'{L.LIB_PATH} missing: run __graft_entry__.build()')
    handle = ctypes.CDLL(L.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(handle, s), f'{s} declared in tecogan_hip.h but not exported'
    # and the Python binding table covers the header exactly
    assert sorted(L.SIGNATURES) == syms


def test_error_codes_not_exceptions():
    lib = L.lib()
    assert lib.tg_version() >= 100
    # null pointers -> TG_E_ARG with a message
    rc = lib.tg_conv3x3_fwd(None, 0, 3, None, 0, None, 64, None, None, 0, None, 0,
                            1, 3, 64, 8, 8, 0, None)
    assert rc == -2 and b'null' in lib.tg_last_error_string()
    rc = lib.tg_backward_warp_fwd(None, None, None, 1, 3, 8, 8, None)
    assert rc == -2
    assert lib.tg_conv3x3_packed_floats(3, 64, 48) == 0          # bad ocb
    assert lib.tg_conv3x3_packed_floats(51, 64, 64) == 7 * 9 * 8 * 64
    assert lib.tg_conv3x3_pick_ocb(32) == 32 and lib.tg_conv3x3_pick_ocb(64) == 64
    bad = L.FrnetCfg(4, 3, 64, 10, 4, 1, 1, 134, 320)            # in_nc must be 3
    assert lib.tg_frnet_workspace_floats(ctypes.byref(bad)) == 0
    ok = L.FrnetCfg(3, 3, 64, 10, 4, 1, 1, 134, 320)
    assert lib.tg_frnet_workspace_floats(ctypes.byref(ok)) > 64 * 16 * 134 * 320


def test_ops_refuse_cpu_tensors():
    from tecogan_pytorch_amd import ops
    with pytest.raises(L.TecoganHipError):
        ops.backward_warp(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))
    with pytest.raises(L.TecoganHipError):
        ops.space_to_depth(torch.zeros(1, 3, 8, 8), 2)


@pytest.mark.parametrize('deg,s,nkeys', [('BD', 4, 78), ('BI', 2, 74), ('BD', 2, 76)])
def test_state_dict_layout_and_strict_load(deg, s, nkeys):
    g = FRNet(3, 3, 64, 10, deg, s)
    sd = generator_state_dict(scale=s, degradation=deg)
    assert len(g.state_dict()) == nkeys
    assert set(g.state_dict()) == set(sd)
    for k, v in g.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    g.load_state_dict(sd, strict=True)


def test_profile_matches_reference_counts(golden):
    g = golden('fullsize')
    for tag, deg, s, size in (('A', 'BD', 4, (3, 134, 320)), ('E', 'BI', 2, (3, 268, 640))):
        net = FRNet(3, 3, 64, 10, deg, s)
        gf, pr = net.profile(size)
        assert list(gf) == ['FNet', 'SRNet'] and list(pr) == ['FNet', 'SRNet']
        assert np.allclose([gf['FNet'], gf['SRNet']], g[f'profile_{tag}_gflops'], rtol=1e-9)
        assert [pr['FNet'], pr['SRNet']] == list(g[f'profile_{tag}_params'])


def test_define_generator_contract():
    opt = {'scale': 4, 'dataset': {'degradation': {'type': 'BD'}},
           'model': {'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10}}}
    net = define_generator(opt)
    assert isinstance(net, FRNet) and net.scale == 4
    opt['model']['generator']['name'] = 'nope'
    with pytest.raises(ValueError):
        define_generator(opt)


def test_default_init_statistics_match_torch_conv():
    """Random-init parity with the reference = PyTorch default Conv2d init."""
    torch.manual_seed(0)
    net = FRNet(3, 3, 64, 10, 'BD', 4)
    w = net.srnet.resblocks[0].conv['0'].weight
    bound = 1 / (64 * 9) ** 0.5
    assert w.abs().max() <= bound + 1e-7 and w.std() > 0.5 * bound


def test_lr_schedules_match_torch():
    """MultiStepLR (FRVSR ymls) and CosineAnnealingRestartLR closed forms vs torch's own."""
    import math
    from tecogan_pytorch_amd.models.optim import MultiStepLR, CosineAnnealingRestartLR, define_lr_schedule

    class Opt:
        def __init__(self, lr):
            self.param_groups = [{'lr': lr}]
    mine = Opt(1e-4)
    sch = define_lr_schedule({'type': 'MultiStepLR', 'milestones': [3, 7], 'gamma': 0.5}, mine)
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.SGD([p], lr=1e-4)
    ref = torch.optim.lr_scheduler.MultiStepLR(ref_opt, milestones=[3, 7], gamma=0.5)
    for _ in range(10):
        sch.step(); ref_opt.step(); ref.step()
        assert abs(mine.param_groups[0]['lr'] - ref_opt.param_groups[0]['lr']) < 1e-12
    mine = Opt(2e-4)
    cs = CosineAnnealingRestartLR(mine, [4, 4], [1, 0.5], 1e-7)
    exp = []
    for it in range(1, 9):
        idx = 0 if it <= 4 else 1
        restart = 0 if idx == 0 else 4
        exp.append(1e-7 + [1, 0.5][idx] * 0.5 * (2e-4 - 1e-7) * (1 + math.cos(math.pi * (it - restart) / 4)))
    got = []
    for _ in range(8):
        cs.step(); got.append(mine.param_groups[0]['lr'])
    assert np.allclose(got, exp, rtol=1e-12)
    assert define_lr_schedule({'type': 'FixedLR'}, mine) is None
    with pytest.raises(ValueError):
        define_lr_schedule({'type': 'Nope'}, mine)


def test_winograd_rule_and_packed_size_are_host_functions():
    """tg_conv3x3_prefers_wino / tg_conv3x3_wino_packed_floats do no device work: the layer-form
    rule the frame plan and the training tape rely on can be checked without a GPU."""
    from tecogan_pytorch_amd import _lib
    lib = _lib.lib()
    assert lib.tg_conv3x3_prefers_wino(1, 64, 64, 134, 320) == 1       # SRNet layer, configs[1]
    assert lib.tg_conv3x3_prefers_wino(1, 51, 64, 134, 320) == 1       # conv_in: two sources, K padded
    assert lib.tg_conv3x3_prefers_wino(1, 64, 64, 268, 640) == 1       # configs[4]
    assert lib.tg_conv3x3_prefers_wino(2, 64, 64, 64, 64) == 0         # training frames: one-shot kernel
    assert lib.tg_conv3x3_prefers_wino(1, 6, 32, 134, 320) == 0        # cin < 16, cout % 64 != 0
    assert lib.tg_conv3x3_prefers_wino(1, 64, 32, 134, 320) == 0
    assert lib.tg_conv3x3_prefers_wino(0, 64, 64, 134, 320) == 0
    # [K steps of 4][oc blocks of 16][4][64 lanes][4], K padded to 16, oc to 64
    assert lib.tg_conv3x3_wino_packed_floats(64, 64) == 16 * 4 * 4 * 64 * 4
    assert lib.tg_conv3x3_wino_packed_floats(51, 64) == 16 * 4 * 4 * 64 * 4
    assert lib.tg_conv3x3_wino_packed_floats(27, 128) == 8 * 8 * 4 * 64 * 4
    assert lib.tg_conv3x3_wino_packed_floats(0, 64) == -1


def test_round4_host_side_shape_rules():
    """Host-side rules of the round-4 entry points (no launch): which Conv2d(4, 2, 1) shapes the direct kernels take,
    how much workspace their small-map forms need (the input channels split over workgroups + a summing launch),
    pack sizes, and null-pointer / shape refusals as error codes."""
    lib = L.lib()
    sup = lib.tg_conv4x4s2_supported
    assert sup(24, 64, 64, 128, 128) and sup(12, 64, 128, 64, 64) and sup(1, 128, 256, 2, 192)
    assert sup(24, 64, 128, 32, 32) and sup(24, 128, 256, 16, 16)          # the small maps of the deeper blocks
    assert not sup(24, 27, 64, 128, 128) and not sup(24, 64, 96, 128, 128)  # 64-channel blocks both ways
    assert not sup(24, 64, 64, 6, 32) and not sup(24, 64, 64, 8, 16) and not sup(24, 64, 64, 16, 48)
    assert not sup(0, 64, 64, 128, 128) and not sup(1, 64, 64, 127, 128)
    wsf = lib.tg_conv4x4s2_workspace_floats
    assert wsf(24, 64, 64, 128, 128, 0) == 0 and wsf(24, 64, 64, 128, 128, 1) == 0      # large maps: one launch
    out_f, dx_f = 24 * 256 * 8 * 8, 24 * 128 * 16 * 16
    kf, kd = wsf(24, 128, 256, 16, 16, 0), wsf(24, 128, 256, 16, 16, 1)
    assert kf % out_f == 0 and 2 <= kf // out_f <= 16 and kd % dx_f == 0 and 2 <= kd // dx_f <= 16
    assert wsf(24, 27, 64, 128, 128, 0) == 0
    assert lib.tg_conv4x4s2_packed_floats(64, 128) == 64 * 128 * 16
    assert lib.tg_conv4x4s2_fwd(None, None, None, None, 24, 64, 64, 128, 128, None) == -2
    assert lib.tg_conv4x4s2_dgrad(None, None, None, 0, None, None, 24, 64, 64, 128, 128, None) == -2
    assert lib.tg_conv4x4s2_pack(None, None, None, 64, 64, None) == -2
    assert lib.tg_conv3x3_wino_resident_ct_floats() == 16 * 4 * 3 * 64 * 4
    assert lib.tg_conv3x3_wino_resident_ct_pack(None, None, None) == -2
    assert lib.tg_backward_warp_s2d_fwd(None, None, None, 1, 3, 8, 8, 4, None) == -2


def test_in_tree_library_is_not_a_lab_build():
    """tg_build_info(): the library the package loads was built by csrc/build.sh with its fixed flags, no lab switch
    compiled in (ablation switches -- one of them wrong-result by design -- sat in the shipped
    translation unit behind -D flags an EXTRA_FLAGS could set)."""
    info = L.lib().tg_build_info().decode()
    assert info.startswith('lab=0 wres_lab_bits=0 flags=--offload-arch=gfx950 -O3 '), info
    assert '-DTG_LAB' not in info and '-DWR_' not in info and '-DWG_ABL' not in info and '-DTG_W' not in info, info
    assert 'tg_conv3x3_wino_res.hip: -fno-slp-vectorize' in info, info     # the per-file flag the resident kernel was tuned with


def test_build_script_refuses_extra_flags_for_the_in_tree_library():
    import subprocess
    r = subprocess.run(['bash', os.path.join(ROOT, 'tecogan-pytorch_amd', 'csrc', 'build.sh')],
                       env=dict(os.environ, EXTRA_FLAGS='-DWR_FAKEBANK=1'), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'refused for the in-tree library' in r.stderr
    r = subprocess.run(['bash', os.path.join(ROOT, 'tecogan-pytorch_amd', 'csrc', 'build.sh')],
                       env=dict(os.environ, TG_LAB_BUILD='1', OUT=os.path.join(ROOT, 'tecogan-pytorch_amd', 'lab')),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'may not write into the package' in r.stderr
    assert not os.path.isdir(os.path.join(ROOT, 'tecogan-pytorch_amd', 'lab')) or not os.listdir(os.path.join(ROOT, 'tecogan-pytorch_amd', 'lab'))
