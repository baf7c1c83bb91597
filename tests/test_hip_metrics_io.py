"""GPU: the steps either side of the recurrent loop:
uint8 HWC frames -> fp32 CHW on the device, BD degradation of a uint8 GT clip, and the
PSNR-Y metric of compute_PSNR on device-resident uint8 frames."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tecogan_oracle as O

DEV = 'cuda'


@pytest.fixture(scope='module')
def ops():
    from tecogan_pytorch_amd import ops as o
    return o


def test_luma_matches_numpy_for_every_rgb_triple(ops):
    """rgb_to_ycbcr's Y (data_utils.py:56-77: float64 r*T0 + g*T1 + b*T2 + 16, clip,
    round-half-even, uint8) for all 2^24 colours.  The kernel rounds every product and sum
    separately (IEEE, platform independent) and must equal that definition exactly.  The
    reference evaluates the sum with np.matmul, i.e. whatever FMA order the host BLAS uses:
    it differs from the separately-rounded value for 3 of the 16.7 M colours, all exact .5
    ties ((1,173,225), (12,174,191), (24,46,73) on this image's numpy), by one level."""
    from tecogan_pytorch_amd.metrics.psnr import rgb_to_ycbcr, _T
    v = np.arange(256, dtype=np.uint8)
    blas_diff = 0
    for r0 in range(0, 256, 32):
        rgb = np.stack(np.meshgrid(v[r0:r0 + 32], v, v, indexing='ij'), -1).reshape(-1, 3)
        f = rgb.astype(np.float64)
        seq = ((f[:, 0] * _T[0, 0] + f[:, 1] * _T[1, 0]) + f[:, 2] * _T[2, 0]) + 16.0
        ref = np.clip(seq, 0, 255).round().astype(np.uint8)
        out = ops.luma_u8(torch.from_numpy(np.ascontiguousarray(rgb)).to(DEV)).cpu().numpy()
        assert np.array_equal(out, ref), (r0, int((out != ref).sum()))
        mm = rgb_to_ycbcr(rgb)[:, 0]
        d = np.abs(out.astype(int) - mm.astype(int))
        assert d.max() <= 1
        blas_diff += int(d.sum())
    assert blas_diff <= 8, blas_diff


@pytest.mark.parametrize('cs', ['y', 'rgb'])
def test_psnr_device_vs_reference_formula(ops, cs):
    from tecogan_pytorch_amd.metrics.psnr import compute_psnr, compute_psnr_device
    g = np.random.RandomState(3)
    true = g.randint(0, 256, (5, 37, 53, 3)).astype(np.uint8)
    pred = np.clip(true.astype(int) + g.randint(-6, 7, true.shape), 0, 255).astype(np.uint8)
    pred[2] = true[2]                                     # identical frame -> inf
    dev = compute_psnr_device(torch.from_numpy(true).to(DEV), torch.from_numpy(pred).to(DEV), cs)
    for i in range(true.shape[0]):
        ref = compute_psnr(true[i], pred[i], cs)
        # equal unless a frame contains one of the 3 tie colours (see the luma test): then
        # the squared-error sum may differ by a single level on that pixel
        assert dev[i] == ref or abs(dev[i] - ref) <= 1e-3, (i, dev[i], ref)
    assert dev[2] == np.inf
    # oracle's psnr (same formula, its own restatement)
    assert abs(O.psnr(true[0], pred[0], y_only=(cs == 'y')) - dev[0]) <= 1e-3


def test_dequantize_u8_hwc(ops):
    g = np.random.RandomState(4)
    x = torch.from_numpy(g.randint(0, 256, (3, 19, 23, 3)).astype(np.uint8))
    ref = x.permute(0, 3, 1, 2).float() / 255.0
    out = ops.dequantize_u8_hwc(x.to(DEV))
    assert torch.equal(out.cpu(), ref)
    with pytest.raises(Exception):
        ops.dequantize_u8_hwc(x.float().to(DEV))


def test_prepare_inference_data_from_uint8_gt(ops):
    """BD test-time path (base_model.py:98-118): uint8 thwc GT -> blurred + decimated LR."""
    from tecogan_pytorch_amd.models import define_model
    g = np.random.RandomState(5)
    gt = torch.from_numpy(g.randint(0, 256, (3, 48, 64, 3)).astype(np.uint8))
    opt = {'scale': 4, 'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1, 'is_train': False,
           'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5}},
           'model': {'name': 'FRVSR', 'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64,
                                                    'nb': 10, 'load_path': None}},
           'test': {'padding_mode': 'reflect', 'num_pad_front': 0}}
    m = define_model(opt)
    m.prepare_inference_data({'gt': gt})
    ref = O.downsample_bd(gt.permute(0, 3, 1, 2).float() / 255.0, 1.5, 4, pad_data=True)
    assert tuple(m.lr_data.shape) == tuple(ref.shape)
    assert (m.lr_data.cpu() - ref).abs().max().item() <= 1e-6


def test_main_test_mode_device_psnr_matches_host_metric():
    """main.test (codes/main.py:132-207): uint8 GT in, BD degradation + inference + PSNR-Y on
    the device; equals the reference's host-side protocol (numpy frames + compute_PSNR)."""
    from procedural_weights import generator_state_dict, smooth_clip
    from tecogan_pytorch_amd import main as M
    from tecogan_pytorch_amd.models import define_model
    from tecogan_pytorch_amd.metrics.psnr import compute_psnr
    opt = M.default_opt()
    opt.update({'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1, 'is_train': False})
    opt['model']['name'] = 'FRVSR'
    opt['test']['num_pad_front'] = 2
    seqs = []
    for i in range(2):
        clip = smooth_clip(5, 3, 64, 96, seed=20 + i)                 # HR GT, fp32 [0,1]
        gt = (clip.permute(0, 2, 3, 1) * 255).round().clamp(0, 255).to(torch.uint8).contiguous()
        seqs.append({'gt': gt, 'seq_idx': f'seq{i}'})
    sd = generator_state_dict(scale=4, degradation='BD')
    import tecogan_pytorch_amd.main as mainmod
    real_define = mainmod.define_model

    def define_with_weights(o):
        m = real_define(o)
        m.net_G.load_state_dict(sd, strict=True)
        return m
    mainmod.define_model = define_with_weights
    try:
        got = M.test(opt, seqs).tolist()
    finally:
        mainmod.define_model = real_define
    m = define_with_weights(opt)
    for i, d in enumerate(seqs):
        m.prepare_inference_data(d)
        hr = m.infer()                                                 # numpy thwc uint8
        ref = float(np.mean([compute_psnr(d['gt'].numpy()[k], hr[k]) for k in range(len(hr))]))
        assert abs(got[i] - ref) <= 1e-3, (i, got[i], ref)
        assert 5.0 < ref < 60.0


def test_copy_ceiling_copies_and_validates_arguments():
    """tg_copy_ceiling (bench.py: roofline_warp*.copy_ceiling): a float4 grid-stride copy with a caller-chosen grid --
    every byte arrives for grids smaller and larger than the data, bad arguments are refused."""
    from tecogan_pytorch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for n4, blocks, threads in ((1, 1, 64), (1000, 3, 256), (514560, 670, 512), (514560, 4096, 256)):
        a = torch.rand(4 * n4, device='cuda')
        b = torch.zeros(4 * n4 + 4, device='cuda')
        L.check(lib.tg_copy_ceiling(a.data_ptr(), b.data_ptr(), 16 * n4, blocks, threads, st), 'tg_copy_ceiling')
        torch.cuda.synchronize()
        assert torch.equal(b[:4 * n4], a) and float(b[4 * n4:].abs().sum()) == 0.0
    a = torch.rand(64, device='cuda')
    b = torch.empty(64, device='cuda')
    assert lib.tg_copy_ceiling(a.data_ptr(), b.data_ptr(), 100, 1, 64, st) != 0          # not a multiple of 16
    assert lib.tg_copy_ceiling(a.data_ptr() + 4, b.data_ptr(), 64, 1, 64, st) != 0       # misaligned
    assert lib.tg_copy_ceiling(a.data_ptr(), b.data_ptr(), 64, 1, 100, st) != 0          # threads not a multiple of 64
    assert lib.tg_copy_ceiling(None, b.data_ptr(), 64, 1, 64, st) != 0
