"""Tensor-level wrappers over the C ABI.  PyTorch is plumbing here: it owns the
device memory and the stream; every computation is a HIP kernel from
libtecogan_hip.so.  All tensors must be CUDA(HIP) fp32 and contiguous."""
import torch

from . import _lib as L
from ._lib import (ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_TANH24,  # noqa: F401
                   UP_NONE, UP_BICUBIC, UP_BILINEAR)

UP_MODE = {'BD': UP_BICUBIC, 'BI': UP_BILINEAR}


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream (the raw-handle query is ~10x cheaper than building a
    torch.cuda.Stream object; a training step asks ~650 times)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=torch.float32):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise L.TecoganHipError(f'{name}: expected a CUDA/HIP tensor (no CPU path exists)')
    if t.dtype != dtype:
        raise L.TecoganHipError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise L.TecoganHipError(f'{name}: tensor must be contiguous')
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def param_version(w):
    """Cache key for anything derived from a parameter (packed weights, plans)."""
    return (w.data_ptr(), w._version, getattr(w, '_tg_version', 0))


def bump_version(w):
    """Call after modifying a parameter through its raw pointer (fused Adam)."""
    w._tg_version = getattr(w, '_tg_version', 0) + 1


def pick_ocb(cout):
    return L.lib().tg_conv3x3_pick_ocb(int(cout))


def pack_conv3x3(weight, transposed=False, ocb=None):
    """nn.Conv2d.weight (cout,cin,3,3) [or ConvTranspose2d (cin,cout,3,3)] ->
    kernel layout.  Returns (packed, cin, cout, ocb)."""
    w = _chk(weight.detach(), 'weight')
    if w.dim() != 4 or w.shape[2:] != (3, 3):
        raise L.TecoganHipError(f'pack_conv3x3: weight shape {tuple(w.shape)}')
    if transposed:
        cin, cout = w.shape[0], w.shape[1]
        ocb = 64
    else:
        cout, cin = w.shape[0], w.shape[1]
        ocb = ocb or pick_ocb(cout)
    nfl = L.lib().tg_conv3x3_packed_floats(cin, cout, ocb)
    out = torch.empty(nfl, dtype=torch.float32, device=w.device)
    L.check(L.lib().tg_conv3x3_pack(w.data_ptr(), out.data_ptr(), cin, cout, ocb,
                                    1 if transposed else 0, _stream()), 'tg_conv3x3_pack')
    return out, cin, cout, ocb


def conv3x3(x, wpk, bias, cin, cout, ocb, act=ACT_NONE, x2=None, res=None, out=None,
            pool=False, ksplit=None, relu_mask=None):
    """y = act(conv3x3(cat[x, x2]) + bias) (+ res) [-> maxpool2 when pool].
    x: (n,c1,h,w), x2: (n,cin-c1,h,w).  Small, deep layers go through the
    deterministic split-K path (ksplit=None: library heuristic).
    relu_mask (shape of y): y is zeroed where relu_mask <= 0 in the same epilogue (the
    ReLU backward of the layer that produced relu_mask)."""
    _chk(x, 'x')
    n, c1, h, w = x.shape
    if ksplit is None:
        ksplit = 1 if (res is not None or relu_mask is not None) else \
            L.lib().tg_conv3x3_pick_ksplit(n, cin, cout, h, w)
    if ksplit > 1:
        if res is not None or relu_mask is not None:
            raise L.TecoganHipError('conv3x3: split-K path has no residual / mask epilogue')
        if x2 is not None:
            _chk(x2, 'x2')
        part = torch.empty(ksplit * n * cout * h * w, dtype=torch.float32, device=x.device)
        oh, ow = (h // 2, w // 2) if pool else (h, w)
        if out is None:
            out = torch.empty(n, cout, oh, ow, dtype=torch.float32, device=x.device)
        L.check(L.lib().tg_conv3x3_splitk_fwd(
            x.data_ptr(), c1 * h * w, c1, _ptr(x2), 0 if x2 is None else x2.shape[1] * h * w,
            wpk.data_ptr(), ocb, _ptr(bias), out.data_ptr(), n, cin, cout, h, w, act, ksplit,
            part.data_ptr(), 1 if pool else 0, _stream()), 'tg_conv3x3_splitk_fwd')
        return out
    if pool:
        return maxpool2(conv3x3(x, wpk, bias, cin, cout, ocb, act, x2=x2, res=res, ksplit=1))
    if x2 is not None:
        _chk(x2, 'x2')
        if x2.shape[0] != n or x2.shape[2:] != x.shape[2:] or c1 + x2.shape[1] != cin:
            raise L.TecoganHipError(f'conv3x3: x {tuple(x.shape)} x2 {tuple(x2.shape)} cin {cin}')
    elif c1 != cin:
        raise L.TecoganHipError(f'conv3x3: x has {c1} channels, weights expect {cin}')
    if out is None:
        out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    _chk(out, 'out')
    if res is not None:
        _chk(res, 'res')
        if res.shape != out.shape:
            raise L.TecoganHipError('conv3x3: residual shape mismatch')
    hw = h * w
    if relu_mask is not None:
        _chk(relu_mask, 'relu_mask')
        if relu_mask.shape != out.shape:
            raise L.TecoganHipError('conv3x3: relu_mask shape mismatch')
        L.check(L.lib().tg_conv3x3_fwd_masked(
            x.data_ptr(), c1 * hw, c1, _ptr(x2), 0 if x2 is None else x2.shape[1] * hw,
            wpk.data_ptr(), ocb, _ptr(bias), _ptr(res), cout * hw, relu_mask.data_ptr(), cout * hw,
            out.data_ptr(), cout * hw, n, cin, cout, h, w, act, _stream()), 'tg_conv3x3_fwd_masked')
        return out
    L.check(L.lib().tg_conv3x3_fwd(
        x.data_ptr(), c1 * hw, c1, _ptr(x2), 0 if x2 is None else x2.shape[1] * hw,
        wpk.data_ptr(), ocb, _ptr(bias), _ptr(res), cout * hw, out.data_ptr(), cout * hw,
        n, cin, cout, h, w, act, _stream()), 'tg_conv3x3_fwd')
    return out


TAPS_ALL, TAPS_01, TAPS_12, TAPS_1 = 0, 1, 2, 3


def conv3x3_phased(x, wpk, cin, cout, ocb, tapsel, cphase, taps_phase0, taps_phase1, out=None, relu_mask=None):
    """conv3x3 (no bias / activation) of a space-to-depth embedded strided conv, skipping the
    taps a sub-pixel phase does not own (include/tecogan_hip.h: tg_conv3x3_fwd_phased)."""
    _chk(x, 'x')
    n, c1, h, w = x.shape
    if c1 != cin:
        raise L.TecoganHipError(f'conv3x3_phased: x has {c1} channels, weights expect {cin}')
    if out is None:
        out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    if relu_mask is not None:
        _chk(relu_mask, 'relu_mask')
        if relu_mask.shape != out.shape:
            raise L.TecoganHipError('conv3x3_phased: relu_mask shape mismatch')
    ks = 1 if relu_mask is not None or not out.is_contiguous() else \
        L.lib().tg_conv3x3_phased_pick_ksplit(n, cin, cout, h, w, ocb)
    if ks > 1:          # too few tiles for the device: channel chunks split over ks workgroup sets + finalize
        part = torch.empty(ks * n * cout * h * w, dtype=torch.float32, device=x.device)
        L.check(L.lib().tg_conv3x3_fwd_phased_splitk(x.data_ptr(), cin * h * w, wpk.data_ptr(), ocb, None,
                                                     out.data_ptr(), n, cin, cout, h, w, ACT_NONE, int(tapsel),
                                                     int(cphase), int(taps_phase0), int(taps_phase1), ks,
                                                     part.data_ptr(), _stream()), 'tg_conv3x3_fwd_phased_splitk')
        return out
    L.check(L.lib().tg_conv3x3_fwd_phased_masked(x.data_ptr(), cin * h * w, wpk.data_ptr(), ocb, None,
                                                 _ptr(relu_mask), cout * h * w, out.data_ptr(), cout * h * w,
                                                 n, cin, cout, h, w, ACT_NONE, int(tapsel), int(cphase),
                                                 int(taps_phase0), int(taps_phase1), _stream()),
            'tg_conv3x3_fwd_phased_masked')
    return out


def pack_conv3x3_m16(weight, transposed=0):
    """tg_conv3x3_pack16: the A-operand layout of the 16 x 16 x 4 chained kernel (cin, cout <= 64);
    transposed = 2: the data gradient (channel roles swapped, taps rotated)."""
    w = _chk(weight.detach().contiguous(), 'weight')
    cout, cin = w.shape[0], w.shape[1]
    out = torch.empty(L.lib().tg_conv3x3_pack16_floats(), dtype=torch.float32, device=w.device)
    if transposed == 2:
        L.check(L.lib().tg_conv3x3_pack16(w.data_ptr(), out.data_ptr(), cout, cin, 2, _stream()), 'tg_conv3x3_pack16')
    else:
        L.check(L.lib().tg_conv3x3_pack16(w.data_ptr(), out.data_ptr(), cin, cout, 0, _stream()), 'tg_conv3x3_pack16')
    return out


def chain_pack(items, layout):
    """tg_conv3x3_chain_pack: items = [(weight (O, I_total, 3, 3), i_off, i_cnt, transposed)] -> list of packed
    tensors (views of one buffer) in the chained kernels' layout `layout` (16 | 64), ONE launch."""
    lib = L.lib()
    arr = (L.PackItem * len(items))()
    sizes = []
    for wt, i_off, i_cnt, tr in items:
        _chk(wt, 'weight')
        cin, cout = (wt.shape[0], i_cnt) if tr == 2 else (i_cnt, wt.shape[0])
        sizes.append(int(lib.tg_conv3x3_chain_packed_floats(layout, cin)))
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=items[0][0].device)
    outs, off = [], 0
    for a, (wt, i_off, i_cnt, tr), sz in zip(arr, items, sizes):
        o = buf[off:off + sz]; off += sz
        outs.append(o)
        a.w, a.out, a.transposed, a.i_total, a.i_off = wt.data_ptr(), o.data_ptr(), tr, wt.shape[1], i_off
        a.cin, a.cout = (wt.shape[0], i_cnt) if tr == 2 else (i_cnt, wt.shape[0])
    L.check(lib.tg_conv3x3_chain_pack(arr, len(items), layout, _stream()), 'tg_conv3x3_chain_pack')
    return outs


def conv3x3s2_supported(n, cin, cout, h_out, w_out):
    return bool(L.lib().tg_conv3x3s2_supported(n, cin, cout, h_out, w_out))


def conv3x3s2(x, wpk, cin, cout, relu_mask=None, out=None):
    """Stride-2 3x3 conv (tg_conv3x3s2_fwd): x (n, cin, 2h, 2w) -> (n, cout, h, w); the data gradient of
    ConvTranspose2d(k3, s2, p1, op1) with wpk = pack_conv3x3(W viewed as (cout = ci, cin = co), ocb 64)."""
    _chk(x, 'x')
    n, c, h2, w2 = x.shape
    if c != cin or h2 % 2 or w2 % 2:
        raise L.TecoganHipError(f'conv3x3s2: x {tuple(x.shape)} cin {cin}')
    h, w = h2 // 2, w2 // 2
    if out is None:
        out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (n, cout, h, w):
        raise L.TecoganHipError(f'conv3x3s2: out {tuple(out.shape)}')
    else:
        _chk(out, 'out')
    if relu_mask is not None:
        _chk(relu_mask, 'relu_mask')
        if relu_mask.shape != out.shape:
            raise L.TecoganHipError('conv3x3s2: relu_mask shape mismatch')
    L.check(L.lib().tg_conv3x3s2_fwd(x.data_ptr(), cin * h2 * w2, wpk.data_ptr(), None, _ptr(relu_mask), cout * h * w,
                                     out.data_ptr(), cout * h * w, n, cin, cout, h, w, ACT_NONE, _stream()),
            'tg_conv3x3s2_fwd')
    return out


def convt3x3s2(x, wpk, bias, cout, act=ACT_NONE, out=None):
    _chk(x, 'x')
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_convt3x3s2_fwd(x.data_ptr(), cin * h * w, wpk.data_ptr(), _ptr(bias),
                                      out.data_ptr(), cout * 4 * h * w, n, cin, cout, h, w, act,
                                      _stream()), 'tg_convt3x3s2_fwd')
    return out


def convt_pack_wz(w_out_oihw):
    """tg_convt_pack_wz: the output conv's (cz, nf, 3, 3) weight as the A operand of the Z-mode contraction."""
    _chk(w_out_oihw, 'w_out')
    cz, nf = w_out_oihw.shape[:2]
    wz = torch.empty(2 * 16 * 64, dtype=torch.float32, device=w_out_oihw.device)
    L.check(L.lib().tg_convt_pack_wz(w_out_oihw.data_ptr(), wz.data_ptr(), cz, nf, _stream()), 'tg_convt_pack_wz')
    return wz


def convt3x3s2_z(x, wpk, bias, wz, cz, cout, act=ACT_NONE, form=-1, out=None):
    """tg_convt3x3s2_z_fwd_form: ConvTranspose2d(cin, cout, 3, 2, 1, 1) + act with the following 3x3 output conv's
    channel contraction in the epilogue -> (n, 32, 2h, 2w) buffer whose first 9 * cz planes are the output conv's tap
    planes (tecogan_nets.py:119-131).  form: -1 the library rule, 0 tiled, 1 streaming, 2 streaming with a static item list, 3 tiled with a split tail (all bit-identical)."""
    _chk(x, 'x')
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, 32, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_convt3x3s2_z_fwd_form(x.data_ptr(), cin * h * w, wpk.data_ptr(), _ptr(bias), wz.data_ptr(), cz,
                                             out.data_ptr(), 32 * 4 * h * w, n, cin, cout, h, w, act, form, _stream()),
            'tg_convt3x3s2_z_fwd_form')
    return out


def convout_tail(z, cz, bias, up_src=None, up_mode=UP_NONE, up_scale=1, want_u8=False, form=-1):
    """tg_convout_tail_form: out[o] = bias[o] + sum_tap z[tap * cz + o] shifted by the tap (zero padding) [+ the
    up-sampled residual of up_src] -> fp32 (n, cz, h, w) [and the uint8 (n, h, w, cz) frame].  z: (n, 32, h, w) planes
    of tg_convt3x3s2_z_fwd.  form: -1 the library's rule, 0 one pixel per thread, 1 four (bit-identical)."""
    _chk(z, 'z')
    n, _, h, w = z.shape
    y = torch.empty(n, cz, h, w, dtype=torch.float32, device=z.device)
    u8 = torch.empty(n, h, w, cz, dtype=torch.uint8, device=z.device) if want_u8 else None
    L.check(L.lib().tg_convout_tail_form(z.data_ptr(), z.shape[1] * h * w, cz, _ptr(bias), _ptr(up_src), up_mode, up_scale,
                                         y.data_ptr(), cz * h * w, u8.data_ptr() if want_u8 else None, n, h, w, form,
                                         _stream()), 'tg_convout_tail_form')
    return (y, u8) if want_u8 else y


def conv3x3_fewin_ok(x, cout, any_size=False):
    """whether tg_conv3x3_fewin_fwd takes the launch -- and pays: below one 4 x 64 tile per CU the MFMA
    kernel is faster (2 x 128 x 128: 8.7 against 11.2 us; 2 x 256 x 256: 27.9 against 18.8 us)"""
    n, cin, h, w = x.shape
    ok = cin <= 4 and w % 4 == 0 and x.data_ptr() % 16 == 0 and (cin * h * w) % 4 == 0 and (cout * h * w) % 4 == 0
    return ok and (any_size or n * ((h + 3) // 4) * ((w + 63) // 64) >= 256)


def conv3x3_fewin(x, w_oihw, relu_mask=None, out=None):
    """tg_conv3x3_fewin_fwd: 3x3 conv from <= 4 input channels (w_oihw: (cout, cin, 3, 3)), no bias /
    activation, y = relu_mask > 0 ? conv : 0 -- the data gradient of a small-cout head."""
    _chk(x, 'x'); _chk(w_oihw, 'weight')
    n, cin, h, w = x.shape
    cout = w_oihw.shape[0]
    if tuple(w_oihw.shape) != (cout, cin, 3, 3):
        raise L.TecoganHipError(f'conv3x3_fewin: weight {tuple(w_oihw.shape)} for {cin} input channels')
    if out is None:
        out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    if relu_mask is not None:
        _chk(relu_mask, 'relu_mask')
        if relu_mask.shape != out.shape:
            raise L.TecoganHipError('conv3x3_fewin: relu_mask shape mismatch')
    L.check(L.lib().tg_conv3x3_fewin_fwd(x.data_ptr(), cin * h * w, w_oihw.data_ptr(), _ptr(relu_mask), cout * h * w,
                                         out.data_ptr(), cout * h * w, n, cin, cout, h, w, _stream()),
            'tg_conv3x3_fewin_fwd')
    return out


def conv3x3_small_res_ok(x, res):
    """whether tg_conv3x3_small_fwd_res takes this launch (w % 4 == 0, aligned planes)"""
    return x.shape[3] % 4 == 0 and x.data_ptr() % 16 == 0 and res.data_ptr() % 16 == 0 and \
        (x.shape[1] * x.shape[2] * x.shape[3]) % 4 == 0 and (res.shape[1] * res.shape[2] * res.shape[3]) % 4 == 0


def conv3x3_small(x, weight, bias, act=ACT_NONE, up_src=None, up_mode=UP_NONE, up_scale=1,
                  out=None, res=None):
    _chk(x, 'x')
    w_ = _chk(weight.detach(), 'weight')
    n, cin, h, w = x.shape
    cout = w_.shape[0]
    if out is None:
        out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    if res is not None:         # explicit residual tensor instead of an up-sampled source
        _chk(res, 'res')
        if up_src is not None or tuple(res.shape) != (n, cout, h, w):
            raise L.TecoganHipError('conv3x3_small: res must be (n, cout, h, w) and excludes up_src')
        L.check(L.lib().tg_conv3x3_small_fwd_res(
            x.data_ptr(), cin * h * w, w_.data_ptr(), _ptr(bias), res.data_ptr(), cout * h * w,
            out.data_ptr(), cout * h * w, n, cin, cout, h, w, act, _stream()), 'tg_conv3x3_small_fwd_res')
        return out
    if up_src is not None:
        _chk(up_src, 'up_src')
    L.check(L.lib().tg_conv3x3_small_fwd(
        x.data_ptr(), cin * h * w, w_.data_ptr(), _ptr(bias), _ptr(up_src), up_mode, up_scale,
        out.data_ptr(), cout * h * w, n, cin, cout, h, w, act, _stream()), 'tg_conv3x3_small_fwd')
    return out


def flowup_warp_s2d(lr_flow, hr_prev, h, w, scale, up_mode, out=None, want_hr_flow=False):
    _chk(lr_flow, 'lr_flow')
    _chk(hr_prev, 'hr_prev')
    n, c = hr_prev.shape[:2]
    fh, fw = lr_flow.shape[2:]
    if out is None:
        out = torch.empty(n, scale * scale * c, h, w, dtype=torch.float32, device=hr_prev.device)
    hr_flow = (torch.empty(n, 2, scale * h, scale * w, dtype=torch.float32, device=hr_prev.device)
               if want_hr_flow else None)
    L.check(L.lib().tg_flowup_warp_s2d_fwd(
        lr_flow.data_ptr(), fh, fw, hr_prev.data_ptr(), out.data_ptr(), scale * scale * c * h * w,
        _ptr(hr_flow), n, c, h, w, scale, up_mode, _stream()), 'tg_flowup_warp_s2d_fwd')
    return (out, hr_flow) if want_hr_flow else out


def backward_warp(x, flow):
    _chk(x, 'x')
    _chk(flow, 'flow')
    n, c, h, w = x.shape
    if flow.shape != (n, 2, h, w):
        raise L.TecoganHipError(f'backward_warp: flow {tuple(flow.shape)} vs x {tuple(x.shape)}')
    out = torch.empty_like(x)
    L.check(L.lib().tg_backward_warp_fwd(x.data_ptr(), flow.data_ptr(), out.data_ptr(), n, c, h, w,
                                         _stream()), 'tg_backward_warp_fwd')
    return out


def backward_warp_s2d(x, flow, scale):
    """space_to_depth(backward_warp(x, flow), scale) in one launch (tg_backward_warp_s2d_fwd)."""
    _chk(x, 'x')
    _chk(flow, 'flow')
    n, c, h, w = x.shape
    if flow.shape != (n, 2, h, w) or h % scale or w % scale:
        raise L.TecoganHipError(f'backward_warp_s2d: flow {tuple(flow.shape)} vs x {tuple(x.shape)}, scale {scale}')
    out = torch.empty(n, scale * scale * c, h // scale, w // scale, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_backward_warp_s2d_fwd(x.data_ptr(), flow.data_ptr(), out.data_ptr(), n, c, h, w, scale,
                                             _stream()), 'tg_backward_warp_s2d_fwd')
    return out


def space_to_depth(x, scale):
    _chk(x, 'x')
    n, c, h, w = x.shape
    oh, ow = h // scale, w // scale
    out = torch.empty(n, scale * scale * c, oh, ow, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_space_to_depth(x.data_ptr(), out.data_ptr(), scale * scale * c * oh * ow,
                                      n, c, h, w, scale, _stream()), 'tg_space_to_depth')
    return out


def upsample(x, scale, up_mode, mul=1.0):
    _chk(x, 'x')
    n, c, h, w = x.shape
    out = torch.empty(n, c, h * scale, w * scale, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_upsample_fwd(x.data_ptr(), out.data_ptr(), n * c, h, w, scale, up_mode,
                                    float(mul), _stream()), 'tg_upsample_fwd')
    return out


def maxpool2(x):
    _chk(x, 'x')
    n, c, h, w = x.shape
    out = torch.empty(n, c, h // 2, w // 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_maxpool2_fwd(x.data_ptr(), out.data_ptr(), n * c, h, w, _stream()),
            'tg_maxpool2_fwd')
    return out


def quantize_u8_hwc(x):
    """(c,h,w) fp32 -> (h,w,c) uint8 on device."""
    _chk(x, 'x')
    c, h, w = x.shape
    out = torch.empty(h, w, c, dtype=torch.uint8, device=x.device)
    L.check(L.lib().tg_quantize_u8_hwc(x.data_ptr(), out.data_ptr(), c, h, w, _stream()),
            'tg_quantize_u8_hwc')
    return out


def _chk_u8(t, name):
    if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise L.TecoganHipError(f'{name}: expected a contiguous CUDA uint8 tensor, got '
                                f'{t.dtype} {t.device} contiguous={t.is_contiguous()}')
    return t


def dequantize_u8_hwc(x):
    """(n,h,w,c) uint8 on device -> (n,c,h,w) fp32 in [0,1] (= permute + float + /255)."""
    _chk_u8(x, 'x')
    n, h, w, c = x.shape
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_dequantize_u8_hwc(x.data_ptr(), out.data_ptr(), n, c, h, w, _stream()),
            'tg_dequantize_u8_hwc')
    return out


def psnr_sse_u8(true_hwc, pred_hwc, y_only=True):
    """Per-frame sum of squared differences of (t,h,w,3) uint8 device tensors -> int64 (t,)."""
    _chk_u8(true_hwc, 'true'); _chk_u8(pred_hwc, 'pred')
    if true_hwc.shape != pred_hwc.shape or true_hwc.dim() != 4 or true_hwc.shape[3] != 3:
        raise L.TecoganHipError(f'psnr_sse_u8: shapes {tuple(true_hwc.shape)} / {tuple(pred_hwc.shape)}')
    t, h, w, _ = true_hwc.shape
    sse = torch.empty(t, dtype=torch.int64, device=true_hwc.device)
    L.check(L.lib().tg_psnr_sse_u8(true_hwc.data_ptr(), pred_hwc.data_ptr(), sse.data_ptr(), t, h, w,
                                   1 if y_only else 0, _stream()), 'tg_psnr_sse_u8')
    return sse


def luma_u8(rgb):
    """(n,3) uint8 -> (n,) uint8, the Y channel of rgb_to_ycbcr."""
    _chk_u8(rgb, 'rgb')
    out = torch.empty(rgb.shape[0], dtype=torch.uint8, device=rgb.device)
    L.check(L.lib().tg_luma_u8(rgb.data_ptr(), out.data_ptr(), rgb.shape[0], _stream()), 'tg_luma_u8')
    return out


# ---------------------------------------------------------------------------
# training-side wrappers (backward kernels, losses, optimiser)
# ---------------------------------------------------------------------------
def pack_conv3x3_dgrad(weight, ocb=None):
    """Pack a Conv2d weight (cout,cin,3,3) for its DATA gradient: a conv3x3 op
    mapping cout -> cin with 180-degree rotated taps.  Returns (packed, op_cin, op_cout, ocb)."""
    w = _chk(weight.detach().contiguous(), 'weight')
    cout, cin = w.shape[0], w.shape[1]
    ocb = ocb or pick_ocb(cin)
    nfl = L.lib().tg_conv3x3_packed_floats(cout, cin, ocb)
    out = torch.empty(nfl, dtype=torch.float32, device=w.device)
    L.check(L.lib().tg_conv3x3_pack(w.data_ptr(), out.data_ptr(), cout, cin, ocb, 2, _stream()),
            'tg_conv3x3_pack(dgrad)')
    return out, cout, cin, ocb


_WGRAD_WS = {}


def _wgrad_workspace(device, nfloats):
    # one scratch buffer per (device, stream): weight-gradient launches of the training step run on
    # the main stream and on a side stream at the same time (Tape.flush_deferred_async)
    key = (str(device), _stream())
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < nfloats:
        ws = torch.empty(int(nfloats), dtype=torch.float32, device=device)
        _WGRAD_WS[key] = ws
    return ws


def wgrad3x3(p, q, grad, cb_off=0, accumulate=True, bias_grad=None):
    """grad[a, cb_off:cb_off+cb, ky, kx] (+)= sum p[n,a,y,x] * q[n,b,y+ky-1,x+kx-1];
    bias_grad (ca,) (+)= sum of p (same launch, tg_wgrad3x3_multi_bias)."""
    if bias_grad is not None:
        return wgrad3x3_multi([p], [q], grad, cb_off=cb_off, accumulate=accumulate, bias_grad=bias_grad)
    _chk(p, 'p'); _chk(q, 'q'); _chk(grad, 'grad')
    n, ca, h, w = p.shape
    cb = q.shape[1]
    cb_total = grad.shape[1]
    if q.shape[0] != n or q.shape[2:] != p.shape[2:] or grad.shape[0] != ca or grad.shape[2:] != (3, 3):
        raise L.TecoganHipError(f'wgrad3x3: p {tuple(p.shape)} q {tuple(q.shape)} grad {tuple(grad.shape)}')
    nfl = L.lib().tg_wgrad3x3_workspace_floats(n, ca, cb_total, h, w)
    ws = _wgrad_workspace(p.device, nfl)
    L.check(L.lib().tg_wgrad3x3(p.data_ptr(), ca * h * w, q.data_ptr(), cb * h * w, grad.data_ptr(),
                                ws.data_ptr(), n, ca, cb, cb_total, cb_off, h, w,
                                1 if accumulate else 0, _stream()), 'tg_wgrad3x3')
    return grad


MAX_SEGS = 64


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def wgrad3x3_multi(p_list, q_list, grad, cb_off=0, accumulate=True, phased=None, bias_grad=None):
    """wgrad3x3 over the concatenation of equally shaped (p_i, q_i) pairs without concatenating
    them: one launch reads up to MAX_SEGS separately allocated segments.
    phased = (cphase, taps_phase0, taps_phase1): q is a space-to-depth embedded tensor; only the
    taps each sub-pixel phase owns are computed (tg_wgrad3x3_multi_phased)."""
    if len(p_list) != len(q_list) or not p_list:
        raise L.TecoganHipError('wgrad3x3_multi: empty or mismatched lists')
    for t in list(p_list) + list(q_list):
        _chk(t, 'segment')
    _chk(grad, 'grad')
    n, ca, h, w = p_list[0].shape
    cb = q_list[0].shape[1]
    cb_total = grad.shape[1]
    for p_, q_ in zip(p_list, q_list):
        if p_.shape != p_list[0].shape or q_.shape != q_list[0].shape or q_.shape[0] != n \
                or q_.shape[2:] != p_.shape[2:]:
            raise L.TecoganHipError('wgrad3x3_multi: segments must share one shape')
    if grad.shape[0] != ca or grad.shape[2:] != (3, 3):
        raise L.TecoganHipError(f'wgrad3x3_multi: grad {tuple(grad.shape)}')
    lib = L.lib()
    for i in range(0, len(p_list), MAX_SEGS):
        ps, qs = p_list[i:i + MAX_SEGS], q_list[i:i + MAX_SEGS]
        nfl = lib.tg_wgrad3x3_workspace_floats(n * len(ps), ca, cb_total, h, w)
        ws = _wgrad_workspace(grad.device, nfl)
        if phased is not None:
            if cb_off or cb != cb_total:
                raise L.TecoganHipError('wgrad3x3_multi: phased taps take the whole q tensor')
            L.check(lib.tg_wgrad3x3_multi_phased(_ptr_array(ps), _ptr_array(qs), len(ps), ca * h * w,
                                                 cb * h * w, grad.data_ptr(), ws.data_ptr(), n, ca, cb, h, w,
                                                 1 if (accumulate or i > 0) else 0, int(phased[0]),
                                                 int(phased[1]), int(phased[2]), _stream()),
                    'tg_wgrad3x3_multi_phased')
            continue
        if bias_grad is not None:       # the layer's bias gradient out of the same pass over p = dZ
            L.check(lib.tg_wgrad3x3_multi_bias(_ptr_array(ps), _ptr_array(qs), len(ps), ca * h * w, cb * h * w,
                                               grad.data_ptr(), bias_grad.data_ptr(), ws.data_ptr(), n, ca, cb,
                                               cb_total, cb_off, h, w, 1 if (accumulate or i > 0) else 0,
                                               _stream()), 'tg_wgrad3x3_multi_bias')
            continue
        L.check(lib.tg_wgrad3x3_multi(_ptr_array(ps), _ptr_array(qs), len(ps), ca * h * w, cb * h * w,
                                      grad.data_ptr(), ws.data_ptr(), n, ca, cb, cb_total, cb_off, h, w,
                                      1 if (accumulate or i > 0) else 0, _stream()), 'tg_wgrad3x3_multi')
    return grad


def wgrad3x3_convt_multi(x_list, dz_list, grad, accumulate=True, bias_grad=None):
    """tg_wgrad3x3_convt_multi: grad (ci, co, 3, 3) (+)= dW of ConvTranspose2d(ci, co, 3, 2, 1, 1) over the
    (input x_i (n, ci, h, w), output gradient dz_i (n, co, 2h, 2w)) pairs, straight from dZ; bias_grad (co,)
    (+)= the sum of dZ (same launch)."""
    if len(x_list) != len(dz_list) or not x_list:
        raise L.TecoganHipError('wgrad3x3_convt_multi: empty or mismatched lists')
    for t in list(x_list) + list(dz_list):
        _chk(t, 'segment')
    _chk(grad, 'grad')
    n, ci, h, w = x_list[0].shape
    co = dz_list[0].shape[1]
    if any(t.shape != x_list[0].shape for t in x_list) or any(t.shape != (n, co, 2 * h, 2 * w) for t in dz_list) \
            or tuple(grad.shape) != (ci, co, 3, 3):
        raise L.TecoganHipError(f'wgrad3x3_convt_multi: x {tuple(x_list[0].shape)} dz {tuple(dz_list[0].shape)} '
                                f'grad {tuple(grad.shape)}')
    lib = L.lib()
    for i in range(0, len(x_list), MAX_SEGS):
        xs, ds = x_list[i:i + MAX_SEGS], dz_list[i:i + MAX_SEGS]
        ws = _wgrad_workspace(grad.device, lib.tg_wgrad3x3_convt_workspace_floats(n * len(xs), ci, co, h, w))
        L.check(lib.tg_wgrad3x3_convt_multi(_ptr_array(xs), _ptr_array(ds), len(xs), grad.data_ptr(), _ptr(bias_grad),
                                            ws.data_ptr(), n, ci, co, h, w, 1 if (accumulate or i > 0) else 0,
                                            _stream()), 'tg_wgrad3x3_convt_multi')
    return grad


def wgrad3x3_body(dz_list, acts_list, grads, accumulate=True, dbs=None):
    """tg_wgrad3x3_body: weight gradients of the chained SRNet body's 2*nb residual-block convs over ALL
    unrolled frames in one launch.  dz_list[f] / acts_list[f]: (1 + 2nb, n, c, h, w) blocks of frame f
    (dz[2nb] = gradient of the body's output); grads: the 2nb weight-gradient tensors."""
    nl, n, c, h, w = dz_list[0].shape
    for t in list(dz_list) + list(acts_list):
        _chk(t, 'block')
        if t.shape != dz_list[0].shape:
            raise L.TecoganHipError('wgrad3x3_body: frames must share one shape')
    if len(grads) != nl - 1 or len(dz_list) != len(acts_list):
        raise L.TecoganHipError('wgrad3x3_body: mismatched lists')
    for g in grads:
        _chk(g, 'grad')
        if tuple(g.shape) != (c, c, 3, 3):
            raise L.TecoganHipError(f'wgrad3x3_body: grad {tuple(g.shape)}')
    lib = L.lib()
    for i in range(0, len(dz_list), MAX_SEGS):
        dzs, acs = dz_list[i:i + MAX_SEGS], acts_list[i:i + MAX_SEGS]
        ws = _wgrad_workspace(grads[0].device, lib.tg_wgrad3x3_body_workspace_floats(len(dzs), n, nl - 1, c, h, w))
        if dbs is not None:         # dbs[L - 1] (+)= bias gradient of layer L = 1 .. 2nb, same launch
            L.check(lib.tg_wgrad3x3_body_bias(_ptr_array(dzs), _ptr_array(acs), len(dzs), n * c * h * w,
                                              nl - 1, _ptr_array(grads), _ptr_array(dbs), ws.data_ptr(), n, c, h, w,
                                              1 if (accumulate or i > 0) else 0, _stream()), 'tg_wgrad3x3_body_bias')
            continue
        L.check(lib.tg_wgrad3x3_body(_ptr_array(dzs), _ptr_array(acs), len(dzs), n * c * h * w,
                                     nl - 1, _ptr_array(grads), ws.data_ptr(), n, c, h, w,
                                     1 if (accumulate or i > 0) else 0, _stream()), 'tg_wgrad3x3_body')


def bias_grad_body(dz_list, dbs):
    """tg_bias_grad_body: dbs[L] += bias gradient of layer L = 0 .. 2nb of the chained body, all frames."""
    nl, n, c, h, w = dz_list[0].shape
    if len(dbs) != nl:
        raise L.TecoganHipError('bias_grad_body: mismatched lists')
    for t in list(dz_list) + list(dbs):
        _chk(t, 'tensor')
    for i in range(0, len(dz_list), MAX_SEGS):
        dzs = dz_list[i:i + MAX_SEGS]
        L.check(L.lib().tg_bias_grad_body(_ptr_array(dzs), len(dzs), n * c * h * w, nl,
                                          _ptr_array(dbs), n, c, h * w, _stream()), 'tg_bias_grad_body')


def bias_grad_multi(dy_list, db, accumulate=True):
    if not dy_list:
        raise L.TecoganHipError('bias_grad_multi: empty list')
    for t in dy_list:
        _chk(t, 'segment')
        if t.shape != dy_list[0].shape:
            raise L.TecoganHipError('bias_grad_multi: segments must share one shape')
    _chk(db, 'db')
    n, c = dy_list[0].shape[:2]
    hw = dy_list[0].numel() // (n * c)
    for i in range(0, len(dy_list), MAX_SEGS):
        seg = dy_list[i:i + MAX_SEGS]
        L.check(L.lib().tg_bias_grad_multi(_ptr_array(seg), len(seg), db.data_ptr(), n, c, hw,
                                           1 if (accumulate or i > 0) else 0, _stream()),
                'tg_bias_grad_multi')
    return db


def act_bwd(dy, y, act, out=None):
    _chk(dy, 'dy'); _chk(y, 'y')
    if out is None:
        out = torch.empty_like(dy)
    L.check(L.lib().tg_act_bwd(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), act,
                               _stream()), 'tg_act_bwd')
    return out


def bias_grad(dy, db, accumulate=True):
    _chk(dy, 'dy'); _chk(db, 'db')
    n, c = dy.shape[:2]
    hw = dy.numel() // (n * c)
    L.check(L.lib().tg_bias_grad(dy.data_ptr(), db.data_ptr(), n, c, hw, 1 if accumulate else 0,
                                 _stream()), 'tg_bias_grad')
    return db


def maxpool2_bwd(x, dy):
    _chk(x, 'x'); _chk(dy, 'dy')
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    L.check(L.lib().tg_maxpool2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n * c, h, w,
                                    _stream()), 'tg_maxpool2_bwd')
    return dx


def upsample_bwd(dy, scale, up_mode, mul=1.0):
    _chk(dy, 'dy')
    n, c, oh, ow = dy.shape
    h, w = oh // scale, ow // scale
    dx = torch.empty(n, c, h, w, dtype=torch.float32, device=dy.device)
    L.check(L.lib().tg_upsample_bwd(dy.data_ptr(), dx.data_ptr(), n * c, h, w, scale, up_mode,
                                    float(mul), _stream()), 'tg_upsample_bwd')
    return dx


def backward_warp_bwd(x, flow, dy, need_img=True, need_flow=True, dflow_out=None, dimg_acc=None, s2d=1):
    """`dflow_out`: write the flow gradient into this (contiguous, flow-shaped) buffer instead
    of a new tensor (a slice of the frame-major gradient of the training unroll).  `dimg_acc`: ADD the
    image gradient to this tensor (the gradient x already has) instead of returning a new one.
    `s2d` > 1: dy is the gradient of backward_warp_s2d's output (space_to_depth layout)."""
    _chk(x, 'x'); _chk(flow, 'flow'); _chk(dy, 'dy')
    n, c, h, w = x.shape
    if s2d > 1:
        assert dy.shape == (n, s2d * s2d * c, h // s2d, w // s2d), (dy.shape, x.shape, s2d)
        if dimg_acc is not None:
            _chk(dimg_acc, 'dimg_acc')
            dimg = dimg_acc
        else:
            dimg = torch.empty_like(x) if need_img else None
        dflow = None
        if need_flow:
            dflow = _chk(dflow_out, 'dflow_out') if dflow_out is not None else torch.empty_like(flow)
        L.check(L.lib().tg_backward_warp_s2d_bwd(x.data_ptr(), flow.data_ptr(), dy.data_ptr(), _ptr(dimg),
                                                 1 if dimg_acc is not None else 0, _ptr(dflow), n, c, h, w, s2d,
                                                 _stream()), 'tg_backward_warp_s2d_bwd')
        return dimg, dflow
    if dimg_acc is not None:
        _chk(dimg_acc, 'dimg_acc')
        assert dimg_acc.shape == x.shape
        dflow = None
        if need_flow:
            dflow = _chk(dflow_out, 'dflow_out') if dflow_out is not None else torch.empty_like(flow)
        L.check(L.lib().tg_backward_warp_bwd_acc(x.data_ptr(), flow.data_ptr(), dy.data_ptr(), dimg_acc.data_ptr(),
                                                 _ptr(dflow), n, c, h, w, _stream()), 'tg_backward_warp_bwd_acc')
        return dimg_acc, dflow
    dimg = torch.empty_like(x) if need_img else None
    if need_flow and dflow_out is not None:
        dflow = _chk(dflow_out, 'dflow_out')
        assert dflow.shape == flow.shape
    else:
        dflow = torch.empty_like(flow) if need_flow else None
    L.check(L.lib().tg_backward_warp_bwd(x.data_ptr(), flow.data_ptr(), dy.data_ptr(), _ptr(dimg),
                                         _ptr(dflow), n, c, h, w, _stream()), 'tg_backward_warp_bwd')
    return dimg, dflow


def depth_to_space(x, scale, act_y=None, act=ACT_NONE):
    """act_y / act: the result is the gradient of the activation output act_y and leaves multiplied by
    act'(.) (tg_depth_to_space_act_bwd); returns (y, fused) then."""
    _chk(x, 'x')
    n, cs, h, w = x.shape
    c = cs // (scale * scale)
    y = torch.empty(n, c, h * scale, w * scale, dtype=torch.float32, device=x.device)
    if act_y is not None:
        _chk(act_y, 'act_y')
        lib = L.lib()
        if act in (ACT_RELU, ACT_LRELU02) and act_y.shape == y.shape and \
                lib.tg_depth_to_space_act_bwd_supported(x.data_ptr(), act_y.data_ptr(), y.data_ptr(), w, scale):
            L.check(lib.tg_depth_to_space_act_bwd(x.data_ptr(), act_y.data_ptr(), act, y.data_ptr(), n, c, h, w,
                                                  scale, _stream()), 'tg_depth_to_space_act_bwd')
            return y, True
        L.check(lib.tg_depth_to_space(x.data_ptr(), y.data_ptr(), n, c, h, w, scale, _stream()), 'tg_depth_to_space')
        return y, False
    L.check(L.lib().tg_depth_to_space(x.data_ptr(), y.data_ptr(), n, c, h, w, scale, _stream()),
            'tg_depth_to_space')
    return y


def conv4x4s2_supported(n, ci, co, h, w):
    return bool(L.lib().tg_conv4x4s2_supported(n, ci, co, h, w))


def pack_conv4x4s2(w):
    """OIHW (co, ci, 4, 4) -> (forward pack, data-gradient pack) of tg_conv4x4s2_fwd / _dgrad."""
    _chk(w, 'w')
    co, ci = w.shape[:2]
    lib = L.lib()
    nfl = lib.tg_conv4x4s2_packed_floats(ci, co)
    pf = torch.empty(nfl, dtype=torch.float32, device=w.device)
    pd = torch.empty(nfl, dtype=torch.float32, device=w.device)
    L.check(lib.tg_conv4x4s2_pack(w.data_ptr(), pf.data_ptr(), pd.data_ptr(), ci, co, _stream()), 'tg_conv4x4s2_pack')
    return pf, pd


def conv4x4s2(x, w_fwd, co, out=None):
    """nn.Conv2d(ci, co, 4, 2, 1, bias=False)(x) (tecogan_nets.py:322-340) on the direct kernel."""
    _chk(x, 'x')
    n, ci, h, w = x.shape
    y = out if out is not None else torch.empty(n, co, h // 2, w // 2, dtype=torch.float32, device=x.device)
    lib = L.lib()
    nws = lib.tg_conv4x4s2_workspace_floats(n, ci, co, h, w, 0)      # partial sums of the small-map form
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    L.check(lib.tg_conv4x4s2_fwd(x.data_ptr(), w_fwd.data_ptr(), y.data_ptr(), _ptr(ws), n, ci, co, h, w, _stream()),
            'tg_conv4x4s2_fwd')
    return y


def conv4x4s2_dgrad(g, w_dgrad, ci, act_y=None, act=ACT_NONE):
    """Gradient of conv4x4s2's input from the gradient g of its output; act_y / act: multiplied by act'(act_y)
    on the way out (act_y = the conv's input when that is an activation output)."""
    _chk(g, 'g')
    n, co, oh, ow = g.shape
    dx = torch.empty(n, ci, 2 * oh, 2 * ow, dtype=torch.float32, device=g.device)
    if act_y is not None:
        _chk(act_y, 'act_y')
        assert act_y.shape == dx.shape
    lib = L.lib()
    nws = lib.tg_conv4x4s2_workspace_floats(n, ci, co, 2 * oh, 2 * ow, 1)
    ws = torch.empty(nws, dtype=torch.float32, device=g.device) if nws else None
    L.check(lib.tg_conv4x4s2_dgrad(g.data_ptr(), w_dgrad.data_ptr(), _ptr(act_y), act, dx.data_ptr(), _ptr(ws), n, ci, co,
                                   2 * oh, 2 * ow, _stream()), 'tg_conv4x4s2_dgrad')
    return dx


def charbonnier(x, y, loss_accum, loss_scale, grad_scale=None, eps=1e-6):
    """loss_accum[0] += loss_scale * sum sqrt((x-y)^2+eps); returns d loss / dx (scaled) or None."""
    _chk(x, 'x'); _chk(y, 'y')
    dx = torch.empty_like(x) if grad_scale is not None else None
    L.check(L.lib().tg_charbonnier(x.data_ptr(), y.data_ptr(), x.numel(), float(eps),
                                   float(loss_scale), _ptr(loss_accum),
                                   float(grad_scale or 0.0), _ptr(dx), _stream()), 'tg_charbonnier')
    return dx


LOSS_L1, LOSS_MSE = 1, 2


def pixel_loss(x, y, mode, loss_accum, loss_scale, grad_scale=None):
    """L1 / MSE: loss_accum[0] += loss_scale * sum v(x-y); returns scaled d loss / dx or None."""
    _chk(x, 'x'); _chk(y, 'y')
    if x.shape != y.shape:
        raise L.TecoganHipError(f'pixel_loss: {tuple(x.shape)} vs {tuple(y.shape)}')
    dx = torch.empty_like(x) if grad_scale is not None else None
    L.check(L.lib().tg_pixel_loss(x.data_ptr(), y.data_ptr(), x.numel(), int(mode), float(loss_scale),
                                  _ptr(loss_accum), float(grad_scale or 0.0), _ptr(dx), _stream()),
            'tg_pixel_loss')
    return dx


def channel_norm(x, mean, std):
    """(x - mean[c]) / std[c] over (n,c,h,w); mean None = 0."""
    _chk(x, 'x'); _chk(std, 'std')
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    L.check(L.lib().tg_channel_norm(x.data_ptr(), _ptr(mean), std.data_ptr(), y.data_ptr(), n, c, h * w,
                                    _stream()), 'tg_channel_norm')
    return y


def cosine_loss(a, b, loss_accum, loss_scale, grad_scale=None, eps=1e-8):
    """loss_accum[0] += loss_scale * sum_pixels (1 - cos_c(a, b)); returns the scaled gradient
    w.r.t. a or None."""
    _chk(a, 'a'); _chk(b, 'b')
    if a.shape != b.shape:
        raise L.TecoganHipError(f'cosine_loss: {tuple(a.shape)} vs {tuple(b.shape)}')
    n, c, h, w = a.shape
    da = torch.empty_like(a) if grad_scale is not None else None
    L.check(L.lib().tg_cosine_loss(a.data_ptr(), b.data_ptr(), n, c, h * w, float(eps),
                                   float(loss_scale), _ptr(loss_accum), float(grad_scale or 0.0),
                                   _ptr(da), _stream()), 'tg_cosine_loss')
    return da


def bce_logits(x, target, stats3, scale, grad_scale=None, dx_out=None, lsgan=False):
    """VanillaGANLoss (BCE with logits) or, lsgan=True, LSGANLoss (MSE) against a constant target."""
    _chk(x, 'x')
    dx = (dx_out if dx_out is not None else torch.empty_like(x)) if grad_scale is not None else None
    fn = L.lib().tg_lsgan_loss if lsgan else L.lib().tg_bce_logits
    L.check(fn(x.data_ptr(), x.numel(), float(target), float(scale), _ptr(stats3), float(grad_scale or 0.0),
               _ptr(dx), _stream()), 'tg_lsgan_loss' if lsgan else 'tg_bce_logits')
    return dx


def adam_step(p, g, m, v, lr, betas, eps, weight_decay, step, skip=None):
    """torch.optim.Adam step in place; `skip`: a device float -- non-zero turns the step into a no-op
    (the fault slot of the flat gradient buffer, see tg_adam_step_guarded)."""
    if skip is not None:
        L.check(L.lib().tg_adam_step_guarded(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                             float(lr), float(betas[0]), float(betas[1]), float(eps),
                                             float(weight_decay), int(step), skip.data_ptr(), _stream()),
                'tg_adam_step_guarded')
        return
    L.check(L.lib().tg_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                 float(lr), float(betas[0]), float(betas[1]), float(eps),
                                 float(weight_decay), int(step), _stream()), 'tg_adam_step')


def fault_to_slot(err_pinned, slot):
    """slot += 1 if the pinned int32 fault counter is non-zero (one thread; device-side read of host memory)."""
    L.check(L.lib().tg_fault_to_slot(err_pinned.data_ptr(), slot.data_ptr(), _stream()), 'tg_fault_to_slot')


def axpy_(y, x, a=1.0):
    _chk(y, 'y'); _chk(x, 'x')
    L.check(L.lib().tg_axpy(y.data_ptr(), x.data_ptr(), float(a), y.numel(), _stream()), 'tg_axpy')
    return y


def div_scalar_(y, d, x=None):
    """y = (x or y) / d, elementwise IEEE division."""
    _chk(y, 'y')
    src = y if x is None else _chk(x, 'x')
    L.check(L.lib().tg_div_scalar(y.data_ptr(), src.data_ptr(), float(d), y.numel(), _stream()), 'tg_div_scalar')
    return y


def bn_lrelu_train_fwd(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, slope=0.2, out=None):
    _chk(x, 'x')
    n, c, h, w = x.shape
    y = out if out is not None else torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_bn_lrelu_train_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          _ptr(running_mean), _ptr(running_var), float(momentum),
                                          float(eps), float(slope), y.data_ptr(), mean.data_ptr(),
                                          invstd.data_ptr(), n, c, h * w, _stream()),
            'tg_bn_lrelu_train_fwd')
    return y, mean, invstd


def bn_lrelu_train_bwd(x, y, dy, gamma, mean, invstd, dgamma=None, dbeta=None, need_dx=True,
                       slope=0.2, dx_out=None):
    n, c, h, w = x.shape
    dx = (dx_out if dx_out is not None else torch.empty_like(x)) if need_dx else None
    scratch = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_bn_lrelu_train_bwd(x.data_ptr(), y.data_ptr(), dy.data_ptr(), gamma.data_ptr(),
                                          mean.data_ptr(), invstd.data_ptr(), float(slope), _ptr(dx),
                                          _ptr(dgamma), _ptr(dbeta), 1, scratch.data_ptr(), n, c,
                                          h * w, _stream()), 'tg_bn_lrelu_train_bwd')
    return dx


def linear1_fwd(x, w, b):
    _chk(x, 'x')
    rows, k = x.shape
    y = torch.empty(rows, 1, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_linear1_fwd(x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), rows, k,
                                   _stream()), 'tg_linear1_fwd')
    return y


def linear1_bwd(x, w, dy, dw=None, db=None, need_dx=True):
    rows, k = x.shape
    dx = torch.empty_like(x) if need_dx else None
    L.check(L.lib().tg_linear1_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), _ptr(dx), _ptr(dw),
                                   _ptr(db), rows, k, 1, _stream()), 'tg_linear1_bwd')
    return dx


_BD_KERNELS = {}


def downsample_bd(x, kernel2d, scale, pad):
    """x (n,c,h,w) -> blurred + decimated (n,c,oh,ow); kernel2d: numpy (k,k) float32."""
    _chk(x, 'x')
    n, c, h, w = x.shape
    ks = int(kernel2d.shape[0])
    key = (str(x.device), ks, float(kernel2d[ks // 2, ks // 2]))
    kd = _BD_KERNELS.get(key)
    if kd is None:
        kd = _BD_KERNELS[key] = torch.from_numpy(kernel2d.copy()).to(x.device).contiguous()
    oh = (h - 1) // scale + 1 if pad else (h - ks) // scale + 1
    ow = (w - 1) // scale + 1 if pad else (w - ks) // scale + 1
    y = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_downsample_bd(x.data_ptr(), kd.data_ptr(), y.data_ptr(), n * c, h, w, ks,
                                     scale, 1 if pad else 0, _stream()), 'tg_downsample_bd')
    return y


# ---- SyncBatchNorm (+LeakyReLU): statistics exchanged over ranks between the halves ----
def sync_bn_lrelu_train_fwd(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5,
                            slope=0.2):
    """Same contract as bn_lrelu_train_fwd with statistics over the GLOBAL batch: every rank
    computes (mean, centred M2) of its slice, ONE all-gather of 2c floats per layer, merged
    with Chan's formula in rank order (equal per-rank batch sizes, as DistributedSampler
    guarantees) -- the formulation torch's SyncBatchNorm uses, with no E[x^2] - mean^2
    cancellation.  With one rank the result is bit-identical to bn_lrelu_train_fwd."""
    from .utils import dist_utils
    _chk(x, 'x')
    n, c, h, w = x.shape
    lib = L.lib()
    local = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    L.check(lib.tg_bn_local_stats(x.data_ptr(), local.data_ptr(), n, c, h * w, _stream()),
            'tg_bn_local_stats')
    gathered = dist_utils.all_gather_flat(local).contiguous()
    world = gathered.shape[0]
    count = float(n * h * w * world)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    L.check(lib.tg_bn_merge_stats(gathered.data_ptr(), world, float(n * h * w), float(eps),
                                  float(momentum), mean.data_ptr(), invstd.data_ptr(),
                                  _ptr(running_mean), _ptr(running_var), c, _stream()),
            'tg_bn_merge_stats')
    y = torch.empty_like(x)
    L.check(lib.tg_bn_lrelu_apply(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                  beta.data_ptr(), float(slope), y.data_ptr(), n, c, h * w,
                                  _stream()), 'tg_bn_lrelu_apply')
    return y, mean, invstd, count


def sync_bn_lrelu_train_bwd(x, y, dy, gamma, mean, invstd, count, dgamma=None, dbeta=None,
                            need_dx=True, slope=0.2):
    """dgamma/dbeta receive the LOCAL sums (DDP averages parameter gradients afterwards, as
    SyncBatchNorm does); dx uses the global sums."""
    n, c, h, w = x.shape
    lib = L.lib()
    sums = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    L.check(lib.tg_bn_lrelu_bwd_reduce(x.data_ptr(), y.data_ptr(), dy.data_ptr(), mean.data_ptr(),
                                       invstd.data_ptr(), float(slope), sums.data_ptr(), n, c, h * w,
                                       _stream()), 'tg_bn_lrelu_bwd_reduce')
    if dgamma is not None:
        axpy_(dgamma, sums[c:], 1.0)           # contiguous slices of the packed vector
        axpy_(dbeta, sums[:c], 1.0)
    from .utils import dist_utils
    dist_utils.all_reduce_sum_(sums)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        L.check(lib.tg_bn_lrelu_bwd_apply(x.data_ptr(), y.data_ptr(), dy.data_ptr(), mean.data_ptr(),
                                          invstd.data_ptr(), gamma.data_ptr(), sums.data_ptr(),
                                          float(slope), 1.0 / count, dx.data_ptr(), n, c, h * w,
                                          _stream()), 'tg_bn_lrelu_bwd_apply')
    return dx


def sync_bn_lrelu_train_fwd_groups(x, groups, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5,
                                   slope=0.2):
    """SyncBatchNorm + LeakyReLU of `groups` INDEPENDENT batches stacked along n (the critic's real and fake
    pass run as one pair batch, vsrgan_model.py:137-153): statistics per group, ONE all-gather of
    groups x 2c floats per layer instead of one per pass, running statistics updated group after group (the
    order of the reference's separate passes).  Returns y and per-group (mean, invstd, count)."""
    from .utils import dist_utils
    _chk(x, 'x')
    n, c, h, w = x.shape
    per = n // groups
    lib = L.lib()
    local = torch.empty(groups * 2 * c, dtype=torch.float32, device=x.device)
    for g in range(groups):
        L.check(lib.tg_bn_local_stats(x[g * per:(g + 1) * per].data_ptr(), local[g * 2 * c:].data_ptr(), per, c, h * w,
                                      _stream()), 'tg_bn_local_stats')
    gathered = dist_utils.all_gather_flat(local)                     # (world, groups * 2c)
    world = gathered.shape[0]
    count = float(per * h * w * world)
    y = torch.empty_like(x)
    stats = []
    for g in range(groups):
        part = gathered[:, g * 2 * c:(g + 1) * 2 * c].contiguous()
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        L.check(lib.tg_bn_merge_stats(part.data_ptr(), world, float(per * h * w), float(eps), float(momentum),
                                      mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var), c,
                                      _stream()), 'tg_bn_merge_stats')
        L.check(lib.tg_bn_lrelu_apply(x[g * per:(g + 1) * per].data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                      gamma.data_ptr(), beta.data_ptr(), float(slope), y[g * per:(g + 1) * per].data_ptr(),
                                      per, c, h * w, _stream()), 'tg_bn_lrelu_apply')
        stats.append((mean, invstd, count))
    return y, stats


def sync_bn_lrelu_train_bwd_groups(x, y, dy, groups, gamma, stats, dgamma=None, dbeta=None, need_dx=True, slope=0.2):
    """Backward of sync_bn_lrelu_train_fwd_groups: ONE all-reduce of the groups x 2c sums per layer."""
    from .utils import dist_utils
    n, c, h, w = x.shape
    per = n // groups
    lib = L.lib()
    sums = torch.empty(groups * 2 * c, dtype=torch.float32, device=x.device)
    for g in range(groups):
        sl = slice(g * per, (g + 1) * per)
        L.check(lib.tg_bn_lrelu_bwd_reduce(x[sl].data_ptr(), y[sl].data_ptr(), dy[sl].data_ptr(), stats[g][0].data_ptr(),
                                           stats[g][1].data_ptr(), float(slope), sums[g * 2 * c:].data_ptr(), per, c,
                                           h * w, _stream()), 'tg_bn_lrelu_bwd_reduce')
        if dgamma is not None:
            axpy_(dgamma, sums[g * 2 * c + c:(g + 1) * 2 * c], 1.0)
            axpy_(dbeta, sums[g * 2 * c:g * 2 * c + c], 1.0)
    dist_utils.all_reduce_sum_(sums)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        for g in range(groups):
            sl = slice(g * per, (g + 1) * per)
            L.check(lib.tg_bn_lrelu_bwd_apply(x[sl].data_ptr(), y[sl].data_ptr(), dy[sl].data_ptr(), stats[g][0].data_ptr(),
                                              stats[g][1].data_ptr(), gamma.data_ptr(), sums[g * 2 * c:].data_ptr(),
                                              float(slope), 1.0 / stats[g][2], dx[sl].data_ptr(), per, c, h * w,
                                              _stream()), 'tg_bn_lrelu_bwd_apply')
    return dx


# ---- data movement of the training step (tg_assemble.hip) ---------------------------------
def time_gather(x, idx):
    """x (n, t, ...) -> (n, len(idx), ...) with out[:, k] = x[:, idx[k]] (one launch)."""
    import ctypes
    _chk(x, 'x')
    n, t = x.shape[:2]
    inner = x[0, 0].numel()
    out = torch.empty((n, len(idx)) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    arr = (ctypes.c_int * len(idx))(*[int(i) for i in idx])
    L.check(L.lib().tg_time_gather(x.data_ptr(), out.data_ptr(), arr, n, t, len(idx), inner, _stream()),
            'tg_time_gather')
    return out


def pingpong(x):
    """cat([x, x.flip(1)[:, 1:]], 1) (vsrgan_model.py:112-119)."""
    t = x.shape[1]
    return time_gather(x, list(range(t)) + list(range(t - 2, -1, -1)))


def pingpong_grad(g, te):
    """Ping-pong loss gradient g (n, te-1, ...) -> (n, 2*te-1, ...): +g | 0 | -flip(g)."""
    _chk(g, 'g')
    n = g.shape[0]
    inner = g[0, 0].numel()
    out = torch.empty((n, 2 * te - 1) + tuple(g.shape[2:]), dtype=torch.float32, device=g.device)
    L.check(L.lib().tg_pingpong_grad(g.data_ptr(), out.data_ptr(), n, te, inner, _stream()), 'tg_pingpong_grad')
    return out


def d_assemble_fwd(data, warped, cond, t, pad, crop, out=None):
    """Discriminator input (n*t/3, 9c, h, w) from data (n, T, c, h, w), warped (n*t, c, h, w),
    cond (n, T', c, h, w); `out`: a contiguous (n*t/3, 9c, h, w) destination (half of a pair batch)."""
    _chk(data, 'data'); _chk(warped, 'warped'); _chk(cond, 'cond')
    n, t_data, c, h, w = data.shape
    x = out if out is not None else torch.empty(n * t // 3, 9 * c, h, w, dtype=torch.float32, device=data.device)
    L.check(L.lib().tg_d_assemble_fwd(data.data_ptr(), t_data, warped.data_ptr(), cond.data_ptr(),
                                      cond.shape[1], x.data_ptr(), n, t, c, h, w, pad, crop, _stream()),
            'tg_d_assemble_fwd')
    return x


def d_assemble_bwd(g, n, t, t_data, c, pad, crop):
    """-> (g_data (n, t_data, c, h, w), g_warped (n*t, c, h, w))."""
    _chk(g, 'g')
    h, w = g.shape[2:]
    g_data = torch.empty(n, t_data, c, h, w, dtype=torch.float32, device=g.device)
    g_warped = torch.empty(n * t, c, h, w, dtype=torch.float32, device=g.device)
    L.check(L.lib().tg_d_assemble_bwd(g.data_ptr(), g_data.data_ptr(), t_data, g_warped.data_ptr(), n, t, c,
                                      h, w, pad, crop, _stream()), 'tg_d_assemble_bwd')
    return g_data, g_warped


def transpose01(x):
    """(a, b, ...) -> (b, a, ...) contiguous (clip-major <-> frame-major), one launch."""
    _chk(x, 'x')
    a, b = x.shape[:2]
    y = torch.empty((b, a) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_transpose01(x.data_ptr(), y.data_ptr(), a, b, x[0, 0].numel(), _stream()),
            'tg_transpose01')
    return y


def stack_time(frames):
    """torch.stack(frames, dim=1) of k <= 64 contiguous (n, ...) tensors, one launch."""
    import ctypes
    for f in frames:
        _chk(f, 'frame')
    n = frames[0].shape[0]
    y = torch.empty((n, len(frames)) + tuple(frames[0].shape[1:]), dtype=torch.float32,
                    device=frames[0].device)
    arr = (ctypes.c_void_p * len(frames))(*[f.data_ptr() for f in frames])
    L.check(L.lib().tg_stack_time(arr, len(frames), y.data_ptr(), n, frames[0][0].numel(), _stream()),
            'tg_stack_time')
    return y


def index_gather(src, idx, out=None, accumulate=False):
    """out.flat[i] (+)= src.flat[idx[i]] (0 where idx[i] is out of range); idx int64 on device."""
    _chk(src, 'src'); _chk(idx, 'idx', torch.int64)
    if out is None:
        assert not accumulate
        out = torch.empty(idx.numel(), dtype=torch.float32, device=src.device)
    else:
        _chk(out, 'out')
        assert out.numel() == idx.numel()
    L.check(L.lib().tg_index_gather(src.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(),
                                    src.numel(), int(accumulate), _stream()), 'tg_index_gather')
    return out


# ---- Winograd F(2x2, 3x3) form of conv3x3 (tg_conv3x3_wino.hip) ----------------------------
def pack_conv3x3_wino(w, transposed=0):
    """OIHW weights -> transformed, lane-ordered U (see tg_pack_conv3x3_wino)."""
    _chk(w, 'w')
    co, ci = (w.shape[0], w.shape[1]) if transposed == 0 else (w.shape[1], w.shape[0])
    nflt = L.lib().tg_conv3x3_wino_packed_floats(ci, co)
    out = torch.empty(nflt, dtype=torch.float32, device=w.device)
    L.check(L.lib().tg_pack_conv3x3_wino(w.data_ptr(), out.data_ptr(), ci, co, transposed, _stream()),
            'tg_pack_conv3x3_wino')
    return out


def conv3x3_wino(x, u_packed, bias, cin, cout, act=ACT_NONE, x2=None, res=None, mask=None, out=None):
    _chk(x, 'x'); _chk(u_packed, 'u_packed')
    n, c1, h, w = x.shape
    if x2 is not None:
        _chk(x2, 'x2')
        assert c1 + x2.shape[1] == cin
    else:
        assert c1 == cin
    y = out if out is not None else torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    L.check(L.lib().tg_conv3x3_wino_fwd(
        x.data_ptr(), x.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0,
        u_packed.data_ptr(), _ptr(bias), _ptr(res), res.stride(0) if res is not None else 0,
        _ptr(mask), mask.stride(0) if mask is not None else 0, y.data_ptr(), y.stride(0),
        n, cin, cout, h, w, act, _stream()), 'tg_conv3x3_wino_fwd')
    return y


class RowChain:
    """tg_conv3x3_chain: dependent 3x3 layers of small frames in one persistent launch.  `layers` is a
    list of dicts (x, w (OIHW weight tensor), bias=None, act=ACT_NONE, x2=None, res=None, mask=None, y);
    the weights are packed here in the layout the launcher's choice for this shape needs."""

    def __init__(self, layers, n, h, w):
        lib = L.lib()
        self.n, self.h, self.w = n, h, w
        self.parts = lib.tg_conv3x3_chain_supported(n, h, w, 64)
        if self.parts == 0:
            raise L.TecoganHipError(f'RowChain: {n}x{h}x{w} cannot run as a chained launch on this device')
        self.layout = 16 if self.parts == 4 else 64
        self.keep = [layers]
        self.arr = (L.ChainLayer * len(layers))()
        for i, d in enumerate(layers):
            a = self.arr[i]
            wt = d['w']
            pk = pack_conv3x3_m16(wt) if self.layout == 16 else pack_conv3x3(wt, ocb=64)[0]
            self.keep.append(pk)
            x, x2, res, mask, y = d['x'], d.get('x2'), d.get('res'), d.get('mask'), d['y']
            a.x, a.x2, a.w_packed, a.bias = x.data_ptr(), _ptr(x2), pk.data_ptr(), _ptr(d.get('bias'))
            a.res, a.relu_mask, a.y = _ptr(res), _ptr(mask), y.data_ptr()
            a.x_nstride, a.y_nstride = x.stride(0), y.stride(0)
            a.x2_nstride = x2.stride(0) if x2 is not None else 0
            a.res_nstride = res.stride(0) if res is not None else 0
            a.mask_nstride = mask.stride(0) if mask is not None else 0
            a.c1, a.cin, a.cout, a.act = x.shape[1], wt.shape[1], wt.shape[0], d.get('act', ACT_NONE)
        self.nl = len(layers)
        self.flags = torch.zeros(lib.tg_conv3x3_chain_flag_ints(self.nl, n, h, w) + 16, dtype=torch.int32,
                                 device=layers[0]['x'].device)
        self.err = self.flags[-16:]
        self.epoch = 0

    def run(self, poll_limit=1 << 21):
        self.epoch += 1
        L.check(L.lib().tg_conv3x3_chain(self.arr, self.nl, self.n, self.h, self.w, self.layout, self.flags.data_ptr(),
                                         self.err.data_ptr(), self.epoch, poll_limit, _stream()), 'tg_conv3x3_chain')

    def faults(self):
        return int(self.err[0].item())


class WinoChain:
    """tg_conv3x3_wino_chain: dependent 3x3 layers in one launch.  `layers` is a list of dicts
    (x, u, bias, cin, act, x2=None, res=None, y) of device tensors; the flag buffer and the epoch
    counter live here."""

    def __init__(self, layers, n, cout, h, w):
        import ctypes
        self.n, self.cout, self.h, self.w = n, cout, h, w
        self.keep = layers
        self.arr = (L.WinoLayer * len(layers))()
        for i, d in enumerate(layers):
            a = self.arr[i]
            x, x2, res, y = d['x'], d.get('x2'), d.get('res'), d['y']
            a.x, a.x2, a.u_packed = x.data_ptr(), _ptr(x2), d['u'].data_ptr()
            a.bias, a.res, a.y = _ptr(d.get('bias')), _ptr(res), y.data_ptr()
            a.x_nstride, a.y_nstride = x.stride(0), y.stride(0)
            a.x2_nstride = x2.stride(0) if x2 is not None else 0
            a.res_nstride = res.stride(0) if res is not None else 0
            a.c1, a.cin, a.act = x.shape[1], d['cin'], d.get('act', ACT_NONE)
        nints = L.lib().tg_conv3x3_wino_chain_flag_ints(len(layers), n, h, w)
        self.flags = torch.zeros(nints, dtype=torch.int32, device=layers[0]['x'].device)
        self.epoch = 0

    def run(self):
        self.epoch += 1
        L.check(L.lib().tg_conv3x3_wino_chain(self.arr, len(self.keep), self.n, self.cout, self.h, self.w,
                                              self.flags.data_ptr(), self.epoch, _stream()),
                'tg_conv3x3_wino_chain')

    def bailouts(self):
        return int(self.flags[-16].item())


def pack_wres_convt(w):
    """nn.ConvTranspose2d(64, 64, 3, 2, 1, 1).weight (cin, cout, 3, 3) -> the A operands of the resident launch's
    transposed-conv tail (tg_conv3x3_wino_resident_ct_pack)."""
    _chk(w, 'w')
    if tuple(w.shape) != (64, 64, 3, 3):
        raise L.TecoganHipError(f'pack_wres_convt: weight {tuple(w.shape)}, expected (64, 64, 3, 3)')
    lib = L.lib()
    out = torch.empty(lib.tg_conv3x3_wino_resident_ct_floats(), dtype=torch.float32, device=w.device)
    L.check(lib.tg_conv3x3_wino_resident_ct_pack(w.data_ptr(), out.data_ptr(), _stream()), 'tg_conv3x3_wino_resident_ct_pack')
    return out


class WinoResident(WinoChain):
    """tg_conv3x3_wino_resident: the same dependent layers of ONE frame on persistent, LDS-resident
    workgroups (tg_conv3x3_wino_res.hip).  Only layers[0]['x'] / ['x2'] are read and only
    layers[-1]['y'] is written."""

    def __init__(self, layers, cout, h, w):
        WinoChain.__init__(self, layers, 1, cout, h, w)
        nbytes = L.lib().tg_conv3x3_wino_resident_ws_bytes(h, w)
        self.ws = torch.zeros(nbytes // 4, dtype=torch.int32, device=layers[0]['x'].device)

    @staticmethod
    def supported(cout, h, w, n=1):
        return bool(L.lib().tg_conv3x3_wino_resident_supported(n, cout, h, w))

    def run(self, convt=None):
        """convt: None, or dict(u=pack_wres_convt(weight), bias=, y=(64, 2h, 2w) output, act=): SRNet's first
        ConvTranspose2d + act as the launch's tail (tg_conv3x3_wino_resident_ct); layers[-1]['y'] is then not written."""
        self.epoch += 1
        if convt is None:
            L.check(L.lib().tg_conv3x3_wino_resident(self.arr, len(self.keep), self.cout, self.h, self.w,
                                                     self.ws.data_ptr(), self.epoch, _stream()),
                    'tg_conv3x3_wino_resident')
            return
        import ctypes as C
        ct = L.WresConvT(convt['u'].data_ptr(), convt['bias'].data_ptr(), convt['y'].data_ptr(), convt.get('act', ACT_RELU))
        L.check(L.lib().tg_conv3x3_wino_resident_ct(self.arr, len(self.keep), self.cout, self.h, self.w,
                                                    self.ws.data_ptr(), self.epoch, C.byref(ct), _stream()),
                'tg_conv3x3_wino_resident_ct')

    def bailouts(self):
        return int(self.ws[-64].item())
