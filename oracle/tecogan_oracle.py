"""CPU oracle for the TecoGAN / FRVSR frame-recurrent hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this file.  The product package (tecogan-pytorch_amd/) never imports
anything under oracle/ and fails loudly when its HIP library is missing.

What this is: a from-the-formulas restatement (own code, fp32, torch-CPU /
numpy) of the reference's hot path, written against SURVEY.md section 8a.
Every function cites the reference file:line it follows (paths relative to
the upstream repo root, skycrapers/TecoGAN-PyTorch @ v1).

Parity pinning: the reference ships NO tests and NO golden vectors for this
path, so the oracle is pinned by fixtures generated in
the authoring container by importing the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz) and checked in
tests/test_oracle_vs_golden.py.  Third-party arithmetic below the path is
PyTorch ATen CPU (torch 2.10.0+rocm7.0 here; the reference pins only
"PyTorch >= 1.4.0"): conv2d / conv_transpose2d / max_pool2d are called here
through torch.nn.functional as the reference does, while every op the
reference *composes itself* (warp, space-to-depth, bicubic, bilinear x s,
reflect pad, quantise, BD blur, PSNR, losses, discriminator input assembly)
is restated from its formula.

Weights travel as a flat dict keyed exactly like the reference's
state_dict ("State-dict key layout").
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

F32 = torch.float32


# --------------------------------------------------------------------------
# G4  upsampling   (codes/utils/net_utils.py:85-156)
# --------------------------------------------------------------------------
def bicubic_kernels(f, a=-0.75):
    """K[d, k], d = sub-pixel phase 0..f-1, k = tap 0..3.

    net_utils.py:113-127: K[d] = cubic @ [1, s, s^2, s^3], s = d / f, with the
    Keys matrix for a = -0.75; all fp32.
    """
    cubic = np.array([[0, a, -2 * a, a],
                      [1, 0, -(a + 3), a + 2],
                      [0, -a, (2 * a + 3), -(a + 2)],
                      [0, 0, a, -a]], dtype=np.float32)
    ks = []
    for d in range(f):
        s = 1.0 * d / f
        v = torch.tensor([1, s, s ** 2, s ** 3], dtype=F32)
        ks.append(torch.matmul(torch.from_numpy(cubic), v))
    return torch.stack(ks)  # (f, 4) fp32


def bicubic_upsample(x, f, kernels=None):
    """net_utils.py:133-156.  Top-left aligned (no half-pixel shift):

        out[f*i+dy, f*j+dx] = sum_q K[dx,q] * ( sum_p K[dy,p] * x[cl(i-1+p), cl(j-1+q)] )

    replicate padding 1 before / 2 after (net_utils.py:141), vertical pass
    first (144-146) then horizontal (149-151).
    """
    if kernels is None:
        kernels = bicubic_kernels(f)
    n, c, h, w = x.shape
    K = kernels.to(x.dtype)
    ri = [torch.clamp(torch.arange(h) - 1 + p, 0, h - 1) for p in range(4)]
    ci = [torch.clamp(torch.arange(w) - 1 + q, 0, w - 1) for q in range(4)]
    # vertical: (n, c, h, f, w)
    v = x.new_zeros(n, c, h, f, w)
    for p in range(4):
        v = v + K[:, p].view(1, 1, 1, f, 1) * x[:, :, ri[p], :].unsqueeze(3)
    v = v.reshape(n, c, h * f, w)
    o = x.new_zeros(n, c, h * f, w, f)
    for q in range(4):
        o = o + K[:, q].view(1, 1, 1, 1, f) * v[:, :, :, ci[q]].unsqueeze(4)
    return o.reshape(n, c, h * f, w * f)


def _bilinear_axis(in_size, s):
    """Source index / lambda for F.interpolate(bilinear, align_corners=False)
    with an integer scale factor s (net_utils.py:86-89, tecogan_nets.py:74-79):
    src = (dst + .5)/s - .5 clamped at 0; i1 = min(i0 + 1, in_size - 1)."""
    dst = torch.arange(in_size * s, dtype=F32)
    src = (dst + 0.5) * (1.0 / s) - 0.5
    src = torch.clamp(src, min=0.0)
    i0 = src.floor().long()
    i1 = torch.clamp(i0 + 1, max=in_size - 1)
    l1 = src - i0.to(F32)
    l0 = 1.0 - l1
    return i0, i1, l0, l1


def bilinear_upsample(x, s):
    n, c, h, w = x.shape
    y0, y1, ly0, ly1 = _bilinear_axis(h, s)
    x0, x1, lx0, lx1 = _bilinear_axis(w, s)
    ly0, ly1 = ly0.view(1, 1, -1, 1), ly1.view(1, 1, -1, 1)
    lx0, lx1 = lx0.view(1, 1, 1, -1), lx1.view(1, 1, 1, -1)
    top, bot = x[:, :, y0, :], x[:, :, y1, :]
    return (ly0 * (lx0 * top[:, :, :, x0] + lx1 * top[:, :, :, x1]) +
            ly1 * (lx0 * bot[:, :, :, x0] + lx1 * bot[:, :, :, x1]))


def upsample(x, s, degradation):
    """get_upsampling_func, net_utils.py:85-97."""
    if degradation == 'BD':
        return bicubic_upsample(x, s)
    if degradation == 'BI':
        return bilinear_upsample(x, s)
    raise ValueError(f'Unrecognized degradation type: {degradation}')


# --------------------------------------------------------------------------
# G5  backward warp   (codes/utils/net_utils.py:50-82)
# --------------------------------------------------------------------------
def linspace_m1_p1(n):
    """fp32 torch.linspace(-1, 1, n) as the ATen CPU kernel evaluates it
    (net_utils.py:62-63): step = fp32(2/(n-1)); first half fma(step, i, -1),
    second half fma(-step, n-1-i, +1) -- one rounding each (the fused
    multiply-add is emulated exactly in float64: a 24-bit x 11-bit product
    plus +-1 fits in 53 bits)."""
    if n == 1:
        return np.array([-1.0], dtype=np.float32)
    step = np.float64(np.float32(np.float32(2.0) / np.float32(n - 1)))
    i = np.arange(n, dtype=np.float64)
    lo = (step * i - 1.0).astype(np.float32)
    hi = (1.0 - step * (n - 1 - i)).astype(np.float32)
    return np.where(np.arange(n) < n // 2, lo, hi).astype(np.float32)


def backward_warp(x, flow):
    """out[n,c,y,x] = bilinear sample of x at (x + flow[n,0], y + flow[n,1]),
    coordinates clamped to the image (padding_mode='border',
    align_corners=True), net_utils.py:50-82.

    The reference goes through normalised coordinates; that round trip is kept
    (fp32) so the sampling positions round identically:
        g  = linspace(-1,1,W)[x] + flow_x / ((W-1)/2)          (62-72)
        px = (g + 1) * ((W-1)/2), clipped to [0, W-1]           (grid_sample)
    """
    n, c, h, w = x.shape
    gx = torch.from_numpy(linspace_m1_p1(w)).view(1, 1, w)
    gy = torch.from_numpy(linspace_m1_p1(h)).view(1, h, 1)
    gx = gx + flow[:, 0] / ((w - 1.0) / 2.0)
    gy = gy + flow[:, 1] / ((h - 1.0) / 2.0)
    sx = torch.tensor((w - 1) / 2.0, dtype=F32)
    sy = torch.tensor((h - 1) / 2.0, dtype=F32)
    px = torch.clamp((gx + 1.0) * sx, 0.0, float(w - 1))
    py = torch.clamp((gy + 1.0) * sy, 0.0, float(h - 1))
    x0 = px.floor()
    y0 = py.floor()
    wx1 = px - x0
    wx0 = 1.0 - wx1
    wy1 = py - y0
    wy0 = 1.0 - wy1
    x0 = x0.long()
    y0 = y0.long()
    x1 = x0 + 1
    y1 = y0 + 1

    flat = x.reshape(n, c, h * w)

    def tap(yy, xx):
        ok = ((xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1))
        idx = (torch.clamp(yy, 0, h - 1) * w + torch.clamp(xx, 0, w - 1))
        v = torch.gather(flat, 2, idx.view(n, 1, h * w).expand(n, c, h * w))
        return v.view(n, c, h, w) * ok.view(n, 1, h, w).to(x.dtype)

    nw = (wy0 * wx0).unsqueeze(1)
    ne = (wy0 * wx1).unsqueeze(1)
    sw = (wy1 * wx0).unsqueeze(1)
    se = (wy1 * wx1).unsqueeze(1)
    return (tap(y0, x0) * nw + tap(y0, x1) * ne +
            tap(y1, x0) * sw + tap(y1, x1) * se)


# --------------------------------------------------------------------------
# G6  space to depth   (codes/utils/net_utils.py:36-47)
# --------------------------------------------------------------------------
def space_to_depth(x, s):
    """out[n, (sy*s+sx)*C + c, oy, ox] = x[n, c, oy*s+sy, ox*s+sx]."""
    n, c, h, w = x.shape
    out = x.new_empty(n, s * s * c, h // s, w // s)
    for sy in range(s):
        for sx in range(s):
            k = (sy * s + sx) * c
            out[:, k:k + c] = x[:, :, sy::s, sx::s][:, :, :h // s, :w // s]
    return out


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def reflect_pad_br(x, pad_h, pad_w):
    """F.pad(x, (0, pad_w, 0, pad_h), 'reflect') -- tecogan_nets.py:239-241:
    new row H+k mirrors row H-2-k (edge not repeated)."""
    if pad_h:
        rows = [x[:, :, x.shape[2] - 2 - k] for k in range(pad_h)]
        x = torch.cat([x, torch.stack(rows, 2)], 2)
    if pad_w:
        cols = [x[:, :, :, x.shape[3] - 2 - k] for k in range(pad_w)]
        x = torch.cat([x, torch.stack(cols, 3)], 3)
    return x


def _conv(x, sd, key, pad=1, stride=1):
    return F.conv2d(x, sd[key + '.weight'], sd.get(key + '.bias'),
                    stride=stride, padding=pad)


def _lrelu(x):
    return torch.where(x >= 0, x, x * 0.2)


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------
# G1  FNet   (codes/models/networks/tecogan_nets.py:16-82)
# --------------------------------------------------------------------------
def fnet_forward(sd, x1, x2):
    """Flow from x1 to x2 in LR pixels; ch0 = x (W), ch1 = y (H); output
    spatial size is floor(H/8)*8 x floor(W/8)*8.  `sd` holds the keys below
    `fnet.` (encoder1.0.weight, ...)."""
    out = torch.cat([x1, x2], 1)                                 # :71
    for enc in ('encoder1', 'encoder2', 'encoder3'):             # :23-42
        out = _lrelu(_conv(out, sd, enc + '.0'))
        out = _lrelu(_conv(out, sd, enc + '.2'))
        out = F.max_pool2d(out, 2, 2)
    for dec in ('decoder1', 'decoder2', 'decoder3'):             # :44-60, 74-79
        out = _lrelu(_conv(out, sd, dec + '.0'))
        out = _lrelu(_conv(out, sd, dec + '.2'))
        out = bilinear_upsample(out, 2)
    out = _lrelu(_conv(out, sd, 'flow.0'))                       # :62-65
    out = _conv(out, sd, 'flow.2')
    return torch.tanh(out) * 24                                  # :80


# --------------------------------------------------------------------------
# G3  SRNet   (codes/models/networks/tecogan_nets.py:85-147)
# --------------------------------------------------------------------------
def srnet_forward(sd, lr_curr, hr_prev_tran, scale, degradation):
    out = torch.relu(_conv(torch.cat([lr_curr, hr_prev_tran], 1), sd,
                           'conv_in.0'))                         # :141
    nb = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('resblocks.'))
    for b in range(nb):                                          # :92-98
        t = torch.relu(_conv(out, sd, f'resblocks.{b}.conv.0'))
        out = _conv(t, sd, f'resblocks.{b}.conv.2') + out
    for u in ([0, 2] if scale == 4 else [0]):                    # :119-126
        out = torch.relu(F.conv_transpose2d(
            out, sd[f'conv_up.{u}.weight'], sd[f'conv_up.{u}.bias'],
            stride=2, padding=1, output_padding=1))
    out = _conv(out, sd, 'conv_out')                             # :131
    return out + upsample(lr_curr, scale, degradation)           # :145


# --------------------------------------------------------------------------
# G2  FRNet.step   (codes/models/networks/tecogan_nets.py:227-252)
# --------------------------------------------------------------------------
def frnet_step(sd, lr_curr, lr_prev, hr_prev, scale, degradation,
               return_parts=False):
    h, w = lr_curr.shape[2:]
    lr_flow = fnet_forward(_sub(sd, 'fnet.'), lr_curr, lr_prev)  # :235
    lr_flow_pad = reflect_pad_br(lr_flow, h - h // 8 * 8, w - w // 8 * 8)
    hr_flow = scale * upsample(lr_flow_pad, scale, degradation)  # :244
    hr_prev_warp = backward_warp(hr_prev, hr_flow)               # :247
    s2d = space_to_depth(hr_prev_warp, scale)
    hr_curr = srnet_forward(_sub(sd, 'srnet.'), lr_curr, s2d, scale,
                            degradation)                         # :250
    if return_parts:
        return hr_curr, dict(lr_flow=lr_flow, hr_flow=hr_flow,
                             hr_prev_warp=hr_prev_warp, s2d=s2d)
    return hr_curr


# --------------------------------------------------------------------------
# G7  infer_sequence + float32_to_uint8
#     (tecogan_nets.py:254-281, codes/utils/data_utils.py:80-87)
# --------------------------------------------------------------------------
def float32_to_uint8(x):
    """uint8(clip(round_half_even(x * 255), 0, 255)) -- np.round is RNE."""
    x = np.asarray(x, dtype=np.float32)
    return np.uint8(np.clip(np.round(x * np.float32(255)), 0, 255))


def infer_sequence(sd, lr_data, scale, degradation):
    """lr_data: (t, c, h, w) fp32 -> (t, s*h, s*w, c) uint8; zero initial
    state (tecogan_nets.py:269-270)."""
    t, c, h, w = lr_data.shape
    lr_prev = torch.zeros(1, c, h, w)
    hr_prev = torch.zeros(1, c, scale * h, scale * w)
    out = []
    with torch.no_grad():
        for i in range(t):
            lr_curr = lr_data[i:i + 1]
            hr_curr = frnet_step(sd, lr_curr, lr_prev, hr_prev, scale,
                                 degradation)
            lr_prev, hr_prev = lr_curr, hr_curr
            out.append(float32_to_uint8(hr_curr[0].numpy()))
    return np.stack(out).transpose(0, 2, 3, 1)


# --------------------------------------------------------------------------
# G8  forward_sequence   (tecogan_nets.py:174-225)
# --------------------------------------------------------------------------
def forward_sequence(sd, lr_data, scale, degradation):
    n, t, c, h, w = lr_data.shape
    lr_prev = lr_data[:, :-1].reshape(n * (t - 1), c, h, w)
    lr_curr = lr_data[:, 1:].reshape(n * (t - 1), c, h, w)
    lr_flow = fnet_forward(_sub(sd, 'fnet.'), lr_curr, lr_prev)  # :184-186
    hr_flow = scale * upsample(lr_flow, scale, degradation)      # :189
    hr_flow = hr_flow.view(n, t - 1, 2, scale * h, scale * w)
    srsd = _sub(sd, 'srnet.')
    hr_prev = srnet_forward(srsd, lr_data[:, 0],
                            lr_data.new_zeros(n, scale * scale * c, h, w),
                            scale, degradation)                  # :194-197
    hr = [hr_prev]
    for i in range(1, t):                                        # :201-212
        warped = backward_warp(hr_prev, hr_flow[:, i - 1])
        hr_prev = srnet_forward(srsd, lr_data[:, i],
                                space_to_depth(warped, scale), scale,
                                degradation)
        hr.append(hr_prev)
    return dict(hr_data=torch.stack(hr, 1), hr_flow=hr_flow, lr_prev=lr_prev,
                lr_curr=lr_curr, lr_flow=lr_flow)


# --------------------------------------------------------------------------
# G9  profile() FLOP / parameter accounting
#     (tecogan_nets.py:295-314, codes/metrics/model_summary.py:16-53)
# --------------------------------------------------------------------------
def profile_counts(in_nc, out_nc, nf, nb, scale, lr_h, lr_w):
    """Returns ({'FNet': gflops, 'SRNet': gflops}, {'FNet': params, 'SRNet': params})."""
    def cg(ci, co, k, h, w):
        return 2 * ci * k * k * co * h * w / 1e9

    def cp(ci, co, k):
        return ci * co * k * k + co

    g = 0.0
    p = 0
    h, w = lr_h, lr_w
    plan = [('enc', 2 * in_nc, 32), ('enc', 32, 64), ('enc', 64, 128),
            ('dec', 128, 256), ('dec', 256, 128), ('dec', 128, 64)]
    for kind, ci, co in plan:
        g += cg(ci, co, 3, h, w) + cg(co, co, 3, h, w)
        p += cp(ci, co, 3) + cp(co, co, 3)
        if kind == 'enc':
            h, w = h // 2, w // 2
        else:
            h, w = h * 2, w * 2
    g += cg(64, 32, 3, h, w) + cg(32, 2, 3, h, w)
    p += cp(64, 32, 3) + cp(32, 2, 3)
    gf, pf = g, p

    g = 0.0
    p = 0
    h, w = lr_h, lr_w
    cin = (scale * scale + 1) * in_nc
    g += cg(cin, nf, 3, h, w)
    p += cp(cin, nf, 3)
    g += 2 * nb * cg(nf, nf, 3, h, w)
    p += 2 * nb * cp(nf, nf, 3)
    for _ in range(2 if scale == 4 else 1):
        g += cg(nf, nf, 3, h, w)       # counted at input resolution
        p += cp(nf, nf, 3)
        h, w = h * 2, w * 2
    g += cg(nf, out_nc, 3, h, w)
    p += cp(nf, out_nc, 3)
    return {'FNet': gf, 'SRNet': g}, {'FNet': pf, 'SRNet': p}


# --------------------------------------------------------------------------
# T3  BD degradation   (codes/utils/data_utils.py:11-53, base_model.py:42-85)
# --------------------------------------------------------------------------
def gaussian_kernel2d(sigma, ksize=None):
    """data_utils.py:11-20: outer product of a sampled Gaussian window
    exp(-x^2 / (2 sigma^2)), normalised to sum 1; ksize = 1 + 2*int(3 sigma)."""
    if ksize is None:
        ksize = 1 + 2 * int(sigma * 3.0)
    xs = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    g1 = np.exp(-0.5 * (xs / sigma) ** 2)
    g2 = np.outer(g1, g1)
    return np.float32(g2 / g2.sum())


def downsample_bd(data, sigma, scale, pad_data):
    """Per-channel Gaussian blur + stride-`scale` decimation
    (data_utils.py:30-53).  pad_data=True: reflect pad (k-1)//2 before,
    k-1-(k-1)//2 after (testing); False: valid conv (training)."""
    k = torch.from_numpy(gaussian_kernel2d(sigma))
    ks = k.shape[0]
    n, c, h, w = data.shape
    if pad_data:
        pb = (ks - 1) // 2
        pa = ks - 1 - pb
        data = F.pad(data, (pb, pa, pb, pa), mode='reflect')
    wgt = k.view(1, 1, ks, ks).repeat(c, 1, 1, 1)
    return F.conv2d(data, wgt, stride=scale, groups=c)


# --------------------------------------------------------------------------
# PSNR-Y (parity metric)   (codes/metrics/metric_calculator.py:228-244,
#                           codes/utils/data_utils.py:56-77)
# --------------------------------------------------------------------------
_YCBCR_T = np.array([[0.256788235294118, -0.148223529411765, 0.439215686274510],
                     [0.504129411764706, -0.290992156862745, -0.367788235294118],
                     [0.097905882352941, 0.439215686274510, -0.071427450980392]],
                    dtype=np.float64)
_YCBCR_O = np.array([16, 128, 128], dtype=np.float64)


def rgb_to_ycbcr(img_u8):
    res = np.matmul(img_u8.astype(np.float64), _YCBCR_T) + _YCBCR_O
    return res.clip(0, 255).round().astype(np.uint8)


def psnr(true_u8, pred_u8, y_only=True):
    """20*log10(255 / sqrt(mean(diff^2))) on the Y channel of hwc uint8
    frames (metric_calculator.py:228-244); inf when identical."""
    if y_only:
        true_u8 = rgb_to_ycbcr(true_u8)[..., 0]
        pred_u8 = rgb_to_ycbcr(pred_u8)[..., 0]
    diff = true_u8.astype(np.float64) - pred_u8.astype(np.float64)
    rmse = np.sqrt(np.mean(np.power(diff, 2)))
    if rmse == 0:
        return np.inf
    return 20 * np.log10(255.0 / rmse)


def psnr_float(a, b, peak=1.0):
    """PSNR between two float frames (used for the 1e-3 dB parity bound)."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    if mse == 0:
        return np.inf
    return 10 * math.log10(peak * peak / mse)


# ==========================================================================
# Training rows (D1, T1-T4).  All ops above are built
# from differentiable torch primitives, so torch autograd over THIS file is
# the gradient oracle for the HIP backward kernels.
# ==========================================================================

# --------------------------------------------------------------------------
# T4  losses   (codes/models/optim/losses.py:6-50)
# --------------------------------------------------------------------------
def charbonnier(x, y, reduction='mean', eps=1e-6):
    """sqrt(d^2 + eps), losses.py:31-50."""
    d = x - y
    v = torch.sqrt(d * d + eps)
    return v.mean() if reduction == 'mean' else v.sum()


def bce_with_logits(x, status, reduction='mean'):
    """VanillaGANLoss, losses.py:6-14: BCEWithLogits against a constant target
    t in {0,1}:  max(x,0) - x*t + log(1 + exp(-|x|))."""
    t = float(int(status))
    v = torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-x.abs()))
    return v.mean() if reduction == 'mean' else v.sum()


def pointwise_criterion(kind, x, y, reduction='mean'):
    """define_criterion's element-wise losses, optim/__init__.py:10-18."""
    if kind == 'CB':
        return charbonnier(x, y, reduction)
    d = x - y
    v = d.abs() if kind == 'L1' else d * d
    return v.mean() if reduction == 'mean' else v.sum()


def cosine_similarity_loss(x, y, eps=1e-8):
    """CosineSimilarityLoss, losses.py:53-62: 1 - mean over pixels of the channel-wise cosine;
    F.cosine_similarity = sum_c x/max(|x|,eps) * y/max(|y|,eps)  (ATen, torch >= 1.12)."""
    # linalg.vector_norm (not sqrt(sum)): its gradient at a zero vector is the subgradient 0,
    # as in ATen's cosine_similarity, so a zero feature vector yields y/(eps|y|), not NaN
    xn = torch.linalg.vector_norm(x, 2, dim=1, keepdim=True).clamp_min(eps)
    yn = torch.linalg.vector_norm(y, 2, dim=1, keepdim=True).clamp_min(eps)
    return 1.0 - ((x / xn) * (y / yn)).sum(1).mean()


# --------------------------------------------------------------------------
# VGG19 perceptual features   (codes/models/networks/vgg_nets.py:6-38;
# architecture = torchvision vgg19 "E": conv indices below, ReLU after each, 'M' = maxpool2)
# --------------------------------------------------------------------------
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
             512, 512, 512, 512, 'M']
VGG_MEAN, VGG_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def vgg19_features(sd, x, feature_indexs=(8, 17, 26, 35)):
    """sd: {'features.N.weight', 'features.N.bias'}; x in [0,1].  Features are read after the
    layer with the given index of torchvision's `vgg19().features` (ReLU / pool layers)."""
    mean = torch.tensor(VGG_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(VGG_STD, dtype=x.dtype).view(1, 3, 1, 1)
    out = (x - mean) / std
    feats, i = [], 0
    for v in VGG19_CFG:
        if v == 'M':
            out = F.max_pool2d(out, 2, 2)
            if i in feature_indexs:
                feats.append(out)
            i += 1
        else:
            out = F.relu(F.conv2d(out, sd[f'features.{i}.weight'], sd[f'features.{i}.bias'],
                                  padding=1))
            if i + 1 in feature_indexs:
                feats.append(out)
            i += 2
        if i > max(feature_indexs):
            break
    return feats


# --------------------------------------------------------------------------
# D1  SpatioTemporalDiscriminator   (tecogan_nets.py:318-477)
# --------------------------------------------------------------------------
def batch_norm_train(x, weight, bias, running_mean, running_var, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d in train mode (tecogan_nets.py:324-339): batch statistics
    (biased variance) for normalisation; running stats updated in place with the
    UNBIASED variance."""
    n = x.numel() / x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    with torch.no_grad():
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * n / (n - 1))
    xh = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps)
    return xh * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def discriminator_forward(sd, data, lr_data, bi_data, hr_flow, spatial_size,
                          crop_border_ratio, use_pp_crit=True, hr_flow_merge=None,
                          sd_G=None, scale=4, degradation='BD'):
    """forward_sequence, tecogan_nets.py:384-477, use_pp_crit branch (:408-411).
    `sd` tensors for BN running stats are updated in place.  Returns
    (logits (n_clip,1), [4 feature maps], hr_flow_merge)."""
    n, t, c, lr_h, lr_w = lr_data.shape
    hr_h, hr_w = data.shape[3:]
    t = t // 3 * 3
    n_clip = n * t // 3
    c_size = int(spatial_size * crop_border_ratio)
    n_pad = (spatial_size - c_size) // 2
    if hr_flow_merge is None:
        bw = hr_flow[:, 0:t:3]
        idle = torch.zeros_like(bw)
        if use_pp_crit:
            fw = hr_flow.flip(1)[:, 1:t:3]
        else:                                   # tecogan_nets.py:413-425 (extra FNet pass)
            cur = lr_data[:, 1:t:3].reshape(n_clip, c, lr_h, lr_w)
            nxt = lr_data[:, 2:t:3].reshape(n_clip, c, lr_h, lr_w)
            with torch.no_grad():
                f = fnet_forward(_sub(sd_G, 'fnet.'), cur, nxt)
                fw = (scale * upsample(f, scale, degradation)).view(n, t // 3, 2, hr_h, hr_w)
        hr_flow_merge = torch.stack([bw, idle, fw], dim=2).reshape(
            n_clip * 3, 2, hr_h, hr_w).detach()

    def triplets(x):   # (n,t,c,h,w) -> (n_clip, c*3, h, w) in rrrgggbbb order (:440-449)
        x = x[:, :t].reshape(n_clip, 3, c, hr_h, hr_w)
        return x.permute(0, 2, 1, 3, 4).reshape(n_clip, c * 3, hr_h, hr_w)

    cond = triplets(bi_data)
    orig = triplets(data)
    warp = backward_warp(data[:, :t].reshape(n * t, c, hr_h, hr_w), hr_flow_merge)
    warp = warp.view(n_clip, 3, c, hr_h, hr_w).permute(0, 2, 1, 3, 4).reshape(
        n_clip, c * 3, hr_h, hr_w)
    warp = F.pad(warp[..., n_pad:n_pad + c_size, n_pad:n_pad + c_size], (n_pad,) * 4)
    x = torch.cat([orig, warp, cond], 1)                                  # :463
    out = _lrelu(_conv(x, sd, 'conv_in.0'))
    feats = []
    for i in range(1, 5):
        p = f'discriminator_block.block{i}'
        out = F.conv2d(out, sd[p + '.0.weight'], None, stride=2, padding=1)
        out = batch_norm_train(out, sd[p + '.1.weight'], sd[p + '.1.bias'],
                               sd[p + '.1.running_mean'], sd[p + '.1.running_var'])
        sd[p + '.1.num_batches_tracked'] += 1
        out = _lrelu(out)
        feats.append(out)
    logits = F.linear(out.reshape(out.shape[0], -1), sd['dense.weight'], sd['dense.bias'])
    return logits, feats, hr_flow_merge


# --------------------------------------------------------------------------
# T4  Adam   (torch.optim.Adam as used at vsr_model.py:47-52, vsrgan_model.py:76-87)
# --------------------------------------------------------------------------
def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """In-place Adam (no amsgrad): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    b1, b2 = betas
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        if weight_decay:
            g = g + weight_decay * p
        m = state.setdefault('m', {}).setdefault(k, torch.zeros_like(p))
        v = state.setdefault('v', {}).setdefault(k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def _leafs(sd):
    """float tensors of a state dict as autograd leaves (buffers stay plain)."""
    out = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(('running_mean', 'running_var', 'kernels')):
            out[k] = v.detach().clone().requires_grad_(True)
        else:
            out[k] = v.clone()
    return out


def prepare_training_data(gt, scale, degradation, sigma=1.5):
    """BaseModel.prepare_training_data, base_model.py:42-85 (BD): valid-conv
    blur+decimate of the bordered GT, then crop the GT border."""
    assert degradation == 'BD'
    n, t, c, gh, gw = gt.shape
    border = int(sigma * 3.0)
    lr_h, lr_w = (gh - 2 * border) // scale, (gw - 2 * border) // scale
    flat = gt.reshape(n * t, c, gh, gw)
    lr = downsample_bd(flat, sigma, scale, False).view(n, t, c, lr_h, lr_w)
    gtc = flat[..., border:border + scale * lr_h, border:border + scale * lr_w]
    return lr, gtc.reshape(n, t, c, scale * lr_h, scale * lr_w)


# --------------------------------------------------------------------------
# T2  VSRModel.train   (codes/models/vsr_model.py:61-95)
# --------------------------------------------------------------------------
def vsr_train_step(sd_G, adam_G, lr_data, gt_data, scale, degradation, lr=1e-4,
                   pix_w=1.0, warp_w=1.0, reduction='mean'):
    """One FRVSR iteration.  sd_G / adam_G are updated in place.
    Returns (log_dict, grads)."""
    P = _leafs(sd_G)
    out = forward_sequence(P, lr_data, scale, degradation)
    l_pix = pix_w * charbonnier(out['hr_data'], gt_data, reduction)
    lr_warp = backward_warp(out['lr_prev'], out['lr_flow'])
    l_warp = warp_w * charbonnier(lr_warp, out['lr_curr'], reduction)
    (l_pix + l_warp).backward()
    grads = {k: v.grad for k, v in P.items() if torch.is_tensor(v) and v.requires_grad
             and v.grad is not None}
    with torch.no_grad():
        adam_step({k: sd_G[k] for k in grads}, grads, adam_G, lr)
    return {'l_pix_G': l_pix.item(), 'l_warp_G': l_warp.item()}, grads


# --------------------------------------------------------------------------
# T1  VSRGANModel.train   (codes/models/vsrgan_model.py:98-286); the perceptual and
#     feature-matching terms are optional (BASELINE config 3 runs without them)
# --------------------------------------------------------------------------
def vsrgan_train_step(sd_G, sd_D, adam_G, adam_D, state, lr_data, gt_data, scale, degradation,
                      spatial_size, tempo_extent, lr_G=5e-5, lr_D=5e-5, pix_w=1.0, warp_w=1.0,
                      pp_w=0.5, gan_w=0.01, crop_border_ratio=0.75, update_threshold=0.4,
                      reduction='mean', sd_F=None, feat_w=0.2, feature_layers=(8, 17, 26, 35),
                      fm=None, gan_type='GAN', feat_type='CosineSimilarity', feat_reduction='mean'):
    """sd_F: VGG19 weights -> perceptual loss (:226-241), criterion feat_type (CosineSimilarity as in the
    shipped ymls, or L1 / MSE / CB: optim/__init__.py:5-35 accepts any).  fm = dict(kind, weight,
    layer_norm, reduction) -> feature-matching loss (:255-271).  gan_type 'GAN' (VanillaGANLoss,
    losses.py:6-14) or 'LSGAN' (losses.py:17-28: MSE against the constant 1 / 0 target)."""
    def gan(x, status):
        if gan_type == 'LSGAN':
            return pointwise_criterion('MSE', x, torch.full_like(x, float(int(status))), reduction)
        return bce_with_logits(x, status, reduction)
    n, t, c, lr_h, lr_w = lr_data.shape
    gt_h, gt_w = gt_data.shape[3:]
    bi = upsample(lr_data.reshape(n * t, c, lr_h, lr_w), scale, degradation).view(
        n, t, c, gt_h, gt_w)                                                  # :106-108
    lr_data = torch.cat([lr_data, lr_data.flip(1)[:, 1:]], 1)                 # :112-119
    gt_data = torch.cat([gt_data, gt_data.flip(1)[:, 1:]], 1)
    bi = torch.cat([bi, bi.flip(1)[:, 1:]], 1)
    PG, PD = _leafs(sd_G), _leafs(sd_D)
    log = {}
    out = forward_sequence(PG, lr_data, scale, degradation)                   # :129
    hr = out['hr_data']
    kw = dict(lr_data=lr_data, bi_data=bi, hr_flow=out['hr_flow'], spatial_size=spatial_size,
              crop_border_ratio=crop_border_ratio)
    real, real_feats, merge = discriminator_forward(PD, gt_data, **kw)        # :148
    fake, _, _ = discriminator_forward(PD, hr.detach(), hr_flow_merge=merge, **kw)   # :154
    lreal = torch.log(torch.sigmoid(real) + 1e-8).mean()                      # :163-164
    lfake = torch.log(torch.sigmoid(fake) + 1e-8).mean()
    distance = (lreal - lfake).item()
    upd_D = distance < update_threshold                                       # :176
    gD = {}
    if upd_D:
        state['cnt_upd_D'] = state.get('cnt_upd_D', 0) + 1.0
        loss_D = gan(real, True) + gan(fake, False)
        loss_D.backward()
        gD = {k: v.grad for k, v in PD.items() if torch.is_tensor(v) and v.requires_grad
              and v.grad is not None}
        with torch.no_grad():
            # BN running stats were advanced on PD's copies; publish everything
            adam_step({k: PD[k] for k in gD}, gD, adam_D, lr_D)
        log['l_gan_D'] = loss_D.item()
    else:
        log['l_gan_D'] = 0.0
    log['p_real_D'] = real.mean().item()
    log['p_fake_D'] = fake.mean().item()
    log['distance'] = distance
    log['n_upd_D'] = state.get('cnt_upd_D', 0)
    # D frozen (its parameters now hold the UPDATED values, :201-202 after :188)
    PDf = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in PD.items()}
    l_pix = pix_w * charbonnier(hr, gt_data, reduction)                       # :208-212
    lr_warp = backward_warp(out['lr_prev'], out['lr_flow'])                   # :219
    l_warp = warp_w * charbonnier(lr_warp, out['lr_curr'], reduction)
    hr_fw = hr[:, :tempo_extent - 1]                                          # :246-247
    hr_bw = hr[:, tempo_extent:].flip(1)
    l_pp = pp_w * charbonnier(hr_fw, hr_bw, reduction)
    extra = 0.0
    if sd_F is not None:                                                      # :226-241
        hr_f = vgg19_features(sd_F, hr.reshape(-1, c, gt_h, gt_w), feature_layers)
        with torch.no_grad():
            gt_f = vgg19_features(sd_F, gt_data.reshape(-1, c, gt_h, gt_w), feature_layers)
        if feat_type == 'CosineSimilarity':
            l_feat = feat_w * sum(cosine_similarity_loss(a, b) for a, b in zip(hr_f, gt_f))
        else:
            l_feat = feat_w * sum(pointwise_criterion(feat_type, a, b, feat_reduction) for a, b in zip(hr_f, gt_f))
        extra = extra + l_feat
        log['l_feat_G'] = l_feat.item()
    fake_g, fake_feats, _ = discriminator_forward(PDf, hr, hr_flow_merge=merge, **kw)  # :257/:275
    if fm is not None:                                                        # :255-271
        ln = fm.get('layer_norm', [12.0, 14.0, 24.0, 100.0])
        l_fm = fm.get('weight', 1) * sum(
            pointwise_criterion(fm['kind'], ff, rf.detach(), fm.get('reduction', 'mean')) / ln[i]
            for i, (ff, rf) in enumerate(zip(fake_feats, real_feats)))
        extra = extra + l_fm
        log['l_fm_G'] = l_fm.item()
    l_gan = gan_w * gan(fake_g, True)
    (l_pix + l_warp + l_pp + l_gan + extra).backward()
    gG = {k: v.grad for k, v in PG.items() if torch.is_tensor(v) and v.requires_grad
          and v.grad is not None}
    with torch.no_grad():
        adam_step({k: sd_G[k] for k in gG}, gG, adam_G, lr_G)
        for k, v in PD.items():          # publish D params / BN buffers back
            sd_D[k].copy_(v.detach()) if torch.is_tensor(sd_D[k]) and sd_D[k].shape == v.shape \
                else None
    log.update({'l_pix_G': l_pix.item(), 'l_warp_G': l_warp.item(), 'l_pp_G': l_pp.item(),
                'l_gan_G': l_gan.item(), 'p_fake_G': fake_g.mean().item()})
    return log, gG, gD


def spatial_discriminator_forward(sd, data, bi_data, use_cond):
    """SpatialDiscriminator.forward_sequence, tecogan_nets.py:480-534 (train-mode BN)."""
    n, t, c, h, w = data.shape
    x = data.reshape(n * t, c, h, w)
    if use_cond:
        x = torch.cat([bi_data.reshape(n * t, c, h, w), x], 1)
    out = _lrelu(_conv(x, sd, 'conv_in.0'))
    feats = []
    for i in range(1, 5):
        p = f'discriminator_block.block{i}'
        out = F.conv2d(out, sd[p + '.0.weight'], None, stride=2, padding=1)
        out = batch_norm_train(out, sd[p + '.1.weight'], sd[p + '.1.bias'],
                               sd[p + '.1.running_mean'], sd[p + '.1.running_var'])
        out = _lrelu(out)
        feats.append(out)
    return F.linear(out.reshape(out.shape[0], -1), sd['dense.weight'], sd['dense.bias']), feats
