"""GPU parity of the training-side kernels against autograd of the CPU oracle /
torch fp32 reference ops (the gradient oracle).  Tolerances: 1e-5 relative to
the gradient's own scale unless stated (different summation order only)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tecogan_oracle as O

T = torch.from_numpy


def rs(seed, shape, lo=-1.0, hi=1.0):
    return T(np.random.RandomState(seed).uniform(lo, hi, shape).astype(np.float32))


def dev(x):
    return x.cuda().contiguous()


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(scope='module')
def ops():
    import tecogan_pytorch_amd.ops as ops_
    return ops_


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 64, 64, 16, 40), (1, 51, 64, 9, 33), (1, 6, 32, 12, 20),
                                            (2, 3, 64, 10, 34), (1, 128, 256, 5, 9), (1, 64, 3, 20, 36),
                                            # narrow maps: the weight gradient folds its pixel tile to 4 x 16 / 8 x 8
                                            (5, 128, 128, 16, 16), (7, 256, 128, 8, 8), (3, 64, 96, 11, 13),
                                            (9, 256, 256, 4, 4), (2, 32, 64, 19, 7), (300, 64, 64, 8, 8),
                                            # waves sharing a tile's pixels (<= 32 channels on a side), <= 4 shifted channels
                                            (2, 27, 64, 20, 36), (3, 32, 64, 9, 33), (2, 64, 32, 12, 40), (2, 3, 32, 16, 24),
                                            (1, 4, 40, 7, 19), (2, 2, 3, 8, 64)])
def test_conv3x3_dgrad_and_wgrad(ops, n, cin, cout, h, w):
    x = rs(1, (n, cin, h, w)).requires_grad_(True)
    wt = (rs(2, (cout, cin, 3, 3)) / (3.0 * cin ** 0.5)).requires_grad_(True)
    dz = rs(3, (n, cout, h, w))
    F.conv2d(x, wt, None, padding=1).backward(dz)
    # data gradient = forward kernel on rot180/transposed weights
    pk, op_cin, op_cout, ocb = ops.pack_conv3x3_dgrad(dev(wt.detach()))
    assert (op_cin, op_cout) == (cout, cin)
    dx = ops.conv3x3(dev(dz), pk, None, cout, cin, ocb, ksplit=1)
    assert relerr(dx, x.grad) <= 1e-5
    # weight gradient
    g = torch.zeros(cout, cin, 3, 3, device='cuda')
    ops.wgrad3x3(dev(dz), dev(x.detach()), g, accumulate=False)
    assert relerr(g, wt.grad) <= 2e-5, relerr(g, wt.grad)
    ops.wgrad3x3(dev(dz), dev(x.detach()), g, accumulate=True)          # accumulate doubles
    assert relerr(g, 2 * wt.grad) <= 2e-5


@pytest.mark.parametrize('nseg,n,ci,co,h,w', [(1, 2, 64, 64, 16, 32), (3, 2, 64, 64, 9, 21), (1, 1, 24, 40, 5, 70),
                                              (19, 2, 64, 64, 32, 32), (2, 3, 64, 16, 7, 33),
                                              (2, 2, 64, 64, 12, 20), (1, 2, 40, 24, 3, 36)])
def test_convt_weight_gradient_straight_from_dz(ops, nseg, n, ci, co, h, w):
    """tg_wgrad3x3_convt_multi: dW of ConvTranspose2d(ci, co, 3, 2, 1, 1) over per-frame (x, dZ) pairs against
    autograd on the concatenated batch."""
    xs = [rs(10 + i, (n, ci, h, w)) for i in range(nseg)]
    dzs = [rs(40 + i, (n, co, 2 * h, 2 * w)) for i in range(nseg)]
    wt = (rs(2, (ci, co, 3, 3)) / (3.0 * ci ** 0.5)).requires_grad_(True)
    bt = torch.zeros(co, requires_grad=True)
    F.conv_transpose2d(torch.cat(xs), wt, bt, stride=2, padding=1, output_padding=1).backward(torch.cat(dzs))
    g = torch.zeros(ci, co, 3, 3, device='cuda')
    db = torch.full((co,), 7.0, device='cuda')
    ops.wgrad3x3_convt_multi([dev(t) for t in xs], [dev(t) for t in dzs], g, accumulate=False, bias_grad=db)
    assert relerr(g, wt.grad) <= 2e-5, relerr(g, wt.grad)
    assert relerr(db, bt.grad) <= 1e-5, relerr(db, bt.grad)                     # (accumulate=False overwrites)
    ops.wgrad3x3_convt_multi([dev(t) for t in xs], [dev(t) for t in dzs], g, accumulate=True, bias_grad=db)
    assert relerr(g, 2 * wt.grad) <= 2e-5 and relerr(db, 2 * bt.grad) <= 1e-5


@pytest.mark.parametrize('cout,c1,c2,h,w', [(64, 3, 48, 16, 24), (64, 16, 16, 12, 40), (32, 12, 20, 9, 36), (48, 2, 30, 7, 21)])
def test_wgrad_two_sources(ops, cout, c1, c2, h, w):
    """cat[x1, x2] input: two calls into column ranges of one gradient (3 + 48: conv_in; the others put
    the column offset on the paths where waves share a tile's pixels / the operands are exchanged)."""
    x = rs(1, (2, c1 + c2, h, w)).requires_grad_(True)
    wt = (rs(2, (cout, c1 + c2, 3, 3)) / 20).requires_grad_(True)
    dz = rs(3, (2, cout, h, w))
    F.conv2d(x, wt, None, padding=1).backward(dz)
    g = torch.zeros(cout, c1 + c2, 3, 3, device='cuda')
    xd = x.detach()
    ops.wgrad3x3(dev(dz), dev(xd[:, :c1]), g, cb_off=0, accumulate=False)
    ops.wgrad3x3(dev(dz), dev(xd[:, c1:]), g, cb_off=c1, accumulate=False)
    assert relerr(g, wt.grad) <= 2e-5, relerr(g, wt.grad)


@pytest.mark.parametrize('act', [1, 2, 3])
def test_act_bwd(ops, act):
    z = rs(1, (3, 5, 7, 9), -2, 2).requires_grad_(True)
    y = {1: torch.relu(z), 2: F.leaky_relu(z, 0.2), 3: torch.tanh(z) * 24}[act]
    dy = rs(2, (3, 5, 7, 9))
    y.backward(dy)
    out = ops.act_bwd(dev(dy), dev(y.detach()), act)
    assert relerr(out, z.grad) <= (2e-5 if act == 3 else 1e-7)


def test_bias_grad_maxpool_bwd(ops):
    dy = rs(1, (3, 7, 10, 11))
    db = torch.zeros(7, device='cuda')
    ops.bias_grad(dev(dy), db, accumulate=False)
    assert relerr(db, dy.sum((0, 2, 3))) <= 1e-6
    x = rs(2, (2, 5, 9, 13)).requires_grad_(True)
    g = rs(3, (2, 5, 4, 6))
    F.max_pool2d(x, 2, 2).backward(g)
    assert torch.equal(ops.maxpool2_bwd(dev(x.detach()), dev(g)).cpu(), x.grad)


@pytest.mark.parametrize('deg,s', [('BD', 4), ('BI', 2), ('BD', 2), ('BI', 4)])
def test_upsample_bwd(ops, deg, s):
    x = rs(1, (2, 3, 9, 13)).requires_grad_(True)
    g = rs(2, (2, 3, 9 * s, 13 * s))
    (float(s) * O.upsample(x, s, deg)).backward(g)
    out = ops.upsample_bwd(dev(g), s, ops.UP_MODE[deg], mul=float(s))
    assert relerr(out, x.grad) <= 1e-5


@pytest.mark.parametrize('scale_flow', [3.0, 40.0])
def test_backward_warp_bwd(ops, scale_flow):
    """autograd of the oracle's backward_warp == grid_sample's backward."""
    x = rs(1, (2, 3, 17, 23), 0, 1).requires_grad_(True)
    fl = (rs(2, (2, 2, 17, 23)) * scale_flow).requires_grad_(True)
    g = rs(3, (2, 3, 17, 23))
    O.backward_warp(x, fl).backward(g)
    dimg, dflow = ops.backward_warp_bwd(dev(x.detach()), dev(fl.detach()), dev(g))
    assert relerr(dimg, x.grad) <= 2e-5
    # flow gradient: positions that sit within 1e-4 px of an integer differ (kink of the
    # bilinear kernel); compare away from kinks
    ref = fl.grad.detach().clone()
    d = (dflow.cpu() - ref).abs()
    assert (d > 1e-3 * ref.abs().max()).float().mean() < 2e-3


def test_warp_bwd_matches_torch_grid_sample(ops):
    """and the oracle's autograd itself equals F.grid_sample's (reference op)."""
    x = rs(1, (1, 3, 12, 15), 0, 1).requires_grad_(True)
    fl = (rs(2, (1, 2, 12, 15)) * 4).requires_grad_(True)
    g = rs(3, (1, 3, 12, 15))
    n, c, h, w = x.shape
    iu = torch.linspace(-1, 1, w).view(1, 1, 1, w).expand(n, -1, h, -1)
    iv = torch.linspace(-1, 1, h).view(1, 1, h, 1).expand(n, -1, -1, w)
    grid = torch.cat([iu, iv], 1) + torch.cat([fl[:, 0:1] / ((w - 1) / 2), fl[:, 1:2] / ((h - 1) / 2)], 1)
    F.grid_sample(x, grid.permute(0, 2, 3, 1), mode='bilinear', padding_mode='border',
                  align_corners=True).backward(g)
    dimg, dflow = ops.backward_warp_bwd(dev(x.detach()), dev(fl.detach()), dev(g))
    assert relerr(dimg, x.grad) <= 2e-5
    assert ((dflow.cpu() - fl.grad).abs() > 1e-3 * fl.grad.abs().max()).float().mean() < 2e-3


def test_depth_to_space_inverts_s2d(ops):
    x = dev(rs(1, (2, 3, 16, 24)))
    for s in (2, 4):
        assert torch.equal(ops.depth_to_space(ops.space_to_depth(x, s), s), x)


def test_losses(ops):
    x = rs(1, (2, 3, 20, 30), 0, 1).requires_grad_(True)
    y = rs(2, (2, 3, 20, 30), 0, 1)
    loss = 0.5 * O.charbonnier(x, y, 'mean')
    loss.backward()
    acc = torch.zeros(1, device='cuda')
    dx = ops.charbonnier(dev(x.detach()), dev(y), acc, 0.5 / x.numel(), grad_scale=0.5 / x.numel())
    assert abs(acc.item() - loss.item()) <= 1e-6 and relerr(dx, x.grad) <= 1e-5
    for target in (1.0, 0.0):
        z = (rs(3, (12, 1)) * 3).requires_grad_(True)
        l = 0.01 * O.bce_with_logits(z, target)
        l.backward()
        st = torch.zeros(3, device='cuda')
        dz = ops.bce_logits(dev(z.detach()), target, st, 1.0 / z.numel(), grad_scale=0.01 / z.numel())
        assert abs(st[0].item() * 0.01 - l.item()) <= 1e-7
        assert abs(st[1].item() - z.mean().item()) <= 1e-6
        assert abs(st[2].item() - torch.log(torch.sigmoid(z) + 1e-8).mean().item()) <= 1e-6
        assert relerr(dz, z.grad) <= 1e-5


def test_adam_matches_oracle(ops):
    p0, g1, g2 = rs(1, (1000,)), rs(2, (1000,)) * 1e-2, rs(3, (1000,)) * 1e-2
    sd, st = {'p': p0.clone()}, {}
    O.adam_step(sd, {'p': g1}, st, 5e-5)
    O.adam_step(sd, {'p': g2}, st, 5e-5)
    p, m, v = dev(p0), torch.zeros(1000, device='cuda'), torch.zeros(1000, device='cuda')
    ops.adam_step(p, dev(g1), m, v, 5e-5, (0.9, 0.999), 1e-8, 0.0, 1)
    ops.adam_step(p, dev(g2), m, v, 5e-5, (0.9, 0.999), 1e-8, 0.0, 2)
    assert (p.cpu() - sd['p']).abs().max() <= 1.5e-7      # 1 ulp at |p| ~ 1; a step is 5e-5


def test_bn_lrelu_and_linear(ops):
    x = rs(1, (4, 8, 6, 5), -2, 2).requires_grad_(True)
    gamma = (1 + 0.2 * rs(2, (8,))).requires_grad_(True)
    beta = (0.1 * rs(3, (8,))).requires_grad_(True)
    rm, rv = torch.zeros(8), torch.ones(8)
    y = O._lrelu(O.batch_norm_train(x, gamma, beta, rm, rv))
    g = rs(4, (4, 8, 6, 5))
    y.backward(g)
    rmd, rvd = torch.zeros(8, device='cuda'), torch.ones(8, device='cuda')
    yd, mean, invstd = ops.bn_lrelu_train_fwd(dev(x.detach()), dev(gamma.detach()), dev(beta.detach()), rmd, rvd)
    assert relerr(yd, y) <= 1e-5 and relerr(rmd, rm) <= 1e-5 and relerr(rvd, rv) <= 1e-5
    dg, db = torch.zeros(8, device='cuda'), torch.zeros(8, device='cuda')
    dx = ops.bn_lrelu_train_bwd(dev(x.detach()), yd, dev(g), dev(gamma.detach()), mean, invstd, dg, db)
    assert relerr(dx, x.grad) <= 2e-5 and relerr(dg, gamma.grad) <= 1e-5 and relerr(db, beta.grad) <= 1e-5
    # Linear(k -> 1)
    xf = rs(5, (6, 1024)).requires_grad_(True)
    w = (rs(6, (1, 1024)) / 32).requires_grad_(True)
    b = rs(7, (1,)).requires_grad_(True)
    out = F.linear(xf, w, b)
    gy = rs(8, (6, 1))
    out.backward(gy)
    od = ops.linear1_fwd(dev(xf.detach()), dev(w.detach()), dev(b.detach()))
    assert relerr(od, out) <= 1e-5
    dw, dbb = torch.zeros(1, 1024, device='cuda'), torch.zeros(1, device='cuda')
    dxf = ops.linear1_bwd(dev(xf.detach()), dev(w.detach()), dev(gy), dw, dbb)
    assert relerr(dxf, xf.grad) <= 1e-6 and relerr(dw, w.grad) <= 1e-5 and relerr(dbb, b.grad) <= 1e-6


def test_sync_bn_path_equals_fused_bn(ops):
    """world_size 1: the SyncBatchNorm halves (moments -> finalize -> apply, bwd reduce ->
    apply) must reproduce the fused single-GPU BatchNorm kernels."""
    x = dev(rs(1, (4, 16, 9, 7), -2, 2))
    gamma, beta = dev(1 + 0.2 * rs(2, (16,))), dev(0.1 * rs(3, (16,)))
    g = dev(rs(4, (4, 16, 9, 7)))
    rm1, rv1 = torch.zeros(16, device='cuda'), torch.ones(16, device='cuda')
    rm2, rv2 = torch.zeros(16, device='cuda'), torch.ones(16, device='cuda')
    y1, m1, i1 = ops.bn_lrelu_train_fwd(x, gamma, beta, rm1, rv1)
    y2, m2, i2, cnt = ops.sync_bn_lrelu_train_fwd(x, gamma, beta, rm2, rv2)
    assert cnt == 4 * 9 * 7
    assert relerr(y2, y1) <= 2e-5 and relerr(m2, m1) <= 1e-5 and relerr(i2, i1) <= 2e-5
    assert relerr(rm2, rm1) <= 1e-5 and relerr(rv2, rv1) <= 2e-5
    dg1, db1 = torch.zeros(16, device='cuda'), torch.zeros(16, device='cuda')
    dg2, db2 = torch.zeros(16, device='cuda'), torch.zeros(16, device='cuda')
    dx1 = ops.bn_lrelu_train_bwd(x, y1, g, gamma, m1, i1, dg1, db1)
    dx2 = ops.sync_bn_lrelu_train_bwd(x, y2, g, gamma, m2, i2, cnt, dg2, db2)
    assert relerr(dx2, dx1) <= 5e-5 and relerr(dg2, dg1) <= 2e-5 and relerr(db2, db1) <= 1e-5


def test_wgrad_and_bias_grad_over_segments_match_concatenation(ops):
    """The per-frame (dZ, X) tensors of an unrolled layer are reduced where they lie: the
    multi-segment launch walks the same tiles in the same order as the launch over the
    concatenated batch, so the weight gradient is bit-identical."""
    g = torch.Generator().manual_seed(11)
    segs, n, ca, cb, h, w = 5, 2, 64, 48, 18, 34
    P = [torch.randn(n, ca, h, w, generator=g).cuda() for _ in range(segs)]
    Q = [torch.randn(n, cb, h, w, generator=g).cuda() for _ in range(segs)]
    ref = torch.zeros(ca, cb + 3, 3, 3, device='cuda')
    ops.wgrad3x3(torch.cat(P), torch.cat(Q), ref, cb_off=3, accumulate=False)
    out = torch.full_like(ref, 7.0)
    out[:, 3:] = 0
    ops.wgrad3x3_multi(P, Q, out, cb_off=3, accumulate=True)
    assert torch.equal(out[:, 3:], ref[:, 3:])
    assert torch.all(out[:, :3] == 7.0)                   # columns outside the slice untouched
    db_ref = torch.zeros(ca, device='cuda')
    ops.bias_grad(torch.cat(P), db_ref, accumulate=False)
    db = torch.ones(ca, device='cuda')
    ops.bias_grad_multi(P, db, accumulate=True)
    assert torch.allclose(db - 1.0, db_ref, rtol=1e-5, atol=1e-3)
    with pytest.raises(Exception):
        ops.wgrad3x3_multi(P, Q[:-1], out)
    with pytest.raises(Exception):
        ops.wgrad3x3_multi(P[:2] + [P[2][:, :, :-1].contiguous()], Q[:3], out)


def test_conv_relu_mask_epilogue_and_fused_resblock_backward(ops):
    """tg_conv3x3_fwd_masked == act_bwd(conv(...)) bit for bit, for aligned and odd widths; the
    hand-fused ResidualBlock backward (2 launches) reproduces the generic tape (5 launches)
    bit for bit in dX and in both weight / bias gradients."""
    from tecogan_pytorch_amd.models import train_graph as TG
    from tecogan_pytorch_amd.models.networks.tecogan_nets import _Conv
    g = torch.Generator().manual_seed(5)
    for (n, h, w) in [(2, 12, 16), (1, 9, 13)]:
        x = torch.randn(n, 64, h, w, generator=g).cuda()
        wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda()
        ymask = torch.randn(n, 64, h, w, generator=g).cuda()
        pk, _, _, ocb = ops.pack_conv3x3(wt)
        ref = ops.conv3x3(x, pk, None, 64, 64, ocb, ksplit=1)
        ref = ops.act_bwd(ref, ymask.clamp_min(0).contiguous(), ops.ACT_RELU)
        out = ops.conv3x3(x, pk, None, 64, 64, ocb, relu_mask=ymask)
        assert torch.equal(out, ref)
    torch.manual_seed(3)
    c1, c2 = _Conv(64, 64).cuda(), _Conv(64, 64).cuda()
    x = torch.randn(2, 64, 16, 24, generator=g).cuda()
    gout = torch.randn(2, 64, 16, 24, generator=g).cuda()
    res = []
    for fused in (True, False):
        for p in list(c1.parameters()) + list(c2.parameters()):
            p.grad = None
        tape = TG.Tape()
        if fused:
            out = TG.resblock(tape, c1, c2, x)
        else:
            t = TG.conv3x3(tape, c1, x, TG.RELU)
            out = TG.conv3x3(tape, c2, t, TG.NONE, res=x)
        tape.add_grad(out, gout.clone())
        tape.backward()
        res.append((out.clone(), tape.grad(x).clone(), c1.weight.grad.clone(), c1.bias.grad.clone(),
                    c2.weight.grad.clone(), c2.bias.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize('ci,co,hw', [(64, 64, 32), (64, 128, 16), (128, 256, 8)])
def test_strided_conv_embeddings_with_phase_restricted_taps(ops, ci, co, hw):
    """Conv2d(k4, s2, p1) forward / data gradient / weight gradient and the ConvTranspose2d
    (k3, s2, p1, op1) data / weight gradients through their space-to-depth embeddings: the
    phase-restricted-tap entry points (4/9 and 9/36 of the MFMAs) against torch autograd on the
    strided ops themselves."""
    import torch.nn.functional as F
    from tecogan_pytorch_amd.models import train_graph as TG

    class Holder(torch.nn.Module):
        def __init__(self, w):
            super().__init__()
            self.weight = torch.nn.Parameter(w)

    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn(3, ci, hw, hw, generator=g)
    w4 = torch.randn(co, ci, 4, 4, generator=g) / (4 * ci ** 0.5)
    xr = x.clone().requires_grad_(True)
    wr = w4.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=2, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    hold = Holder(w4.cuda())
    tape = TG.Tape()
    xd = x.cuda()
    y = TG.conv4x4s2(tape, hold, xd)
    def err(a, b):
        return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()
    assert err(y, yr) <= 2e-4 * max(1.0, yr.abs().max().item())
    hold.weight.grad = torch.zeros_like(hold.weight)
    tape.add_grad(y, gy.cuda())
    tape.backward()
    assert err(tape.grad(xd), xr.grad) <= 2e-4 * max(1.0, xr.grad.abs().max().item())
    assert err(hold.weight.grad, wr.grad) <= 3e-4 * max(1.0, wr.grad.abs().max().item())

    # transposed conv (SRNet.conv_up): forward is the sub-pixel kernel, backward the embedding
    from tecogan_pytorch_amd.models.networks.tecogan_nets import _Conv
    layer = _Conv(ci, 64, transposed=True).cuda()
    xt = torch.randn(2, ci, hw, hw, generator=g)
    xtr = xt.clone().requires_grad_(True)
    wtr = layer.weight.detach().cpu().clone().requires_grad_(True)
    btr = layer.bias.detach().cpu().clone().requires_grad_(True)
    ytr = F.relu(F.conv_transpose2d(xtr, wtr, btr, stride=2, padding=1, output_padding=1))
    gyt = torch.randn(ytr.shape, generator=g)
    ytr.backward(gyt)
    tape = TG.Tape()
    xtd = xt.cuda()
    yt = TG.convt3x3s2(tape, layer, xtd, TG.RELU)
    assert err(yt, ytr) <= 2e-4
    layer.weight.grad = torch.zeros_like(layer.weight)
    layer.bias.grad = torch.zeros_like(layer.bias)
    tape.add_grad(yt, gyt.cuda())
    tape.backward()
    assert err(tape.grad(xtd), xtr.grad) <= 2e-4 * max(1.0, xtr.grad.abs().max().item())
    assert err(layer.weight.grad, wtr.grad) <= 3e-4 * max(1.0, wtr.grad.abs().max().item())
    assert err(layer.bias.grad, btr.grad) <= 3e-4 * max(1.0, btr.grad.abs().max().item())


def test_time_gather_pingpong_and_its_gradient(ops):
    """tg_time_gather / tg_pingpong_grad against the ATen slice / flip / cat chains they
    replace (vsrgan_model.py:112-119, :246-247) — bit-exact, they only move data."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 10, 3, 8, 12, generator=g).cuda()
    assert torch.equal(ops.pingpong(x), torch.cat([x, x.flip(1)[:, 1:]], 1))
    te = 10
    pp = ops.pingpong(x)                                 # (2, 19, ...)
    assert torch.equal(ops.time_gather(pp, list(range(te - 1))), pp[:, :te - 1])
    assert torch.equal(ops.time_gather(pp, [2 * te - 2 - k for k in range(te - 1)]), pp[:, te:].flip(1))
    z = ops.time_gather(x, [3, -1, 0])
    assert torch.equal(z[:, 0], x[:, 3]) and torch.equal(z[:, 2], x[:, 0]) and not z[:, 1].any()
    gg = torch.randn(2, te - 1, 3, 8, 12, generator=g).cuda()
    want = torch.zeros(2, 2 * te - 1, 3, 8, 12, device='cuda')
    want[:, :te - 1] = gg
    want[:, te:] = -gg.flip(1)
    assert torch.equal(ops.pingpong_grad(gg, te), want)
    # and it IS the gradient: d/d(pp) sum(w * (pp[:, :te-1] - pp[:, te:].flip(1)))
    ppr = pp.clone().requires_grad_(True)
    ((ppr[:, :te - 1] - ppr[:, te:].flip(1)) * gg).sum().backward()
    assert torch.equal(ppr.grad, want)


@pytest.mark.parametrize('t_data,pad,crop', [(7, 2, 12), (6, 0, 16)])
def test_discriminator_input_assembly_and_adjoint(ops, t_data, pad, crop):
    """tg_d_assemble_fwd / _bwd against the reference's view / permute / pad / cat chain
    (tecogan_nets.py:440-463) and its autograd adjoint."""
    g = torch.Generator().manual_seed(5)
    n, t, c, H, W = 2, 6, 3, 16, 16
    data = torch.randn(n, t_data, c, H, W, generator=g).cuda().requires_grad_(True)
    warped = torch.randn(n * t, c, H, W, generator=g).cuda().requires_grad_(True)
    cond = torch.randn(n, t_data, c, H, W, generator=g).cuda()
    nclip = n * t // 3

    def trip(x):
        return x.reshape(nclip, 3, c, H, W).permute(0, 2, 1, 3, 4).reshape(nclip, 3 * c, H, W)
    wc = trip(warped)[:, :, pad:pad + crop, pad:pad + crop]
    wc = F.pad(wc, (pad, W - pad - crop, pad, H - pad - crop))
    want = torch.cat([trip(data[:, :t].reshape(n * t, c, H, W)), wc,
                      trip(cond[:, :t].reshape(n * t, c, H, W))], 1)
    got = ops.d_assemble_fwd(data.detach(), warped.detach(), cond, t, pad, crop)
    assert torch.equal(got, want)
    gy = torch.randn(want.shape, generator=g).cuda()
    want.backward(gy)
    g_data, g_warped = ops.d_assemble_bwd(gy, n, t, t_data, c, pad, crop)
    assert torch.equal(g_data, data.grad) and torch.equal(g_warped, warped.grad)


def test_transpose01_and_stack_time(ops):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 5, 2, 4, 8, generator=g).cuda()
    assert torch.equal(ops.transpose01(x), x.transpose(0, 1).contiguous())
    frames = [torch.randn(3, 2, 4, 8, generator=g).cuda() for _ in range(7)]
    assert torch.equal(ops.stack_time(frames), torch.stack(frames, 1))


def test_index_gather_applies_the_embedding_tables(ops):
    from tecogan_pytorch_amd.models import train_graph as TG
    g = torch.Generator().manual_seed(4)
    for kind, shape in (('convt', (6, 4, 3, 3)), ('conv4', (7, 5, 4, 4))):
        w = torch.randn(shape, generator=g).cuda()
        fwd, inv = TG._embed_index(kind, shape[0], shape[1], w.device)
        flat = torch.cat([w.reshape(-1), w.new_zeros(1)])
        want = flat.index_select(0, fwd).view(shape[0], 4 * shape[1], 3, 3)
        got = (TG._convt_embed if kind == 'convt' else TG._conv4_embed)(w)
        assert torch.equal(got, want)
        acc = torch.randn(shape, generator=g).cuda()
        want_acc = acc + want.reshape(-1).index_select(0, inv).view(shape)
        ops.index_gather(want, inv, out=acc, accumulate=True)
        assert torch.equal(acc, want_acc) and torch.equal(want.reshape(-1).index_select(0, inv).view(shape), w)


# ---------------------------------------------------------------------------
# SRNet's conv_in + residual blocks of a training frame as ONE chained launch
# (tg_srnet_body_fwd / _bwd) against the per-layer tape nodes.
# ---------------------------------------------------------------------------
def _srnet(nb=3, scale=4, seed=5, nf=64):
    from tecogan_pytorch_amd.models.networks.tecogan_nets import SRNet
    from tecogan_pytorch_amd.utils.net_utils import get_upsampling_func
    torch.manual_seed(seed)
    net = SRNet(3, 3, nf, nb, get_upsampling_func(scale, 'BD'), scale).cuda()
    for p in net.parameters():                       # O(1) activations through the residual chain
        if p.dim() == 4:
            p.data.mul_(1.5)
    return net


def _body_both_ways(net, lr, tran, g_out):
    """(out, d_tran, {param: grad}) through the chained launches and through one launch per layer."""
    from tecogan_pytorch_amd.models import train_graph as TG
    res = []
    for chained in (True, False):
        for p in net.parameters():
            p.grad = None
        tape = TG.Tape()
        if chained:
            out = TG.srnet_body(tape, net, lr, tran)
        else:
            out = TG.conv3x3(tape, net.conv_in['0'], lr, TG.RELU, x2=tran, need_dx=False, need_dx2=True)
            for rb in net.resblocks:
                out = TG.resblock(tape, rb.conv['0'], rb.conv['2'], out)
        tape.add_grad(out, g_out.clone())
        tape.backward()
        torch.cuda.synchronize()
        TG.chain_check()
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        res.append((out.clone(), tape.grad(tran).clone(), grads))
    return res


@pytest.mark.parametrize('n,h,w,nb,scale,nf', [(2, 32, 32, 10, 4, 64), (2, 64, 64, 3, 4, 64), (1, 20, 37, 2, 4, 64),
                                               (3, 9, 70, 1, 4, 64), (2, 16, 16, 2, 2, 64), (2, 24, 40, 2, 2, 32)])
def test_srnet_body_chain_equals_layer_launches(ops, n, h, w, nb, scale, nf):
    """(scale 2: 12 warped-frame channels -- the data-gradient pack of conv_in would default to the
    32-channel-block layout; nf = 32: so would every forward pack.)"""
    from tecogan_pytorch_amd import _lib as L
    parts = L.lib().tg_conv3x3_chain_supported(n, h, w, 64)
    assert parts in (1, 2, 4)
    net = _srnet(nb, scale=scale, nf=nf)
    ct = 3 * scale * scale
    lr, tran, g = dev(rs(1, (n, 3, h, w), 0, 1)), dev(rs(2, (n, ct, h, w), 0, 1)), dev(rs(3, (n, nf, h, w)))
    (o1, t1, g1), (o2, t2, g2) = _body_both_ways(net, lr, tran, g)
    # same MFMA order and the same fixed-order K reduction as the one-shot kernel when a workgroup
    # owns all 64 channels of a tile; the 32-channel form reduces 8 single-chunk groups instead of
    # 4 groups of 2 chunks: summation order only
    assert relerr(o1, o2) <= 2e-5, relerr(o1, o2)
    # The gradients are NOT continuous in the forward values: where a pre-activation lies within
    # rounding of zero the two forms may disagree about relu'(x) (one such pixel at 2 x 32 x 32:
    # conv_in, image 1, (31, 21)), and that pixel's 3x3 footprint then differs by O(1).  So: equal
    # up to summation order on all but a handful of elements, and close in the L2 sense.
    def mostly_equal(x, y, what, max_bad=1.0):
        x, y = x.double().cpu(), y.double().cpu()
        d = (x - y).abs()
        bad = (d > 2e-5 * y.abs().max()).double().mean().item()
        l2 = (d.norm() / (y.norm() + 1e-30)).item()
        assert bad <= max_bad and l2 <= 3e-2, (what, bad, l2)
    mostly_equal(t1, t2, 'd_tran', 4e-3)      # (a flipped pixel touches a whole filter slice of a weight gradient)
    assert g1.keys() == g2.keys() and len(g1) == 2 * (1 + 2 * nb)
    for k in g1:
        mostly_equal(g1[k], g2[k], k)
    # An exact check of the sweep's algebra: torch fp64 ops driven by the ReLU masks of the chained
    # forward pass itself (so no sign decision is taken twice).
    from tecogan_pytorch_amd.models import train_graph as TG
    import ctypes
    tape = TG.Tape()
    for p in net.parameters():
        p.grad = None
    out = TG.srnet_body(tape, net, lr, tran)
    acts = out._base if out._base is not None else None
    assert acts is not None and acts.shape[0] == 1 + 2 * nb
    tape.add_grad(out, g.clone())
    tape.backward()
    torch.cuda.synchronize()
    TG.chain_check()
    A = acts.double().cpu()
    W = {k: p.detach().double().cpu() for k, p in net.named_parameters()}
    x_in = torch.cat([lr, tran], 1).double().cpu()

    def dgrad(dz, wt):
        return F.conv_transpose2d(dz, wt, padding=1)

    def wgrad(dz, x, wt):
        return torch.nn.grad.conv2d_weight(x, wt.shape, dz, padding=1)
    ref = {}
    gg = g.double().cpu()
    for b in reversed(range(nb)):
        w1, w2 = W[f'resblocks.{b}.conv.0.weight'], W[f'resblocks.{b}.conv.2.weight']
        ref[f'resblocks.{b}.conv.2.weight'] = wgrad(gg, A[1 + 2 * b], w2)
        ref[f'resblocks.{b}.conv.2.bias'] = gg.sum((0, 2, 3))
        dz1 = dgrad(gg, w2) * (A[1 + 2 * b] > 0)
        ref[f'resblocks.{b}.conv.0.weight'] = wgrad(dz1, A[2 * b], w1)
        ref[f'resblocks.{b}.conv.0.bias'] = dz1.sum((0, 2, 3))
        gg = dgrad(dz1, w1) + gg
    dz0 = gg * (A[0] > 0)
    ref['conv_in.0.weight'] = wgrad(dz0, x_in, W['conv_in.0.weight'])
    ref['conv_in.0.bias'] = dz0.sum((0, 2, 3))
    d_tran_ref = dgrad(dz0, W['conv_in.0.weight'])[:, 3:]
    assert relerr(tape.grad(tran), d_tran_ref) <= 2e-5, relerr(tape.grad(tran), d_tran_ref)
    assert {k for k, p in net.named_parameters() if p.grad is not None} == set(ref)
    for k, p in net.named_parameters():
        if k in ref:
            assert relerr(p.grad, ref[k]) <= 1e-4, (k, relerr(p.grad, ref[k]))
    # and the forward pass against the fp32 reference ops
    xr = F.relu(F.conv2d(x_in, W['conv_in.0.weight'], W['conv_in.0.bias'], padding=1))
    for b in range(nb):
        t = F.relu(F.conv2d(xr, W[f'resblocks.{b}.conv.0.weight'], W[f'resblocks.{b}.conv.0.bias'], padding=1))
        xr = F.conv2d(t, W[f'resblocks.{b}.conv.2.weight'], W[f'resblocks.{b}.conv.2.bias'], padding=1) + xr
    assert relerr(o1, xr) <= 2e-5, relerr(o1, xr)


def test_srnet_body_chain_is_deterministic_and_repeatable(ops):
    """40 consecutive chained launches (forward + sweep) on the same buffers: bit-identical results
    every time (fixed-order reductions; flags re-armed by the epoch, never cleared)."""
    net = _srnet(10)
    lr, tran, g = dev(rs(1, (2, 3, 32, 32), 0, 1)), dev(rs(2, (2, 48, 32, 32), 0, 1)), dev(rs(3, (2, 64, 32, 32)))
    from tecogan_pytorch_amd.models import train_graph as TG
    first = None
    for it in range(40):
        tape = TG.Tape()
        out = TG.srnet_body(tape, net, lr, tran)
        tape.add_grad(out, g.clone())
        tape.backward()
        cur = (out.clone(), tape.grad(tran).clone())
        if first is None:
            first = cur
        assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]), it
    torch.cuda.synchronize()
    TG.chain_check()


def test_srnet_body_chain_fault_is_reported_and_falls_back(ops):
    """Fault injection (negative poll limit): chain_check() raises after the synchronisation and the
    chained launch is off for the rest of the process (run in a subprocess for that reason)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from tests.test_hip_train_ops import _srnet, dev, rs\n"
        "from tecogan_pytorch_amd.models import train_graph as TG\n"
        "from tecogan_pytorch_amd import _lib\n"
        "net = _srnet(2)\n"
        "lr, tran = dev(rs(1, (2, 3, 32, 32), 0, 1)), dev(rs(2, (2, 48, 32, 32), 0, 1))\n"
        "assert TG._ChainState.usable(2, 64, 51, 32, 32, 2)\n"
        "ref = TG.srnet_body(None, net, lr, tran).clone(); torch.cuda.synchronize(); TG.chain_check()\n"
        "TG._ChainState.poll_limit = -1\n"
        "bad = TG.srnet_body(None, net, lr, tran); torch.cuda.synchronize()\n"
        "try:\n"
        "    TG.chain_check(); raise SystemExit('no error reported')\n"
        "except _lib.TecoganHipError as e:\n"
        "    assert 'timed out' in str(e)\n"
        "assert not TG._ChainState.usable(2, 64, 51, 32, 32, 2)\n"
        "out = net(lr, tran, tape=TG.Tape()); torch.cuda.synchronize(); TG.chain_check()\n"
        "print('BODY-FAILSAFE-OK')\n" % root)
    r = subprocess.run([sys.executable, '-c', script], timeout=600, capture_output=True, text=True)
    assert r.returncode == 0 and 'BODY-FAILSAFE-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize('n,c,h,w,s', [(2, 5, 8, 16, 2), (1, 64, 12, 32, 4), (2, 3, 16, 48, 4), (1, 7, 6, 24, 2),
                                       (1, 3, 8, 12, 2), (2, 4, 8, 20, 4)])
def test_space_to_depth_vector_path_is_a_permutation(ops, n, c, h, w, s):
    """The 16-byte vector forms of space_to_depth / depth_to_space (w % (4 s) == 0) and the scalar
    forms (other widths) against the reference's view / permute chain (net_utils.py:36-47), bit for
    bit, and as exact inverses of each other."""
    x = rs(1, (n, c, h, w))
    ref = x.view(n, c, h // s, s, w // s, s).permute(0, 3, 5, 1, 2, 4).reshape(n, s * s * c, h // s, w // s)
    y = ops.space_to_depth(dev(x), s)
    assert torch.equal(y.cpu(), ref)
    assert torch.equal(ops.depth_to_space(y, s).cpu(), x)


@pytest.mark.parametrize('n,ci,co,h,w', [(2, 64, 64, 32, 32), (2, 64, 64, 64, 64), (1, 24, 40, 9, 21), (3, 64, 16, 5, 40),
                                          (2, 64, 64, 128, 128)])
def test_convt_data_gradient_as_stride2_conv(ops, n, ci, co, h, w):
    """tg_conv3x3s2_fwd: dX of ConvTranspose2d(ci, co, 3, 2, 1, 1) taken directly from dY (and the
    ReLU-backward mask of the layer below in the same epilogue) against autograd."""
    x = rs(1, (n, ci, h, w)).requires_grad_(True)
    wt = (rs(2, (ci, co, 3, 3)) / (3.0 * ci ** 0.5))
    dy = rs(3, (n, co, 2 * h, 2 * w))
    F.conv_transpose2d(torch.relu(x), wt, None, stride=2, padding=1, output_padding=1).backward(dy)
    assert ops.conv3x3s2_supported(n, co, ci, h, w)
    wk = ops.pack_conv3x3(dev(wt), ocb=64)[0]
    got = ops.conv3x3s2(dev(dy), wk, co, ci, relu_mask=dev(torch.relu(x.detach())))
    assert relerr(got, x.grad) <= 1e-5, relerr(got, x.grad)
    assert not ops.conv3x3s2_supported(8, 64, 64, 256, 256)


@pytest.mark.parametrize('n,h,w', [(2, 32, 32), (2, 64, 64), (3, 9, 70), (1, 20, 37)])
def test_row_chain_with_reused_buffers_equals_layer_launches(ops, n, h, w):
    """tg_conv3x3_chain in SRNet's INFERENCE buffer pattern -- two ping-pong tensors and in-place
    residual sums (include/tecogan_hip.h: 'buffers may be reused along the chain in SRNet's patterns') --
    in all three workgroup forms, launch after launch, against one launch per layer: a write-after-read
    race or a lost flag would show as a mismatch."""
    g = torch.Generator().manual_seed(23)
    nb = 4
    lr, s2d = dev(torch.rand(n, 3, h, w, generator=g)), dev(torch.rand(n, 48, h, w, generator=g))
    ws = [dev(torch.randn(64, 51, 3, 3, generator=g) * 0.04)] + \
         [dev(torch.randn(64, 64, 3, 3, generator=g) * 0.03) for _ in range(2 * nb)]
    bs = [dev(torch.randn(64, generator=g) * 0.1) for _ in range(2 * nb + 1)]
    pks = [ops.pack_conv3x3(x, ocb=64) for x in ws]

    def make(A, B):
        layers = [dict(x=lr, x2=s2d, w=ws[0], bias=bs[0], act=1, y=A)]
        for b in range(nb):
            layers.append(dict(x=A, w=ws[1 + 2 * b], bias=bs[1 + 2 * b], act=1, y=B))
            layers.append(dict(x=B, w=ws[2 + 2 * b], bias=bs[2 + 2 * b], act=0, res=A, y=A))
        return layers
    A1, B1, A2, B2 = (torch.empty(n, 64, h, w, device='cuda') for _ in range(4))
    seq, chain = make(A1, B1), ops.RowChain(make(A2, B2), n, h, w)
    for it in range(6):
        lr.uniform_(); s2d.uniform_()
        for i, d in enumerate(seq):
            ops.conv3x3(d['x'], pks[i][0], d['bias'], d['w'].shape[1], 64, 64, d['act'], x2=d.get('x2'),
                        res=d.get('res'), out=d['y'], ksplit=1)
        chain.run()
        torch.cuda.synchronize()
        assert relerr(A2, A1) <= 2e-5 and relerr(B2, B1) <= 2e-5, (it, chain.parts, relerr(A2, A1), relerr(B2, B1))
        if it:
            assert torch.equal(A2, prev_a) is False        # (new inputs every iteration)
        prev_a = A2.clone()
    assert chain.faults() == 0
    with pytest.raises(Exception):                          # weights in the wrong layout are refused
        from tecogan_pytorch_amd import _lib as L
        L.check(L.lib().tg_conv3x3_chain(chain.arr, chain.nl, n, h, w, 16 if chain.layout == 64 else 64,
                                         chain.flags.data_ptr(), chain.err.data_ptr(), 999, 1 << 21, None), 'layout')


def test_body_weight_and_bias_gradients_in_one_launch(ops):
    """tg_wgrad3x3_body / tg_bias_grad_body over per-frame blocks against one wgrad3x3_multi /
    bias_grad_multi per layer."""
    nf, nl, n, h, w, frames = 64, 7, 2, 16, 24, 5
    acts = [dev(rs(10 + f, (nl, n, nf, h, w))) for f in range(frames)]
    dz = [dev(rs(50 + f, (nl, n, nf, h, w))) for f in range(frames)]
    grads = [torch.zeros(nf, nf, 3, 3, device='cuda') for _ in range(nl - 1)]
    dbs = [torch.zeros(nf, device='cuda') for _ in range(nl)]
    ops.wgrad3x3_body(dz, acts, grads)
    ops.bias_grad_body(dz, dbs)
    for L_ in range(1, nl):
        ref = torch.zeros(nf, nf, 3, 3, device='cuda')
        ops.wgrad3x3_multi([d[L_] for d in dz], [a[L_ - 1] for a in acts], ref)
        assert relerr(grads[L_ - 1], ref) <= 2e-5, (L_, relerr(grads[L_ - 1], ref))
    for L_ in range(nl):
        ref = torch.zeros(nf, device='cuda')
        ops.bias_grad_multi([d[L_] for d in dz], ref)
        assert relerr(dbs[L_], ref) <= 1e-5, L_
    ops.wgrad3x3_body(dz, acts, grads)                      # accumulate doubles
    ref = torch.zeros(nf, nf, 3, 3, device='cuda')
    ops.wgrad3x3_multi([d[1] for d in dz], [a[0] for a in acts], ref)
    assert relerr(grads[0], 2 * ref) <= 2e-5


def test_div_scalar_is_ieee_division(ops):
    x = dev(rs(3, (1000,), -5, 5))
    y = x.clone()
    ops.div_scalar_(y, 3.0)
    assert torch.equal(y.cpu(), x.cpu() / 3.0)        # (the host's division: torch's device kernel multiplies by 1/3)
    z = torch.empty_like(x)
    ops.div_scalar_(z, 7.0, x)
    assert torch.equal(z.cpu(), x.cpu() / 7.0)


@pytest.mark.parametrize('n,cin,cout,h,w,mask', [(2, 64, 3, 24, 64, True), (2, 64, 3, 128, 128, True), (3, 32, 2, 9, 12, False),
                                                 (1, 70, 4, 5, 136, True), (2, 16, 1, 7, 8, False)])
def test_data_gradient_of_a_small_cout_head(ops, n, cin, cout, h, w, mask):
    """tg_conv3x3_fewin_fwd: dX of a cout <= 4 conv (conv_out, the flow head) from its <= 4-channel dZ, with the
    ReLU mask of the layer below, against autograd."""
    z = rs(1, (n, cin, h, w), -1, 1)
    x = (torch.relu(z) if mask else z).requires_grad_(True)
    wt = rs(2, (cout, cin, 3, 3)) / (3.0 * cin ** 0.5)
    dy = rs(3, (n, cout, h, w))
    F.conv2d(x, wt, None, padding=1).backward(dy)
    want = x.grad * (x.detach() > 0) if mask else x.grad
    assert ops.conv3x3_fewin_ok(dev(dy), cin, any_size=True)
    wd = dev(wt).transpose(0, 1).flip(2, 3).contiguous()
    got = ops.conv3x3_fewin(dev(dy), wd, relu_mask=dev(x.detach()) if mask else None)
    assert relerr(got, want) <= 1e-5, relerr(got, want)


@pytest.mark.parametrize('rows,k', [(12, 65536), (5, 16388), (3, 1001), (2, 9216)])
def test_linear1_forward_long_rows(ops, rows, k):
    """The critic's dense layer on long rows (16-byte loads, eight in flight; odd lengths element-wise)."""
    x, w, b = rs(1, (rows, k)), rs(2, (1, k)) / k ** 0.5, rs(3, (1,))
    ref = (x.double() @ w.double().t() + b.double()).float()
    got = ops.linear1_fwd(dev(x), dev(w), dev(b))
    assert relerr(got, ref) <= 2e-6, relerr(got, ref)


@pytest.mark.parametrize('act', [1, 2])
def test_depth_to_space_with_the_activation_derivative(ops, act):
    """tg_depth_to_space_act_bwd = depth_to_space followed by act_bwd; and the tape path: a strided conv on
    an activation output delivers the gradient with act'(.) applied, a second consumer is brought to the same form."""
    from tecogan_pytorch_amd.models import train_graph as TG
    ds = dev(rs(1, (3, 4 * 8, 6, 12)))
    y = dev(rs(2, (3, 8, 12, 24), -1, 1))
    got, fused = ops.depth_to_space(ds, 2, act_y=y, act=act)
    assert fused
    want = ops.act_bwd(ops.depth_to_space(ds, 2), y, act)
    assert torch.equal(got, want)

    class Holder(torch.nn.Module):
        def __init__(self, w):
            super().__init__()
            self.weight = torch.nn.Parameter(w)
            self.bias = torch.nn.Parameter(torch.zeros(w.shape[0], device=w.device))
            self.cin, self.cout = w.shape[1], w.shape[0]

        def packed(self):
            pk, _, _, ocb = ops.pack_conv3x3(self.weight)
            return pk, ocb
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 16, 24, generator=g)
    w3 = torch.randn(64, 16, 3, 3, generator=g) * 0.1
    w4 = torch.randn(64, 64, 4, 4, generator=g) * 0.05
    w3b = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    xr = x.clone().requires_grad_(True)
    fa = (lambda t: torch.relu(t)) if act == 1 else (lambda t: F.leaky_relu(t, 0.2))
    yr = fa(F.conv2d(xr, w3, None, padding=1))
    out_r = F.conv2d(yr, w4, None, stride=2, padding=1).sum() * 0.5 + (F.conv2d(yr, w3b, None, padding=1) ** 2).sum() * 0.01
    out_r.backward()
    tape = TG.Tape()
    c3, c4, c3b = Holder(w3.cuda()), Holder(w4.cuda()), Holder(w3b.cuda())
    for m in (c3, c4, c3b):
        m.weight.requires_grad_(False); m.bias.requires_grad_(False)
    xd = x.cuda()
    yd = TG.conv3x3(tape, c3, xd, act=act)
    z3 = TG.conv3x3(tape, c3b, yd)
    z4 = TG.conv4x4s2(tape, c4, yd)          # recorded last: its backward runs first
    tape.add_grad(z4, torch.full_like(z4, 0.5))
    tape.add_grad(z3, 0.02 * z3)
    tape.backward()
    assert id(yd) in tape.act_applied                    # (the strided conv's gradient arrived first, fused)
    assert relerr(tape.grad(xd), xr.grad) <= 2e-5, relerr(tape.grad(xd), xr.grad)


@pytest.mark.parametrize('nseg,n,ca,cb,h,w', [(3, 2, 64, 64, 16, 24), (1, 4, 64, 27, 12, 40), (2, 3, 32, 32, 9, 36),
                                               (1, 5, 128, 64, 16, 16), (1, 6, 256, 128, 8, 8), (2, 2, 64, 3, 10, 8),
                                               (1, 2, 3, 64, 8, 12), (1, 2, 64, 48, 7, 21)])
def test_bias_gradient_rides_on_the_weight_gradient_launch(ops, nseg, n, ca, cb, h, w):
    """tg_wgrad3x3_multi_bias: db out of the staged dZ on every form of the launch (vector / element-wise
    staging, folded tiles, shared pixels, exchanged operands, small-ca) = tg_bias_grad_multi."""
    ps = [dev(rs(10 + i, (n, ca, h, w))) for i in range(nseg)]
    qs = [dev(rs(40 + i, (n, cb, h, w))) for i in range(nseg)]
    g1, g2 = torch.zeros(ca, cb, 3, 3, device='cuda'), torch.zeros(ca, cb, 3, 3, device='cuda')
    db1, db2 = torch.full((ca,), 3.0, device='cuda'), torch.full((ca,), 3.0, device='cuda')
    ops.wgrad3x3_multi(ps, qs, g1, accumulate=True, bias_grad=db1)
    ops.wgrad3x3_multi(ps, qs, g2, accumulate=True)
    ops.bias_grad_multi(ps, db2, accumulate=True)
    assert torch.equal(g1, g2)
    assert relerr(db1, db2) <= 1e-5, relerr(db1, db2)


def test_body_bias_gradients_ride_on_the_layered_launch(ops):
    nf, nl, n, h, w, frames = 64, 5, 2, 16, 24, 3
    acts = [dev(rs(10 + f, (nl, n, nf, h, w))) for f in range(frames)]
    dz = [dev(rs(50 + f, (nl, n, nf, h, w))) for f in range(frames)]
    g1 = [torch.zeros(nf, nf, 3, 3, device='cuda') for _ in range(nl - 1)]
    g2 = [torch.zeros(nf, nf, 3, 3, device='cuda') for _ in range(nl - 1)]
    db1 = [torch.zeros(nf, device='cuda') for _ in range(nl - 1)]
    db2 = [torch.zeros(nf, device='cuda') for _ in range(nl)]
    ops.wgrad3x3_body(dz, acts, g1, dbs=db1)
    ops.wgrad3x3_body(dz, acts, g2)
    ops.bias_grad_body(dz, db2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    for L_ in range(1, nl):
        assert relerr(db1[L_ - 1], db2[L_]) <= 1e-5, L_


@pytest.mark.parametrize('n,ci,co,h,w', [(2, 64, 64, 8, 64), (1, 64, 128, 6, 128), (3, 128, 64, 10, 64), (1, 64, 64, 2, 192),
                                         (3, 64, 128, 32, 32), (2, 128, 256, 16, 16), (24, 128, 256, 16, 16), (1, 64, 64, 8, 32)])
def test_conv4x4s2_direct_kernels_vs_torch(n, ci, co, h, w):
    """tg_conv4x4s2_fwd / _dgrad (Conv2d(ci, co, 4, 2, 1, bias=False) of the discriminator blocks,
    tecogan_nets.py:322-340, taken directly) against torch's CPU conv2d and its autograd input gradient
    (fp64 reference; tolerance 2e-5 relative to the output scale: K = 16 ci fp32 products in another order),
    incl. heights that do not fill the 4-row tiles, both tile forms (64 / 32 output columns), the small-map forms
    (w = 32 / 16: folded rows, input channels split over workgroups + the summing launch) and the act'(.) factor."""
    from tecogan_pytorch_amd import ops
    g_ = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g_)
    wt = torch.randn(co, ci, 4, 4, generator=g_) * 0.05
    assert ops.conv4x4s2_supported(n, ci, co, h, w)
    pf, pd = ops.pack_conv4x4s2(wt.cuda())
    y = ops.conv4x4s2(x.cuda(), pf, co)
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, wt.double(), None, 2, 1)
    assert y.shape == ref.shape
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    g = torch.randn(ref.shape, generator=g_)
    ref.backward(g.double())
    dx = ops.conv4x4s2_dgrad(g.cuda(), pd, ci)
    assert (dx.cpu().double() - xd.grad).abs().max().item() <= 2e-5 * xd.grad.abs().max().item()
    act_y = torch.randn(n, ci, h, w, generator=g_)
    dxa = ops.conv4x4s2_dgrad(g.cuda(), pd, ci, act_y=act_y.cuda(), act=ops.ACT_LRELU02)
    assert torch.equal(dxa.cpu(), torch.where(act_y > 0, dx.cpu(), dx.cpu() * 0.2))
    dxr = ops.conv4x4s2_dgrad(g.cuda(), pd, ci, act_y=act_y.cuda(), act=ops.ACT_RELU)
    assert torch.equal(dxr.cpu(), torch.where(act_y > 0, dx.cpu(), dx.cpu() * 0.0))
    assert not ops.conv4x4s2_supported(n, ci, co, 6, 32) and not ops.conv4x4s2_supported(n, 27, co, h, w)
