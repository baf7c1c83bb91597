"""FNet / SRNet / FRNet with the reference's API surface
(codes/models/networks/tecogan_nets.py) on the MI355X HIP path.

Modules only *hold* parameters (same names, shapes and default init as the
reference so its .pth files load strictly); forward passes are sequences of
libtecogan_hip.so kernel launches.  FRNet.step / infer_sequence run the whole
frame through one C-ABI call (tg_frnet_step) on a cached plan.
"""
import ctypes
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from ... import _lib as L
from ... import ops
from ...utils.net_utils import get_upsampling_func


class _Conv(nn.Module):
    """Parameter holder shaped like nn.Conv2d / nn.ConvTranspose2d (k=3) with
    PyTorch's default initialisation (the reference relies on it: its
    initialize_weights is never called)."""

    def __init__(self, cin, cout, transposed=False):
        super().__init__()
        self.cin, self.cout, self.transposed = cin, cout, transposed
        shape = (cin, cout, 3, 3) if transposed else (cout, cin, 3, 3)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = self.weight.shape[1] * 9
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(self.bias, -bound, bound)
        self._cache = None

    def packed(self):
        """(packed weights, ocb) on the parameter's device; re-packed only when
        the parameter changed (optimizer step / load_state_dict)."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._cache is None or self._cache[0] != key:
            pk, _, _, ocb = ops.pack_conv3x3(w, transposed=self.transposed)
            self._cache = (key, pk, ocb)
        return self._cache[1], self._cache[2]

    def forward(self, x, act=ops.ACT_NONE, x2=None, res=None, out=None, pool=False):
        pk, ocb = self.packed()
        if self.transposed:
            return ops.convt3x3s2(x, pk, self.bias, self.cout, act, out=out)
        return ops.conv3x3(x, pk, self.bias, self.cin, self.cout, ocb, act, x2=x2, res=res,
                           out=out, pool=pool)


def _block(pairs):
    """ModuleDict with numeric child names -> keys like `encoder1.0.weight`,
    `encoder1.2.weight` exactly as the reference's nn.Sequential indices."""
    return nn.ModuleDict({str(i): m for i, m in pairs})


class FNet(nn.Module):
    """Optical flow estimator, tecogan_nets.py:16-82.  forward(x1, x2) -> flow
    from x1 to x2 in LR pixels (ch0 = x, ch1 = y), spatial size floor(./8)*8."""

    def __init__(self, in_nc):
        super().__init__()
        plan = [('encoder1', 2 * in_nc, 32), ('encoder2', 32, 64), ('encoder3', 64, 128),
                ('decoder1', 128, 256), ('decoder2', 256, 128), ('decoder3', 128, 64)]
        for name, ci, co in plan:
            setattr(self, name, _block([(0, _Conv(ci, co)), (2, _Conv(co, co))]))
        self.flow = _block([(0, _Conv(64, 32)), (2, _Conv(32, 2))])

    def layers(self):
        out = []
        for name in ('encoder1', 'encoder2', 'encoder3', 'decoder1', 'decoder2', 'decoder3',
                     'flow'):
            blk = getattr(self, name)
            out += [blk['0'], blk['2']]
        return out

    def forward(self, x1, x2):
        x1, x2 = x1.contiguous(), x2.contiguous()
        out = None
        for i, name in enumerate(('encoder1', 'encoder2', 'encoder3')):
            blk = getattr(self, name)
            out = blk['0'](x1, ops.ACT_LRELU02, x2=x2) if i == 0 else blk['0'](out, ops.ACT_LRELU02)
            out = blk['2'](out, ops.ACT_LRELU02, pool=True)
        for name in ('decoder1', 'decoder2', 'decoder3'):
            blk = getattr(self, name)
            out = blk['2'](blk['0'](out, ops.ACT_LRELU02), ops.ACT_LRELU02)
            out = ops.upsample(out, 2, ops.UP_BILINEAR)
        out = self.flow['0'](out, ops.ACT_LRELU02)
        head = self.flow['2']
        return ops.conv3x3_small(out, head.weight, head.bias, ops.ACT_TANH24)


class _ResBlock(nn.Module):
    def __init__(self, nf):
        super().__init__()
        self.conv = _block([(0, _Conv(nf, nf)), (2, _Conv(nf, nf))])


class SRNet(nn.Module):
    """Reconstruction network, tecogan_nets.py:103-147."""

    def __init__(self, in_nc, out_nc, nf, nb, upsample_func, scale):
        super().__init__()
        self.scale = scale
        self.conv_in = _block([(0, _Conv((scale ** 2 + 1) * in_nc, nf))])
        self.resblocks = nn.ModuleList([_ResBlock(nf) for _ in range(nb)])
        ups = [(0, _Conv(nf, nf, transposed=True))]
        if scale == 4:
            ups.append((2, _Conv(nf, nf, transposed=True)))
        self.conv_up = _block(ups)
        self.conv_out = _Conv(nf, out_nc)
        self.upsample_func = upsample_func

    def layers(self):
        out = [self.conv_in['0']]
        for rb in self.resblocks:
            out += [rb.conv['0'], rb.conv['2']]
        out += [self.conv_up[k] for k in self.conv_up]
        out.append(self.conv_out)
        return out

    def up_mode(self):
        return ops.UP_BICUBIC if isinstance(self.upsample_func, nn.Module) else ops.UP_BILINEAR

    def forward(self, lr_curr, hr_prev_tran):
        lr_curr, hr_prev_tran = lr_curr.contiguous(), hr_prev_tran.contiguous()
        out = self.conv_in['0'](lr_curr, ops.ACT_RELU, x2=hr_prev_tran)
        for rb in self.resblocks:
            t = rb.conv['0'](out, ops.ACT_RELU)
            out = rb.conv['2'](t, ops.ACT_NONE, res=out)
        for k in self.conv_up:
            out = self.conv_up[k](out, ops.ACT_RELU)
        head = self.conv_out
        return ops.conv3x3_small(out, head.weight, head.bias, ops.ACT_NONE, up_src=lr_curr,
                                 up_mode=self.up_mode(), up_scale=self.scale)


class _StepPlan:
    """Caller-owned device state behind one tg_frnet_plan: packed weights,
    workspace, and the opaque plan handle."""

    def __init__(self, net, n, h, w, device):
        lib = L.lib()
        self.cfg = L.FrnetCfg(net.in_nc, net.out_nc, net.nf, net.nb, net.scale,
                              net.srnet.up_mode(), n, h, w)
        nfl = lib.tg_frnet_workspace_floats(ctypes.byref(self.cfg))
        if nfl == 0:
            raise L.TecoganHipError(f'tg_frnet_workspace_floats: unsupported config '
                                    f'n={n} h={h} w={w} nf={net.nf} scale={net.scale}')
        self.workspace = torch.empty(nfl, dtype=torch.float32, device=device)
        self.keep = []          # packed tensors must outlive the plan
        layers = net.fnet.layers() + net.srnet.layers()
        arr = (L.LayerWeights * len(layers))()
        for i, m in enumerate(layers):
            if m.cout <= 4:      # direct small-cout kernel takes plain OIHW
                wt = m.weight.detach().contiguous()
            else:
                wt, _ = m.packed()
            b = m.bias.detach().contiguous()
            self.keep += [wt, b]
            arr[i].w, arr[i].b = wt.data_ptr(), b.data_ptr()
        self.handle = ctypes.c_void_p()
        L.check(lib.tg_frnet_plan_create(ctypes.byref(self.cfg), arr, len(layers),
                                         self.workspace.data_ptr(), ctypes.byref(self.handle)),
                'tg_frnet_plan_create')

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                L.lib().tg_frnet_plan_destroy(self.handle)
        except Exception:
            pass


class FRNet(nn.Module):
    """Frame-recurrent generator, tecogan_nets.py:150-314."""

    def __init__(self, in_nc, out_nc, nf, nb, degradation, scale):
        super().__init__()
        self.in_nc, self.out_nc, self.nf, self.nb = in_nc, out_nc, nf, nb
        self.scale = scale
        self.degradation = degradation
        self.upsample_func = get_upsampling_func(self.scale, degradation)
        self.fnet = FNet(in_nc)
        self.srnet = SRNet(in_nc, out_nc, nf, nb, self.upsample_func, self.scale)
        self._plan = None
        self._plan_key = None

    # -- plan cache ---------------------------------------------------------
    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _get_plan(self, n, h, w, device):
        key = (n, h, w, str(device), self._weights_key())
        if self._plan is None or self._plan_key != key:
            self._plan = _StepPlan(self, n, h, w, device)
            self._plan_key = key
        return self._plan

    # -- reference API ------------------------------------------------------
    def forward(self, lr_data, device=None):
        if self.training:
            return self.forward_sequence(lr_data)
        return self.infer_sequence(lr_data, device)

    def step(self, lr_curr, lr_prev, hr_prev, out=None, u8_out=None):
        """One recurrent frame (tecogan_nets.py:227-252), inference only: a
        single C-ABI call enqueues the ~47 kernels of the frame."""
        lr_curr = ops._chk(lr_curr.contiguous(), 'lr_curr')
        lr_prev = ops._chk(lr_prev.contiguous(), 'lr_prev')
        hr_prev = ops._chk(hr_prev.contiguous(), 'hr_prev')
        n, c, h, w = lr_curr.shape
        s = self.scale
        if lr_prev.shape != lr_curr.shape or hr_prev.shape != (n, c, s * h, s * w):
            raise L.TecoganHipError('step: inconsistent input shapes')
        plan = self._get_plan(n, h, w, lr_curr.device)
        if out is None:
            out = torch.empty(n, self.out_nc, s * h, s * w, dtype=torch.float32,
                              device=lr_curr.device)
        L.check(L.lib().tg_frnet_step(plan.handle, lr_curr.data_ptr(), lr_prev.data_ptr(),
                                      hr_prev.data_ptr(), out.data_ptr(),
                                      None if u8_out is None else u8_out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), 'tg_frnet_step')
        return out

    def step_ops(self, lr_curr, lr_prev, hr_prev):
        """Same frame through the individual op entry points (used by tests to
        localise a mismatch; also the building block of forward_sequence)."""
        n, c, h, w = lr_curr.shape
        lr_flow = self.fnet(lr_curr, lr_prev)
        s2d = ops.flowup_warp_s2d(lr_flow, hr_prev.contiguous(), h, w, self.scale,
                                  self.srnet.up_mode())
        return self.srnet(lr_curr, s2d)

    def infer_sequence(self, lr_data, device):
        """lr_data: (t,c,h,w) fp32 (host or device) -> (t, s*h, s*w, c) uint8
        numpy, zero initial state (tecogan_nets.py:254-281).  The whole clip is
        uploaded once, frames are quantised on the device, and there is one
        host synchronisation at the end instead of one per frame."""
        tot_frm, c, h, w = lr_data.size()
        s = self.scale
        dev = torch.device(device) if device is not None else lr_data.device
        lr = lr_data.to(dev, dtype=torch.float32, non_blocking=True).contiguous()
        zeros_lr = torch.zeros(1, c, h, w, dtype=torch.float32, device=dev)
        hr = [torch.zeros(1, c, s * h, s * w, dtype=torch.float32, device=dev),
              torch.empty(1, c, s * h, s * w, dtype=torch.float32, device=dev)]
        u8 = torch.empty(tot_frm, s * h, s * w, c, dtype=torch.uint8, device=dev)
        with torch.no_grad():
            for i in range(tot_frm):
                lr_prev = zeros_lr if i == 0 else lr[i - 1:i]
                self.step(lr[i:i + 1], lr_prev, hr[i & 1], out=hr[(i + 1) & 1], u8_out=u8[i])
        return u8.cpu().numpy()

    def forward_sequence(self, lr_data):
        raise L.TecoganHipError(
            'forward_sequence (training unroll with autograd) is not built yet on the HIP path; '
            'there is deliberately no ATen fallback')

    def generate_dummy_data(self, lr_size, device):
        c, lr_h, lr_w = lr_size
        s = self.scale
        lr_curr = torch.rand(1, c, lr_h, lr_w, dtype=torch.float32).to(device)
        lr_prev = torch.rand(1, c, lr_h, lr_w, dtype=torch.float32).to(device)
        hr_prev = torch.rand(1, c, s * lr_h, s * lr_w, dtype=torch.float32).to(device)
        return [lr_curr, lr_prev, hr_prev]

    def profile(self, lr_size, device=None):
        """(gflops_dict, params_dict) with the reference's counting convention
        (model_summary.py:16-53: 2*Cin*k*k*Cout*Hout*Wout per conv, transposed
        convs at their input resolution).  Pure host arithmetic."""
        c, h, w = lr_size
        gflops, params = OrderedDict(), OrderedDict()

        def walk(layers, sizes):
            g, p = 0.0, 0
            for m, (hh, ww) in zip(layers, sizes):
                g += 2 * m.cin * 9 * m.cout * hh * ww / 1e9
                p += m.weight.numel() + m.bias.numel()
            return g, p

        sz, hh, ww = [], h, w
        for _ in range(3):
            sz += [(hh, ww)] * 2
            hh, ww = hh // 2, ww // 2
        for _ in range(3):
            sz += [(hh, ww)] * 2
            hh, ww = hh * 2, ww * 2
        sz += [(hh, ww)] * 2
        gflops['FNet'], params['FNet'] = walk(self.fnet.layers(), sz)
        sz = [(h, w)] * (1 + 2 * self.nb)
        hh, ww = h, w
        for _ in self.srnet.conv_up:
            sz.append((hh, ww))
            hh, ww = hh * 2, ww * 2
        sz.append((hh, ww))
        gflops['SRNet'], params['SRNet'] = walk(self.srnet.layers(), sz)
        return gflops, params
