"""FNet / SRNet / FRNet with the reference's API surface
(codes/models/networks/tecogan_nets.py) on the MI355X HIP path.

Modules only *hold* parameters (same names, shapes and default init as the
reference so its .pth files load strictly); forward passes are sequences of
libtecogan_hip.so kernel launches.  FRNet.step / infer_sequence run the whole
frame through one C-ABI call (tg_frnet_step) on a cached plan.
"""
import ctypes
import math
import os
from collections import OrderedDict

import numpy as np
import functools

import torch
import torch.nn as nn

from ... import _lib as L
from ... import ops
from .. import train_graph as TG
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
        self._cache_u = None

    def packed(self):
        """(packed weights, ocb) on the parameter's device; re-packed only when
        the parameter changed (optimizer step / load_state_dict)."""
        w = self.weight
        key = (ops.param_version(w), w.device)
        if self._cache is None or self._cache[0] != key:
            pk, _, _, ocb = ops.pack_conv3x3(w, transposed=self.transposed)
            self._cache = (key, pk, ocb)
        return self._cache[1], self._cache[2]

    def packed_wino(self):
        """Winograd-form weights (tg_pack_conv3x3_wino) of a plain 3x3 layer, or None when the
        layer has no such form (transposed, cout not a multiple of 64, cin < 16)."""
        if self.transposed or self.cout % 64 != 0 or self.cin < 16:
            return None
        w = self.weight
        key = (ops.param_version(w), w.device)
        if self._cache_u is None or self._cache_u[0] != key:
            self._cache_u = (key, ops.pack_conv3x3_wino(w.detach().contiguous()))
        return self._cache_u[1]

    def forward(self, x, act=ops.ACT_NONE, x2=None, res=None, out=None, pool=False):
        if not self.transposed and not pool and \
                TG._prefers_wino(x.shape[0], self.cin, self.cout, x.shape[2], x.shape[3]):
            u = self.packed_wino()
            if u is not None:
                return ops.conv3x3_wino(x, u, self.bias, self.cin, self.cout, act, x2=x2, res=res, out=out)
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

    def forward(self, x1, x2, tape=None):
        x1, x2 = x1.contiguous(), x2.contiguous()
        if tape is not None:
            return self._forward_train(x1, x2, tape)
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


    def _forward_train(self, x1, x2, tape):
        """Same network on the tape (activations kept, backward closures recorded)."""
        out = None
        for i, name in enumerate(('encoder1', 'encoder2', 'encoder3')):
            blk = getattr(self, name)
            if i == 0:
                out = TG.conv3x3(tape, blk['0'], x1, TG.LRELU, x2=x2, need_dx=False, need_dx2=False)
            else:
                out = TG.conv3x3(tape, blk['0'], out, TG.LRELU)
            out = TG.conv3x3(tape, blk['2'], out, TG.LRELU)
            out = TG.maxpool2(tape, out)
        for name in ('decoder1', 'decoder2', 'decoder3'):
            blk = getattr(self, name)
            out = TG.conv3x3(tape, blk['0'], out, TG.LRELU)
            out = TG.conv3x3(tape, blk['2'], out, TG.LRELU)
            out = TG.upsample(tape, out, 2, ops.UP_BILINEAR)
        out = TG.conv3x3(tape, self.flow['0'], out, TG.LRELU)
        return TG.conv3x3_small(tape, self.flow['2'], out, TG.TANH24)


class _ResBlock(nn.Module):
    def __init__(self, nf):
        super().__init__()
        self.conv = _block([(0, _Conv(nf, nf)), (2, _Conv(nf, nf))])


class SRNet(nn.Module):
    """Reconstruction network, tecogan_nets.py:103-147."""

    # training: conv_in + residual blocks of a frame as one chained launch when the shape allows
    # (class-wide switch; an instance may override it)
    chain_body = True

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

    def forward(self, lr_curr, hr_prev_tran, tape=None, bi=None):
        """bi (training only): `upsample_func(lr_curr)` computed by the caller for all frames at once."""
        lr_curr, hr_prev_tran = lr_curr.contiguous(), hr_prev_tran.contiguous()
        if tape is not None:
            n_, _, h_, w_ = lr_curr.shape
            if self.chain_body and TG._ChainState.usable(
                    n_, self.conv_in['0'].cout, self.conv_in['0'].cin, h_, w_, len(self.resblocks)):
                # conv_in + the residual blocks as ONE launch (and one for their reverse sweep)
                out = TG.srnet_body(tape, self, lr_curr, hr_prev_tran)
            else:
                out = TG.conv3x3(tape, self.conv_in['0'], lr_curr, TG.RELU, x2=hr_prev_tran,
                                 need_dx=False, need_dx2=True)
                for rb in self.resblocks:
                    out = TG.resblock(tape, rb.conv['0'], rb.conv['2'], out)
            for k in self.conv_up:
                out = TG.convt3x3s2(tape, self.conv_up[k], out, TG.RELU)
            return TG.conv3x3_small(tape, self.conv_out, out, TG.NONE, up_src=lr_curr,
                                    up_mode=self.up_mode(), up_scale=self.scale, res=bi)
        out = self.conv_in['0'](lr_curr, ops.ACT_RELU, x2=hr_prev_tran)
        for rb in self.resblocks:
            t = rb.conv['0'](out, ops.ACT_RELU)
            out = rb.conv['2'](t, ops.ACT_NONE, res=out)
        for k in self.conv_up:
            out = self.conv_up[k](out, ops.ACT_RELU)
        head = self.conv_out
        return ops.conv3x3_small(out, head.weight, head.bias, ops.ACT_NONE, up_src=lr_curr,
                                 up_mode=self.up_mode(), up_scale=self.scale)


def _norm_device(device):
    """'cuda' -> cuda:<current>: plans and side streams are cached per device, and torch.device('cuda')
    != torch.device('cuda', 0) -- rounds 1-2 re-created the side stream on every clip because of it."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


_SIDE_STREAMS = {}


def side_stream(dev, kind='side'):
    """ONE long-lived side stream per (device, kind) for the whole process.  Handing out a new pool
    stream per clip (what rounds 1-2 did by accident) walks torch's stream pool across the runtime's
    four hardware queues, so every few clips the side stream shares the main stream's queue and the
    overlap is silently lost; a stable stream was measured at the full two-stream rate with and
    without DEBUG_HIP_DYNAMIC_QUEUES, with HIP initialised before or after the import, and next
    to an RCCL process group (tools/stream_probe.py).  A stream created with a
    CU mask (a hardware queue of its own) and a high-priority stream were measured WORSE."""
    key = (str(_norm_device(dev)), kind)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=_norm_device(dev))
    return st


class _StepPlan:
    """Caller-owned device state behind one tg_frnet_plan: packed weights,
    workspace, and the opaque plan handle."""

    def __init__(self, net, n, h, w, device, fnet_only=False):
        lib = L.lib()
        self.cfg = L.FrnetCfg(net.in_nc, net.out_nc, net.nf, net.nb, net.scale,
                              net.srnet.up_mode(), n, h, w, 1 if fnet_only else 0)
        self.n, self.fh, self.fw = n, h // 8 * 8, w // 8 * 8
        nfl = lib.tg_frnet_workspace_floats(ctypes.byref(self.cfg))
        if nfl == 0:
            raise L.TecoganHipError(f'tg_frnet_workspace_floats: unsupported config '
                                    f'n={n} h={h} w={w} nf={net.nf} scale={net.scale}')
        self.workspace = torch.empty(nfl, dtype=torch.float32, device=device)
        self.keep = []          # packed tensors must outlive the plan
        layers = net.fnet.layers() + net.srnet.layers()
        first_up = net.srnet.conv_up['0'] if hasattr(net.srnet, 'conv_up') and '0' in net.srnet.conv_up else None
        arr = (L.LayerWeights * len(layers))()
        for i, m in enumerate(layers):
            if m.cout <= 4:      # direct small-cout kernel takes plain OIHW
                wt = m.weight.detach().contiguous()
            else:
                wt, _ = m.packed()
            b = m.bias.detach().contiguous()
            u = m.packed_wino() if m.cout > 4 else None
            if m is first_up and not fnet_only and net.scale == 4 and net.nf == 64 and n == 1 and \
                    os.environ.get('TG_WINO_RES_CT', '1') != '0':
                # SRNet's first up-sampling layer as the tail of the resident body launch (tg_conv3x3_wino_res.hip);
                # the plan uses it when the frame runs that launch.  TG_WINO_RES_CT=0: always a launch of its own.
                u = ops.pack_wres_convt(m.weight.detach().contiguous())
            self.keep += [wt, b, u]
            arr[i].w, arr[i].b = wt.data_ptr(), b.data_ptr()
            arr[i].u = u.data_ptr() if u is not None else None
        self.handle = ctypes.c_void_p()
        L.check(lib.tg_frnet_plan_create(ctypes.byref(self.cfg), arr, len(layers),
                                         self.workspace.data_ptr(), ctypes.byref(self.handle)),
                'tg_frnet_plan_create')

    def check_chain(self):
        """Raise if a workgroup of the chained SRNet launch gave up waiting for a producer tile since
        the last check (tg_frnet_plan_chain_status; the frames enqueued on this plan since then were
        built on stale data).  A host read of a pinned counter -- no synchronisation; after a
        synchronisation it covers everything enqueued so far.  Every tg_frnet_step* call makes the
        same check on entry, so a fault also surfaces on the NEXT call of any kind on this plan;
        the plan then runs one launch per layer until it re-arms (set_chain_rearm: after 64 clean frames by default,
        the wait doubling with every fault of a re-armed body)."""
        L.check(L.lib().tg_frnet_plan_chain_status(self.handle, None, None), 'chained SRNet launch')

    def chain_state(self):
        """(faults reported so far, chained launch still in use)."""
        f, a = ctypes.c_int(0), ctypes.c_int(0)
        L.lib().tg_frnet_plan_chain_status(self.handle, ctypes.byref(f), ctypes.byref(a))
        return f.value, bool(a.value)

    def set_chain_rearm(self, first_after_frames):
        """Frames on the per-layer fallback before a faulted one-launch body is tried again (0: never; default 64,
        doubled by every fault of a re-armed body) -- tg_frnet_plan_set_chain_rearm."""
        L.check(L.lib().tg_frnet_plan_set_chain_rearm(self.handle, int(first_after_frames)), 'tg_frnet_plan_set_chain_rearm')

    def rearm_state(self):
        """(times the one-launch body was armed again, back-off in force in frames)."""
        r, w = ctypes.c_int(0), ctypes.c_int(0)
        L.check(L.lib().tg_frnet_plan_chain_rearms(self.handle, ctypes.byref(r), ctypes.byref(w)), 'tg_frnet_plan_chain_rearms')
        return r.value, w.value

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
        self._plan = {}
        self._plan_key = None
        # training: True = the weight gradients of the swept half of the unroll go to a side stream
        # under the rest of the sweep (train_graph.Tape.flush_deferred_async).  Off: the weight-gradient
        # kernel holds a whole CU per workgroup (464 registers, 107 KB LDS), so on one GPU nothing of
        # the sweep runs beside it and the second flush only adds launches (measured: 24.5 vs 24.8 ms).
        # round 6: an int = that many hand-over points spread over the sweep (TG_WGRAD_SIDE, read at construction)
        self.wgrad_side_stream = int(os.environ.get('TG_WGRAD_SIDE', '0') or 0)

    # -- plan cache ---------------------------------------------------------
    def _weights_key(self):
        return tuple(ops.param_version(p) for p in self.parameters())

    def _get_plan(self, n, h, w, device, fnet_only=False, wk=None):
        """Plans are cached per (batch, size, device, kind); a weight update drops them all.
        wk: the caller's _weights_key() of THIS call (a walk over ~100 parameters, ~50 us: infer_sequence takes it once
        per clip instead of once per flow pass)."""
        device = _norm_device(device)
        if wk is None:
            wk = self._weights_key()
        if self._plan_key != wk:
            self._plan, self._plan_key = {}, wk
        key = (n, h, w, str(device), fnet_only)
        plan = self._plan.get(key)
        if plan is None:
            plan = self._plan[key] = _StepPlan(self, n, h, w, device, fnet_only)
        return plan

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

    def infer_sequence(self, lr_data, device, pipeline=True, return_device_tensor=False, on_fault='rerun'):
        """tecogan_nets.py:254-281; see _infer_sequence for the data path.

        on_fault: what happens when a one-launch SRNet body (the LDS-resident / chained launch, whose workgroups wait
        for each other and therefore need the GPU to themselves) recorded a fault in this clip -- another tenant or a
        co-running stream kept part of the grid from starting.  The plan has then fallen back to one launch per layer
        (until it re-arms, _StepPlan.set_chain_rearm).  'rerun' (default): warn and compute the clip again on that path -- the caller gets correct frames,
        as from the reference, which has no such failure mode; 'raise': TecoganHipError.  With
        return_device_tensor=True nothing is synchronised, so a fault can only be REPORTED (at the next call's entry
        or by check_faults() after the caller's own synchronisation): that mode always raises -- including here, at the
        entry of the NEXT clip, whatever its on_fault says: the invalid device tensor is already in the caller's hands
        (it used to be downgraded to this clip's rerun warning)."""
        self.check_faults()          # a fault that belongs to an EARLIER clip is never this clip's to repair
        try:
            return self._infer_sequence(lr_data, device, pipeline, return_device_tensor)
        except L.TecoganHipError as e:
            if on_fault != 'rerun' or return_device_tensor or 'timed out' not in str(e):
                raise
            import warnings
            if torch.cuda.is_available():
                torch.cuda.synchronize()             # whatever of the faulted clip is still queued
            warnings.warn('infer_sequence: %s -- the clip is computed again with one launch per layer' % e, RuntimeWarning)
            return self._infer_sequence(lr_data, device, pipeline, False)

    def _infer_sequence(self, lr_data, device, pipeline=True, return_device_tensor=False):
        """lr_data: (t,c,h,w) fp32 (host or device) -> (t, s*h, s*w, c) uint8
        numpy, zero initial state (tecogan_nets.py:254-281).  Frames are quantised on the
        device and there is ONE host synchronisation at the end instead of one per frame.

        A 5-D input (k, t, c, h, w) is a batch of k INDEPENDENT clips of equal length and size,
        advanced in lockstep: every launch of a frame step covers the k clips (the recurrence is
        serial in t only), which fills the tile-quantisation holes a single 134x320 frame leaves
        on 256 CUs; the result is (k, t, s*h, s*w, c).  Each clip's frames equal those of running
        that clip alone up to the summation order of differently batched launches (tile variant
        and split-K factor depend on the pixel count): one uint8 level on a few pixels at most.

        pipeline=True: FNet depends only on the LR frames, so the flows of the next
        TG_FNET_BATCH (default 8) frame pairs are estimated by one batched FNet pass on a
        second HIP stream while warp+SRNet runs frame by frame on the first (two flow
        slots, ordered by events); the serial part of the recurrence is SRNet alone.  Frame 0 warps the
        zero state -- its warped frame is zero whatever the flow -- so it runs on a zero flow buffer, ahead
        of the first flow pass, and its (unused) flow is not estimated: bit-identical, one FNet pass less
        per clip on the critical path.
        pipeline='one_stream': the same batched flow passes, enqueued on the CALLER's stream ahead of
        their frames (nothing runs concurrently; what a single-stream deployment does).
        pipeline=False: the reference's loop shape, one FNet pass per frame inside step().

        Host I/O (the reference moves every frame H2D and back D2H inside its loop, :273-279):
        a host clip is uploaded batch by batch and the uint8 frames of a finished batch are
        copied to pinned host memory on a third (copy) stream while the next batch computes;
        the returned array is backed by that pinned buffer."""
        multi = lr_data.dim() == 5
        if multi:                                           # (k, t, c, h, w) -> frame-major (t, k, c, h, w)
            lr_data = lr_data.permute(1, 0, 2, 3, 4)
        else:
            lr_data = lr_data.unsqueeze(1)
        tot_frm, k, c, h, w = lr_data.size()
        s = self.scale
        dev = _norm_device(device if device is not None else lr_data.device)
        host_in = not lr_data.is_cuda
        stream_io = bool(pipeline) and tot_frm >= 2 and not return_device_tensor
        if host_in and stream_io:
            lr_ext = torch.empty(tot_frm + 1, k, c, h, w, dtype=torch.float32, device=dev)
            lr_ext[0].zero_()                               # frame -1 = zeros (tecogan_nets.py:266)
            lr_host = lr_data.to(dtype=torch.float32).contiguous()
        else:
            lr_ext = torch.cat([torch.zeros(1, k, c, h, w, dtype=torch.float32, device=dev),
                                lr_data.to(dev, dtype=torch.float32, non_blocking=True)], 0)
        lr = lr_ext[1:]
        hr = [torch.zeros(k, c, s * h, s * w, dtype=torch.float32, device=dev),
              torch.empty(k, c, s * h, s * w, dtype=torch.float32, device=dev)]
        u8 = torch.empty(tot_frm, k, s * h, s * w, c, dtype=torch.uint8, device=dev)
        host_out = None
        with torch.no_grad():
            if not pipeline or tot_frm < 2:
                for i in range(tot_frm):
                    self.step(lr[i], lr_ext[i], hr[i & 1], out=hr[(i + 1) & 1], u8_out=u8[i])
            else:
                # FNet needs only the LR frames: the flows of FNET_BATCH consecutive frame
                # pairs are estimated in ONE batched pass on the side stream (large grids, no
                # split-K) while the main stream runs warp + SRNet frame by frame on the
                # previous batch -- the serial part of the recurrence is SRNet alone.
                nb_ = max(1, min(max(1, int(os.environ.get('TG_FNET_BATCH', '8')) // k), tot_frm))
                nb0 = max(1, min(nb_, max(1, int(os.environ.get('TG_FNET_FIRST_BATCH', '8')) // k)))   # lab knob
                batches, i0_ = [], 0           # (first frame, frames); frame 0 needs no flow: the first batch has one more
                while i0_ < tot_frm:
                    cnt_ = min(nb0 + 1 if not batches else nb_, tot_frm - i0_)
                    batches.append((i0_, cnt_))
                    i0_ += cnt_
                wk = self._weights_key()
                plan = self._get_plan(k, h, w, dev, wk=wk)
                lib = L.lib()
                main = torch.cuda.current_stream(dev)
                side = main if pipeline == 'one_stream' else self._side_stream(dev)
                side.wait_stream(main)                      # inputs / weights are ready
                nbatch = len(batches)
                ev_f, ev_s = self._events(nbatch)
                copy = self._copy_stream(dev) if stream_io else None
                ev_in = [torch.cuda.Event() for _ in range(nbatch)] if (stream_io and host_in) else None
                if stream_io:
                    host_out = torch.empty(tot_frm, k, s * h, s * w, c, dtype=torch.uint8, pin_memory=True)
                    copy.wait_stream(main)                  # lr_ext / u8 allocations are visible
                if ev_in is not None:                       # uploads run ahead on the copy stream
                    for b_, (i0, cnt) in enumerate(batches):
                        with torch.cuda.stream(copy):
                            lr_ext[i0 + 1:i0 + 1 + cnt].copy_(lr_host[i0:i0 + cnt], non_blocking=True)
                            ev_in[b_].record(copy)
                fsz = k * 2 * plan.fh * plan.fw * 4         # bytes of one frame's LR flows (k clips)
                # Frame 0 of a clip warps the ZERO state (hr_prev = 0, tecogan_nets.py:266-268): its warped frame is
                # zero whatever the flow is, so it runs on an all-zero flow buffer WITHOUT waiting for the first flow
                # pass -- bit-identical, and the only frame whose flow pass nothing could hide (0.4 ms per clip).
                zkey = (k, plan.fh, plan.fw, str(dev))
                zflow = getattr(self, '_zero_flow', None)
                if zflow is None or zflow[0] != zkey:
                    zflow = self._zero_flow = (zkey, torch.zeros(k * 2 * plan.fh * plan.fw, dtype=torch.float32, device=dev))
                def srnet_frame(i, flow_ptr):
                    L.check(lib.tg_frnet_step_srnet(plan.handle, flow_ptr, lr[i].data_ptr(), hr[i & 1].data_ptr(),
                                                    hr[(i + 1) & 1].data_ptr(), u8[i].data_ptr(),
                                                    main.cuda_stream), 'tg_frnet_step_srnet')

                for b_, (i0, cnt) in enumerate(batches):
                    if ev_in is not None:
                        side.wait_event(ev_in[b_])
                        main.wait_event(ev_in[b_])
                    if i0 == 0:
                        # enqueued BEFORE the first flow pass: the GPU has work ~0.3 ms earlier (the flow pass is one C
                        # call of ~20 launches), which is all a 20-frame clip loses against a 60-frame one per frame
                        srnet_frame(0, zflow[1].data_ptr())
                    f0 = 1 if i0 == 0 else 0                 # frame 0's flow is never used: not estimated
                    npair = cnt - f0
                    if npair > 0:
                        fplan = self._get_plan(npair * k, h, w, dev, fnet_only=True, wk=wk)
                        if b_ >= 2:
                            side.wait_event(ev_s[b_ - 2])   # flow slot b_&1 consumed by batch b_-2
                        L.check(lib.tg_frnet_step_phase(fplan.handle, 1, b_ & 1,
                                                        lr_ext[i0 + f0 + 1:i0 + 1 + cnt].data_ptr(),
                                                        lr_ext[i0 + f0:i0 + cnt].data_ptr(), None, None, None,
                                                        side.cuda_stream), 'tg_frnet_step_phase(1)')
                        ev_f[b_].record(side)
                        main.wait_event(ev_f[b_])
                        flow0 = lib.tg_frnet_plan_flow(fplan.handle, b_ & 1)
                        for j in range(f0, cnt):
                            srnet_frame(i0 + j, flow0 + (j - f0) * fsz)
                    ev_s[b_].record(main)
                    if stream_io:                           # download batch b_ while batch b_+1 computes
                        copy.wait_event(ev_s[b_])
                        with torch.cuda.stream(copy):
                            host_out[i0:i0 + cnt].copy_(u8[i0:i0 + cnt], non_blocking=True)
                main.wait_stream(side)
                if stream_io:
                    main.wait_stream(copy)
        if return_device_tensor:
            # nothing is synchronised here: a fault of the chained launch in THIS clip is caught by
            # the entry check of the next call on the plan, or by check_faults() after the caller's
            # own synchronisation; faults of earlier clips are caught now
            self._get_plan(k, h, w, dev).check_chain()
            return u8.permute(1, 0, 2, 3, 4) if multi else u8[:, 0]
        if host_out is not None:
            torch.cuda.current_stream(dev).synchronize()
            out = host_out.numpy()
        else:
            out = u8.cpu().numpy()
        self._get_plan(k, h, w, dev).check_chain()   # (the clip has been synchronised: a 4-byte read)
        return out.transpose(1, 0, 2, 3, 4) if multi else out[:, 0]

    def check_faults(self):
        """Call after synchronising: raises TecoganHipError if any cached frame plan recorded a fault
        of its chained SRNet launch (see _StepPlan.check_chain).  main.test and bench.py call it
        after every synchronised clip."""
        for plan in self._plan.values():
            plan.check_chain()

    def _copy_stream(self, dev):
        return side_stream(dev, 'copy')

    def _events(self, n):
        """Two event rings, created once and re-recorded by every clip."""
        ev = getattr(self, '_ev', None)
        if ev is None or len(ev[0]) < n:
            ev = self._ev = ([torch.cuda.Event() for _ in range(n)],
                             [torch.cuda.Event() for _ in range(n)])
        return ev

    def _side_stream(self, dev):
        st = getattr(self, '_side', None)        # (tools/stream_probe.py plants other kinds of stream here)
        if st is None or st.device != _norm_device(dev):
            st = self._side = side_stream(dev)
        return st

    def forward_sequence(self, lr_data):
        """Training unroll (tecogan_nets.py:174-225): lr_data (n,t,c,h,w) -> dict with
        hr_data (n,t,c,sh,sw), hr_flow (n,t-1,2,sh,sw), lr_prev, lr_curr, lr_flow.
        The recorded tape is kept in `self.tape`; the wrapper seeds it with
        d loss / d hr_data and d loss / d lr_flow and calls `self.tape.backward()`."""
        lr_data = ops._chk(lr_data.contiguous(), 'lr_data')
        n, t, c, h, w = lr_data.shape
        s = self.scale
        tape = TG.Tape()
        tape.side = side_stream(lr_data.device, 'wgrad') if self.wgrad_side_stream else None
        lr_prev = ops.time_gather(lr_data, list(range(t - 1))).view(n * (t - 1), c, h, w)
        lr_curr = ops.time_gather(lr_data, list(range(1, t))).view(n * (t - 1), c, h, w)
        lr_flow = self.fnet(lr_curr, lr_prev, tape=tape)
        hr_flow_all = TG.upsample(tape, lr_flow, s, self.srnet.up_mode(), mul=float(s))
        hr_flow = hr_flow_all.view(n, t - 1, 2, s * h, s * w)
        # The recurrence walks time: frame-major copies (t, n, ...) of the LR frames and the HR
        # flow make every step's operands contiguous slices (3 transposes instead of a copy per
        # frame and operand), and the flow gradient is written slice by slice into one
        # frame-major buffer that is transposed back once.
        lr_fm = ops.transpose01(lr_data)
        flow_fm = ops.transpose01(hr_flow)
        g_flow_fm = {}

        def flow_grad_finalize():          # runs after every frame's warp has written its slice
            if 'g' in g_flow_fm:
                tape.add_grad(hr_flow_all, ops.transpose01(g_flow_fm.pop('g')).view_as(hr_flow_all))
        tape.record(flow_grad_finalize)

        def flow_grad_slice(i):
            if 'g' not in g_flow_fm:
                g_flow_fm['g'] = torch.zeros_like(flow_fm)
            return g_flow_fm['g'][i]

        frames = []
        zeros = torch.zeros(n, s * s * c, h, w, dtype=torch.float32, device=lr_data.device)
        # upsample_func(lr_curr) of every frame in one launch (tecogan_nets.py:145 evaluates it per frame)
        bi_fm = ops.upsample(lr_fm.view(t * n, c, h, w), s, self.srnet.up_mode()).view(t, n, c, s * h, s * w)
        hr_prev = self.srnet(lr_fm[0], zeros, tape=tape, bi=bi_fm[0])
        frames.append(hr_prev)
        k_side = int(self.wgrad_side_stream) if tape.side is not None else 0
        flush_at = {max(1, round(t * j / (k_side + 1))) for j in range(1, k_side + 1)} if k_side else set()
        for i in range(1, t):
            if i in flush_at:
                # recorded BEFORE frame i's nodes => runs right after frames t-1 .. i have been swept:
                # their weight gradients start on the side stream under the sweep of frames i-1 .. 0
                tape.record(tape.flush_deferred_async)
            tran = TG.backward_warp(tape, hr_prev, flow_fm[i - 1],          # warp -> space_to_depth, one launch
                                    dflow_out=functools.partial(flow_grad_slice, i - 1), s2d=s)
            hr_prev = self.srnet(lr_fm[i], tran, tape=tape, bi=bi_fm[i])
            frames.append(hr_prev)
        hr_data = ops.stack_time(frames)

        def stack_bwd():
            g = tape.pop_grad(hr_data)
            if g is not None:
                g_fm = ops.transpose01(g)
                for i, f in enumerate(frames):
                    tape.add_grad(f, g_fm[i])
        tape.record(stack_bwd)
        self.tape = tape
        return {'hr_data': hr_data, 'hr_flow': hr_flow, 'lr_prev': lr_prev, 'lr_curr': lr_curr,
                'lr_flow': lr_flow}

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


# ====================== discriminator ====================== #
class _Conv4(nn.Module):
    """nn.Conv2d(ci, co, 4, 2, 1, bias=False) parameter holder (default init)."""

    def __init__(self, ci, co):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(co, ci, 4, 4))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class _BN(nn.Module):
    """nn.BatchNorm2d(c) parameter / buffer holder (same state-dict entries)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._pending = 0       # forward passes since the buffer was last brought up to date

    # nn.BatchNorm2d bumps num_batches_tracked with a device kernel per forward (12 per training
    # step here); nothing on the path reads it, so passes are counted on the host and folded into
    # the buffer when a state dict is taken.
    def count_pass(self):
        self._pending += 1

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending:
            self.num_batches_tracked += self._pending
            self._pending = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending = 0
        super()._load_from_state_dict(*args, **kwargs)


class _Linear1(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, k))
        self.bias = nn.Parameter(torch.empty(1))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(k)
        nn.init.uniform_(self.bias, -bound, bound)


class DiscriminatorBlocks(nn.Module):
    """tecogan_nets.py:318-349: 4 x (conv4x4 s2 no-bias, BatchNorm, LeakyReLU)."""

    def __init__(self):
        super().__init__()
        for i, (ci, co) in enumerate([(64, 64), (64, 64), (64, 128), (128, 256)], 1):
            setattr(self, f'block{i}', _block([(0, _Conv4(ci, co)), (1, _BN(co))]))

    def forward(self, x, tape, need_dx, groups=1):
        feats = []
        for i in range(1, 5):
            blk = getattr(self, f'block{i}')
            x = TG.conv4x4s2(tape, blk['0'], x, need_dx=True)   # conv_in below it has parameters
            x = TG.bn_lrelu(tape, blk['1'], x, need_dx=True, groups=groups)
            feats.append(x)
        return x, feats


class SpatioTemporalDiscriminator(nn.Module):
    """tecogan_nets.py:352-477.  Always runs in train mode (batch statistics), as
    the reference does for all three passes of an iteration.

    args_dict carries what the reference's does (net_G, lr_data, bi_data,
    use_pp_crit, crop_border_ratio, hr_flow [, hr_flow_merge]) plus
    `tape` (Tape or None) and `need_input_grad` (third pass: gradient to `data`)."""

    def __init__(self, in_nc, spatial_size, tempo_range, degradation, scale):
        super().__init__()
        assert tempo_range == 3, 'currently only support 3 as tempo_range'
        self.spatial_size, self.tempo_range, self.scale = spatial_size, tempo_range, scale
        self.conv_in = _block([(0, _Conv(in_nc * tempo_range * 3, 64))])
        self.discriminator_block = DiscriminatorBlocks()
        self.dense = _Linear1(256 * spatial_size // 16 * spatial_size // 16)
        self.upsample_func = get_upsampling_func(scale, degradation)

    def forward(self, data, args_dict):
        return self.forward_sequence(data, args_dict)

    def forward_pair(self, data_a, data_b, args_dict):
        """The two critic passes of the D update -- real, then fake (vsrgan_model.py:137-153; neither needs a
        gradient to its input) -- as ONE pass over the stacked clip batch [a | b]: every convolution, its data
        and weight gradients and the dense layer run once on 2 n_clip clips (12-clip batches of 8x8 .. 64x64
        maps leave most of the GPU idle), BatchNorm keeps SEPARATE batch statistics per half and updates the
        running statistics a-then-b as the reference's two passes do, and under data parallelism each layer
        needs one statistics exchange instead of two.  Returns ((logits (2 n_clip, 1), feats), ret_dict):
        rows [:n_clip] belong to data_a."""
        tape = args_dict.get('tape')
        lr_data = args_dict['lr_data']
        n_clip = lr_data.size(0) * (lr_data.size(1) // 3)
        hr_h, hr_w = data_a.shape[3:]
        x = torch.empty(2 * n_clip, 9 * lr_data.size(2), hr_h, hr_w, dtype=torch.float32, device=data_a.device)
        # (both halves are assembled straight into the pair batch; the flow triplets are built once)
        _, ret = self._assemble(data_a, args_dict, None, False, out=x[:n_clip])
        d2 = dict(args_dict)
        d2.update(ret)
        self._assemble(data_b, d2, None, False, out=x[n_clip:])
        out = TG.conv3x3(tape, self.conv_in['0'], x, TG.LRELU, need_dx=False)
        out, feats = self.discriminator_block(out, tape, False, groups=2)
        flat = out.reshape(out.size(0), -1)
        if tape is not None:
            def flat_bwd():
                g = tape.pop_grad(flat)
                if g is not None:
                    tape.add_grad(out, g.view_as(out))
            tape.record(flat_bwd)
        logits = TG.linear1(tape, self.dense, flat, need_dx=True)
        return (logits, feats), ret

    def forward_sequence(self, data, args_dict):
        tape = args_dict.get('tape')
        need_in = bool(args_dict.get('need_input_grad', False))
        x, ret = self._assemble(data, args_dict, tape, need_in)
        out = TG.conv3x3(tape, self.conv_in['0'], x, TG.LRELU, need_dx=need_in)
        out, feats = self.discriminator_block(out, tape, need_in)
        flat = out.reshape(out.size(0), -1)
        if tape is not None:
            def flat_bwd():        # recorded before the linear node => runs right after it
                g = tape.pop_grad(flat)
                if g is not None:
                    tape.add_grad(out, g.view_as(out))
            tape.record(flat_bwd)
        logits = TG.linear1(tape, self.dense, flat, need_dx=True)
        return (logits, feats), ret

    def _assemble(self, data, args_dict, tape, need_in, out=None):
        """The critic's input of one pass: flow triplets, warped frames, channel assembly
        (tecogan_nets.py:384-463) -> (x (n_clip, 27, H, W), {'hr_flow_merge': ...})."""
        lr_data, bi_data, hr_flow = args_dict['lr_data'], args_dict['bi_data'], args_dict['hr_flow']
        n, t, c, lr_h, lr_w = lr_data.size()
        hr_h, hr_w = data.shape[3:]
        s_size = self.spatial_size
        t = t // 3 * 3
        n_clip = n * t // 3
        c_size = int(s_size * args_dict['crop_border_ratio'])
        n_pad = (s_size - c_size) // 2

        if 'hr_flow_merge' not in args_dict:
            if args_dict['use_pp_crit']:
                # [backward flow | 0 | forward flow read off the time-reversed half: valid
                # because the ping-pong sequence is time-symmetric] per triplet, one launch
                tf = hr_flow.shape[1]
                idx = []
                for j in range(t // 3):
                    idx += [3 * j, -1, tf - 1 - (3 * j + 1)]
                hr_flow_merge = ops.time_gather(hr_flow, idx).view(n_clip * 3, 2, hr_h, hr_w)
            else:
                # forward flow frame1 -> frame2 from an extra FNet pass (tecogan_nets.py:413-425);
                # detached, so no tape
                bw = hr_flow[:, 0:t:3]
                net_G = args_dict['net_G']
                lr_curr = lr_data[:, 1:t:3].reshape(n_clip, c, lr_h, lr_w).contiguous()
                lr_next = lr_data[:, 2:t:3].reshape(n_clip, c, lr_h, lr_w).contiguous()
                lr_flow_fw = net_G.fnet(lr_curr, lr_next)
                fw = ops.upsample(lr_flow_fw, self.scale, net_G.srnet.up_mode(),
                                  mul=float(self.scale)).view(n, t // 3, 2, hr_h, hr_w)
                merge = torch.stack([bw, torch.zeros_like(bw), fw], dim=2)
                hr_flow_merge = merge.reshape(n_clip * 3, 2, hr_h, hr_w).contiguous()
        else:
            hr_flow_merge = args_dict['hr_flow_merge']

        t_data = data.shape[1]
        if t == t_data and data.is_contiguous():
            frames = data.view(n * t, c, hr_h, hr_w)
        else:
            frames = ops.time_gather(data, list(range(t))).view(n * t, c, hr_h, hr_w)
        track = tape is not None and need_in
        held = {}
        if track:
            # recorded first => runs last: the gradient of the frames through the warp joins the
            # one through the triplet channels and goes back to `data`
            def frames_bwd():
                g = tape.pop_grad(frames)
                full = held.pop('g_data', None)
                if full is None:
                    return
                if g is not None:
                    g = g.view(n, t, c, hr_h, hr_w)
                    for i in range(n):
                        ops.axpy_(full[i, :t], g[i], 1.0)
                tape.add_grad(data, full)
            tape.record(frames_bwd)
        warped = TG.backward_warp(tape if track else None, frames, hr_flow_merge,
                                  need_dimg=True, need_dflow=False)
        x = ops.d_assemble_fwd(data, warped, bi_data, t, n_pad, c_size, out=out)

        if track:
            def assemble_bwd():
                g = tape.pop_grad(x)
                if g is None:
                    return
                held['g_data'], g_warped = ops.d_assemble_bwd(g, n, t, t_data, c, n_pad, c_size)
                tape.add_grad(warped, g_warped)
            tape.record(assemble_bwd)
        return x, {'hr_flow_merge': hr_flow_merge}


class SpatialDiscriminator(nn.Module):
    """Per-frame critic with optional bicubic condition (tecogan_nets.py:480-534); not
    selected by any shipped config, provided for API completeness."""

    def __init__(self, in_nc, spatial_size, use_cond):
        super().__init__()
        self.use_cond = use_cond
        mult = 2 if use_cond else 1
        self.conv_in = _block([(0, _Conv(in_nc * mult, 64))])
        self.discriminator_block = DiscriminatorBlocks()
        self.dense = _Linear1(256 * spatial_size // 16 * spatial_size // 16)

    def forward(self, data, args_dict):
        return self.forward_sequence(data, args_dict)

    def step(self, x, tape=None, need_dx=False):
        out = TG.conv3x3(tape, self.conv_in['0'], x, TG.LRELU, need_dx=need_dx)
        out, feats = self.discriminator_block(out, tape, need_dx)
        flat = out.reshape(out.size(0), -1)
        if tape is not None:
            def flat_bwd():
                g = tape.pop_grad(flat)
                if g is not None:
                    tape.add_grad(out, g.view_as(out))
            tape.record(flat_bwd)
        return TG.linear1(tape, self.dense, flat, need_dx=True), feats

    def forward_sequence(self, data, args_dict):
        tape = args_dict.get('tape')
        need_in = bool(args_dict.get('need_input_grad', False))
        n, t, c, hr_h, hr_w = data.size()
        frames = data.reshape(n * t, c, hr_h, hr_w).contiguous()
        track = tape is not None and need_in
        if self.use_cond:
            bi = args_dict['bi_data'].reshape(n * t, c, hr_h, hr_w)
            x = torch.cat([bi, frames], dim=1).contiguous()
        else:
            x = frames
        if track:
            def input_bwd():
                g = tape.pop_grad(x)
                if g is not None:
                    gd = g[:, c:] if self.use_cond else g
                    tape.add_grad(data, gd.contiguous().view_as(data).clone())
            tape.record(input_bwd)
        pred = self.step(x, tape, need_in)
        return pred, {}
