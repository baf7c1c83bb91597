"""A minimal reverse-mode tape whose every arithmetic node is a HIP kernel.

The reference trains through torch.autograd over ATen ops; here the training
graph of this one model family is recorded explicitly: each differentiable op
below runs its forward kernel, and registers a closure that runs the matching
backward kernels (conv data gradients reuse the forward MFMA kernel on
rot180-packed weights, weight gradients use the MFMA wgrad kernel, transposed
and strided convs go through their space-to-depth embedding).  torch is used
for storage and for pure data movement (slicing / stacking / zero fill) only.

Gradients of intermediate tensors are keyed by tensor identity; parameter
gradients accumulate into `param.grad` (allocated and zeroed by the caller).
"""
import os

import torch

from .. import ops

NONE, RELU, LRELU, TANH24 = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LRELU02, ops.ACT_TANH24


class Tape:
    def __init__(self):
        self.nodes = []
        self.grads = {}
        self.keep = []          # tensors that must outlive backward (id() stability)
        self.deferred = {}
        self.deferred_bias = {}
        self.deferred_body = {}  # chained SRNet bodies: per network, the (acts, dz, g_out) blocks of every swept frame
        # ReLU-backward fused into the producer of a gradient: a ReLU layer registers its output in
        # `relu_outputs`; a node that delivers the gradient w.r.t. such a tensor may apply the mask in
        # its own epilogue and say so (add_grad(..., masked=True)); the ReLU layer then skips its
        # act_bwd pass unless some other contribution arrived unmasked.
        self.relu_outputs = set()
        self.unmasked = set()
        # the same idea for the generic conv node's activations (ReLU or LeakyReLU): a conv3x3 node
        # registers its activation output here; a producer of its gradient that can apply act'(.) itself
        # (the depth-to-space of a strided conv's data gradient) delivers it with act_applied=True and the
        # node skips act_bwd.  LeakyReLU' is not idempotent: add_grad brings any other contribution to the
        # same tensor to the same form before it accumulates.
        self.act_outputs = {}
        self.act_applied = set()
        # a consumer that will copy the gradient of t into a buffer of its own anyway (the chained body puts
        # the gradient of its output into the last slot of its dZ block) may reserve that buffer here; a
        # producer that can write into a given tensor then delivers the gradient in place
        self.reserved = {}
        self.side = None        # side stream of the asynchronous weight-gradient flushes
        self._inflight = []     # tensors the side stream still reads (kept alive until the join)

    # -- gradient bookkeeping -------------------------------------------------
    def add_grad(self, t, g, masked=False, act_applied=False):
        """Accumulate g into the gradient of t.  Takes ownership of g.  masked: g already carries
        the ReLU-backward mask of t (t is in relu_outputs).  act_applied: g already is the gradient of
        the pre-activation of t (t is in act_outputs)."""
        k = id(t)
        if not masked:
            self.unmasked.add(k)
        cur = self.grads.get(k)
        if act_applied or k in self.act_applied:
            # LeakyReLU' is not idempotent: every contribution must carry it exactly once
            a = self.act_outputs[k]
            if k in self.act_applied and not act_applied:
                g = ops.act_bwd(g, t, a, out=g)
            elif k not in self.act_applied and cur is not None:
                ops.act_bwd(cur, t, a, out=cur)
            self.act_applied.add(k)
        if cur is None:
            self.grads[k] = g
            self.keep.append(t)
        else:
            ops.axpy_(cur, g, 1.0)

    def grad(self, t):
        return self.grads.get(id(t))

    def pop_grad(self, t):
        return self.grads.pop(id(t), None)

    def record(self, fn):
        self.nodes.append(fn)

    def backward(self):
        for fn in reversed(self.nodes):
            fn()
        self.nodes = []
        if self.side is not None and (self.deferred or self.deferred_bias or self.deferred_body):
            self.flush_deferred_async()      # the tail goes behind the earlier chunks on the side stream
        else:
            self.flush_deferred()
        self.join()

    # -- weight gradients beside the reverse sweep ------------------------------------------------
    # The reverse sweep over the unrolled frames is a serial chain of small, latency-bound launches
    # (one 64-channel layer of a 2 x 64 x 64 frame = 256 workgroups for ~11 us); the weight gradients
    # are large, MFMA-bound launches that depend only on tensors the sweep has already produced.
    # A checkpoint node (FRNet.forward_sequence places one in the middle of the unroll) hands the
    # (dZ, X) pairs collected so far to ONE long-lived side stream; the rest follows at the end of
    # backward(), behind the first chunk (both accumulate into the same gradient buffers: stream
    # order makes that a fixed summation order, run to run).  join() orders the main stream after
    # the side stream before anything reads the gradients.
    def flush_deferred_async(self):
        if self.side is None:
            return self.flush_deferred()
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)                      # everything deferred so far has been enqueued
        for ent in self.deferred.values():
            self._inflight += ent['p'] + ent['q']
        for _, dzs in self.deferred_bias.values():
            self._inflight += dzs
        for ent in self.deferred_body.values():
            self._inflight += ent['acts'] + ent['dz']
        with torch.cuda.stream(self.side):
            self.flush_deferred()

    def join(self):
        if self.side is not None and self._inflight:
            torch.cuda.current_stream().wait_stream(self.side)
        self._inflight = []

    # -- deferred parameter gradients -------------------------------------------
    # A layer is applied once per unrolled frame (19x for SRNet), each time on a small
    # image.  Its weight gradient is a sum over all applications, so the (dZ, X) pairs
    # are collected during the reverse sweep and reduced by ONE wgrad launch per layer
    # over the concatenated batch: 19x fewer launches / split-K reductions, and enough
    # pixel tiles per launch to fill the GPU.
    def defer_wgrad(self, key, p, q, target, cb_off=0, post=None, phased=None, convt=False, bias=None):
        """bias: the layer's bias-gradient buffer -- db = sum of p = dZ comes out of the same launch
        (instead of defer_bias: a second pass over every dZ)."""
        ent = self.deferred.setdefault(key, {'p': [], 'q': [], 'target': target, 'cb_off': cb_off,
                                             'post': post, 'phased': phased, 'convt': convt, 'bias': bias})
        ent['p'].append(p)
        ent['q'].append(q)

    def defer_bias(self, buf, dz):
        self.deferred_bias.setdefault(id(buf), (buf, []))[1].append(dz)

    def defer_body(self, key, layers, acts, dz):
        ent = self.deferred_body.setdefault(key, {'layers': layers, 'acts': [], 'dz': []})
        ent['acts'].append(acts)
        ent['dz'].append(dz)

    def flush_deferred(self):
        def same(ts):
            return all(t.shape == ts[0].shape and t.is_contiguous() for t in ts)
        for ent in self.deferred.values():
            multi = len(ent['p']) > 1 and same(ent['p']) and same(ent['q'])
            if ent.get('convt'):       # (x, dZ at twice the resolution) pairs of a transposed conv
                # (`post` carries the layer's bias-gradient buffer: the same launch sums dZ)
                if same(ent['p']) and same(ent['q']):
                    ops.wgrad3x3_convt_multi(ent['p'], ent['q'], ent['target'], accumulate=True, bias_grad=ent['post'])
                else:
                    for x_, d_ in zip(ent['p'], ent['q']):
                        ops.wgrad3x3_convt_multi([x_.contiguous()], [d_.contiguous()], ent['target'], accumulate=True,
                                                 bias_grad=ent['post'])
            elif ent['post'] is None:
                if multi:      # one launch over the per-frame tensors where they lie
                    ops.wgrad3x3_multi(ent['p'], ent['q'], ent['target'], cb_off=ent['cb_off'],
                                       accumulate=True, bias_grad=ent.get('bias'))
                else:
                    P = ent['p'][0] if len(ent['p']) == 1 else torch.cat(ent['p'], 0)
                    Q = ent['q'][0] if len(ent['q']) == 1 else torch.cat(ent['q'], 0)
                    ops.wgrad3x3(P, Q, ent['target'], cb_off=ent['cb_off'], accumulate=True, bias_grad=ent.get('bias'))
            else:
                p0, q0 = ent['p'][0], ent['q'][0]
                # accumulate=False below: every tap the post hook reads is overwritten
                ge = torch.empty(p0.shape[1], q0.shape[1], 3, 3, dtype=torch.float32, device=p0.device)
                phased = ent.get('phased')
                if phased is not None and (phased[0] % 64 != 0 or not (same(ent['p']) and same(ent['q']))):
                    phased = None                 # the kernel dispatches on 64-channel blocks
                if phased is not None:            # skips the taps a sub-pixel phase does not own
                    ops.wgrad3x3_multi(ent['p'], ent['q'], ge, accumulate=False, phased=phased)
                elif multi:
                    ops.wgrad3x3_multi(ent['p'], ent['q'], ge, accumulate=False)
                else:
                    P = p0 if len(ent['p']) == 1 else torch.cat(ent['p'], 0)
                    Q = q0 if len(ent['q']) == 1 else torch.cat(ent['q'], 0)
                    ops.wgrad3x3(P, Q, ge, accumulate=False)
                ent['post'](ge)
        for ent in self.deferred_body.values():
            # the 2*nb residual-block convs of every swept frame: ONE weight-gradient launch (+ one
            # reduce) and one bias-gradient launch instead of 2*nb of each per flush
            layers = ent['layers']
            if not layers[1].weight.requires_grad:
                continue
            ops.wgrad3x3_body(ent['dz'], ent['acts'], [_grad_buf(m.weight) for m in layers[1:]],
                              dbs=[_grad_buf(m.bias) for m in layers[1:]])
        self.deferred_body = {}
        for buf, dzs in self.deferred_bias.values():
            if len(dzs) > 1 and same(dzs):
                ops.bias_grad_multi(dzs, buf, accumulate=True)
            else:
                D = dzs[0] if len(dzs) == 1 else torch.cat(dzs, 0)
                ops.bias_grad(D, buf, accumulate=True)
        self.deferred, self.deferred_bias = {}, {}


def _grad_buf(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class ConvCache:
    """Per-layer packed weights for the training step (forward, data-gradient and the
    space-to-depth embeddings), rebuilt when the parameter version changes.  The entries live
    ON the layer object: a process-wide table keyed by id(layer) would hand a new model the
    stale packs of a freed one whose id, storage address and version counters it re-uses."""

    @staticmethod
    def get(owner, key, version, build):
        store = owner.__dict__.setdefault('_tg_pack_cache', {})
        ent = store.get(key)
        if ent is None or ent[0] != version:
            ent = (version, build())
            store[key] = ent
        return ent[1]


_CACHE = ConvCache()


def _ver(w):
    return ops.param_version(w)


# ---------------------------------------------------------------------------
# conv3x3 (+bias, +activation, two-source input, residual)
# ---------------------------------------------------------------------------
_WINO_RULE = {}


def _prefers_wino(n, cin, cout, h, w):
    """tg_conv3x3_prefers_wino, memoised per shape (the tape asks ~2000 times per training step)."""
    if cout % 64 or cin < 16:
        return False
    key = (n, cin, cout, h, w)
    r = _WINO_RULE.get(key)
    if r is None:
        from .. import _lib as L
        r = _WINO_RULE[key] = bool(L.lib().tg_conv3x3_prefers_wino(n, cin, cout, h, w))
    return r


def conv3x3(tape, layer, x, act=NONE, x2=None, res=None, need_dx=True, need_dx2=True):
    w, b = layer.weight, layer.bias
    cout, cin = w.shape[0], w.shape[1]
    n_, _, h_, w_ = x.shape
    u = layer.packed_wino() if hasattr(layer, 'packed_wino') and _prefers_wino(n_, cin, cout, h_, w_) else None
    if u is not None:     # large layers (D's conv_in, VGG19 on the HR frames): Winograd form
        y = ops.conv3x3_wino(x, u, b, cin, cout, act, x2=x2, res=res)
    else:
        pk, ocb = layer.packed()
        y = ops.conv3x3(x, pk, b, cin, cout, ocb, act, x2=x2, res=res, ksplit=None if res is None else 1)
    if tape is None:
        return y
    c1 = x.shape[1]
    if act in (RELU, LRELU) and res is None:
        tape.act_outputs[id(y)] = act

    def bwd():
        g = tape.pop_grad(y)
        if g is None:
            return
        if res is not None:
            tape.add_grad(res, g.clone())
        if id(y) in tape.act_applied:            # the producer of g applied act'(.) already
            dz = g
        else:
            dz = ops.act_bwd(g, y, act, out=g) if act != NONE else g
        if w.requires_grad:
            gw = _grad_buf(w)
            # (the bias gradient rides on one of the layer's weight-gradient launches)
            tape.defer_wgrad(('w', id(layer), 0), dz, x, gw, 0, bias=None if x2 is not None else _grad_buf(b))
            if x2 is not None:
                tape.defer_wgrad(('w', id(layer), 1), dz, x2, gw, c1, bias=_grad_buf(b))
        wd = w.detach()
        if need_dx and c1 <= 4 and cout <= 64 and x2 is None:
            # data gradient onto an image (VGG's first conv): cout -> <=4 channels is the small
            # kernel's shape; its weights are the 180-degree rotated, transposed taps
            wdg = _CACHE.get(layer, ('dgs',), _ver(w),
                             lambda: wd.flip(2, 3).permute(1, 0, 2, 3).contiguous())
            tape.add_grad(x, ops.conv3x3_small(dz, wdg, None))
        elif need_dx and x2 is None and _prefers_wino(n_, cout, cin, h_, w_):
            ud = _CACHE.get(layer, ('dgu',), _ver(w), lambda: ops.pack_conv3x3_wino(wd.contiguous(), transposed=2))
            tape.add_grad(x, ops.conv3x3_wino(dz, ud, None, cout, cin, NONE))
        elif need_dx:
            pkd = _CACHE.get(layer, ('dg', 0), _ver(w), lambda: ops.pack_conv3x3_dgrad(
                wd[:, :c1].contiguous()))
            tape.add_grad(x, ops.conv3x3(dz, pkd[0], None, cout, c1, pkd[3], ksplit=None))
        if x2 is not None and need_dx2:
            pkd = _CACHE.get(layer, ('dg', 1), _ver(w), lambda: ops.pack_conv3x3_dgrad(
                wd[:, c1:].contiguous()))
            tape.add_grad(x2, ops.conv3x3(dz, pkd[0], None, cout, cin - c1, pkd[3], ksplit=None))
    tape.record(bwd)
    return y


def resblock(tape, conv1, conv2, x):
    """ResidualBlock (tecogan_nets.py:85-100): out = x + conv2(relu(conv1(x))) with a hand-fused
    backward -- two data-gradient launches per block instead of five:
      dZ1 = relu'(y1) * dgrad2(dOut)      ReLU mask applied in the conv epilogue
      dX  = dgrad1(dZ1) + dOut            the skip connection's gradient rides the residual input
    (the generic tape would run: clone for the skip, dgrad2, act_bwd, dgrad1, axpy)."""
    y1 = conv3x3(None, conv1, x, RELU)
    out = conv3x3(None, conv2, y1, NONE, res=x)
    if tape is None:
        return out
    w1, b1, w2, b2 = conv1.weight, conv1.bias, conv2.weight, conv2.bias
    c = w1.shape[0]

    def bwd():
        g = tape.pop_grad(out)
        if g is None:
            return
        if w2.requires_grad:
            tape.defer_wgrad(('w', id(conv2), 0), g, y1, _grad_buf(w2), 0, bias=_grad_buf(b2))
        pk2 = _CACHE.get(conv2, ('dg', 0), _ver(w2), lambda: ops.pack_conv3x3_dgrad(w2.detach().contiguous()))
        dz1 = ops.conv3x3(g, pk2[0], None, c, c, pk2[3], relu_mask=y1)
        if w1.requires_grad:
            tape.defer_wgrad(('w', id(conv1), 0), dz1, x, _grad_buf(w1), 0, bias=_grad_buf(b1))
        pk1 = _CACHE.get(conv1, ('dg', 0), _ver(w1), lambda: ops.pack_conv3x3_dgrad(w1.detach().contiguous()))
        tape.add_grad(x, ops.conv3x3(dz1, pk1[0], None, c, w1.shape[1], pk1[3], res=g))
    tape.record(bwd)
    return out


# ---------------------------------------------------------------------------
# SRNet's conv_in + residual blocks of one unrolled frame as ONE chained launch
# (tg_srnet_body_fwd / _bwd, csrc/tg_conv3x3_chain.hip), forward and reverse sweep.
# ---------------------------------------------------------------------------
class _ChainState:
    """Process-wide state of the chained launches: flag buffers per shape, the epoch counter, the
    pinned-host fault counter (the kernel adds to it with system scope when a workgroup gives up
    waiting for a neighbour) and the permanent fallback switch."""
    disabled = False
    poll_limit = 1 << 21
    epoch = 0
    flags = {}
    err = None
    supported = {}
    dirty = False           # chained launches were enqueued since the last SYNCHRONOUS look at the counter
    reported_epoch = 0      # launches up to this epoch are covered by a fault that has already been raised
    # Recovery from a transient fault (round 6, as in the frame plan: tg_frnet_plan_set_chain_rearm): after `rearm_wait`
    # clean ITERATIONS on the per-layer path (counted by chain_check, which every rank calls once per iteration, and a
    # fault reaches every rank through the all-reduced slot: the count is rank-symmetric) the chained body launches are
    # tried again; a fault of the re-armed launches doubles the wait.  rearm_first = 0: off for good (rounds 3-5).
    rearm_first = 64
    rearm_wait = 0
    clean_iters = 0
    rearms = 0

    @classmethod
    def note_fault(cls):
        cls.disabled = True
        cls.clean_iters = 0
        cls.rearm_wait = min(2 * cls.rearm_wait, 1 << 20) if cls.rearm_wait else cls.rearm_first

    @classmethod
    def note_clean_iteration(cls):
        if cls.disabled and cls.rearm_first > 0 and cls.rearm_wait > 0:
            cls.clean_iters += 1
            if cls.clean_iters >= cls.rearm_wait:
                cls.disabled = False
                cls.clean_iters = 0
                cls.rearms += 1

    @classmethod
    def buffers(cls, nlayer, n, h, w, device):
        from .. import _lib as L
        # one flag buffer per (shape, device, stream): two chained launches of one shape in flight at once
        # (a second model, a side stream, another device) must not overwrite each other's epochs
        key = (n, h, w, str(device), ops._stream())
        fl = cls.flags.get(key)
        need = L.lib().tg_conv3x3_chain_flag_ints(24, n, h, w)
        if fl is None or fl.numel() < need:
            fl = cls.flags[key] = torch.zeros(need, dtype=torch.int32, device=device)
        if cls.err is None:
            cls.err = torch.zeros(16, dtype=torch.int32).pin_memory()
        cls.epoch = cls.epoch + 1 if cls.epoch < 0x7fffffff else 1
        cls.dirty = True
        return fl, cls.err, cls.epoch

    @classmethod
    def parts(cls, n, h, w):
        """workgroups per tile the launcher uses for this shape (0: not supported): 4 needs the
        16 x 16 x 4 weight layout (ops.pack_conv3x3_m16), 1 / 2 the 64-channel-block layout."""
        key = (n, h, w, torch.cuda.current_device() if torch.cuda.is_available() else -1)
        r = cls.supported.get(key)
        if r is None:
            from .. import _lib as L
            r = cls.supported[key] = int(L.lib().tg_conv3x3_chain_supported(n, h, w, 64))
        return r

    MAX_LAYERS = 24      # RC_MAXL of tg_conv3x3_chain.hip: tg_srnet_body_fwd / _bwd walk 1 + 2 nb (+ 1) layers

    @classmethod
    def usable(cls, n, nf, cin0, h, w, nb):
        """nb: residual blocks of the body (a free yml parameter in the reference, tecogan_nets.py:108-116):
        beyond 11 the body does not fit one chained launch and runs one launch per layer."""
        if cls.disabled or nf > 64 or cin0 > 64 or nb < 1 or 2 * nb + 2 > cls.MAX_LAYERS:
            return False
        return cls.parts(n, h, w) > 0


def _distributed_world():
    try:
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    except ImportError:
        return 1


def guard_available(optim):
    """The device-side guard of the chained launches' fail-safe exists for this optimiser: its flat gradient buffer
    (with the fault slot behind the last tensor) is still what the parameters' .grad views point into."""
    return optim is not None and getattr(optim, 'fault_slot', None) is not None and optim._is_flat()


def stamp_fault(optim):
    """Before a network's gradient exchange / optimiser step: add 1 to the FAULT SLOT of its flat gradient
    buffer if a chained launch of this process has recorded a fault (a one-thread kernel reads the pinned
    counter).  The slot is all-reduced with the gradients and guards the Adam step on the device
    (tg_adam_step_guarded): gradients built on stale tiles are then applied on NO rank.

    An optimiser WITHOUT that slot (`Adam(flatten=False)`, views replaced by `net.to()` / `p.grad = None`) has no
    asynchronous guard: the step then falls back to the synchronous protocol ONCE -- wait for the device, look at the
    counter (on every rank: one max-all-reduce of the flag), raise BEFORE the update is applied -- and every later
    step runs one launch per layer, so the check is never needed again."""
    err = _ChainState.err
    if err is None or optim is None:
        return
    if guard_available(optim):
        ops.fault_to_slot(err, optim.fault_slot)
        return
    # An optimiser that HAD its flat buffer at construction (every rank compared that: BaseModel.agreement_vector) and
    # lost it on this rank only -- `p.grad = None`, a `net.to()` -- would enter the synchronous protocol below, whose
    # max-all-reduce the other ranks never join: a hang.  A loud rank-local error instead (the peers then fail in RCCL's
    # own time-out rather than waiting for ever).
    if getattr(optim, 'fault_slot', None) is not None and _distributed_world() > 1:
        from .. import _lib as L_
        raise L_.TecoganHipError(
            'an optimiser of a data-parallel run lost its flat gradient buffer on this rank (the .grad views were '
            'replaced): the device-side fault guard and the one-collective gradient exchange are rank-symmetric by '
            'construction -- rebuild the optimiser on every rank (Adam(..., flatten=True)) instead')
    if not _ChainState.dirty:
        return
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _ChainState.dirty = False
    _ChainState.disabled = True          # (for good: without the device-side guard every chained step would need this sync)
    _ChainState.rearm_wait = 0
    local = int(err[0]) != 0
    anywhere = local
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from ..utils import dist_utils
            anywhere = dist_utils.max_over_ranks(1.0 if local else 0.0, device=optim.params[0].device) != 0.0
    except ImportError:
        pass
    if anywhere:
        chain_check(0.0 if local else 1.0, counter=True)


def chain_check(slot_value=0.0, counter=True, epoch=None):
    """Raises if a chained launch recorded a fault: on any rank in the iteration whose scalars are being looked at
    (`slot_value`: the fault slot of the generator's gradient buffer after its all-reduce, read with that iteration's
    scalars), or -- `counter`, for callers that have just synchronised the device -- on this rank since the last check
    (pinned counter).  The training step resolves its scalars asynchronously (base_model: a later iteration may
    already be in flight and adding to the counter), so it passes counter=False and relies on the slot, which is
    exact per iteration; the counter is cleared only after a device synchronisation.  The iteration's generator
    update was dropped on every rank by the device-side guard; every later step runs one launch per layer.

    `epoch`: the chained-launch epoch at which the iteration was stamped.  An iteration that was already in flight
    when an earlier one raised carries the same fault in its slot (the pinned counter is cleared only by the raise):
    it is reported ONCE -- the call then returns True (update dropped, nothing raised).  Returns False when the
    iteration is clean."""
    err = _ChainState.err
    local = counter and err is not None and int(err[0]) != 0
    if not (local or slot_value != 0.0):
        if not counter:                  # (the per-iteration call of the training step)
            _ChainState.note_clean_iteration()
        return False
    if not local and epoch is not None and epoch <= _ChainState.reported_epoch:
        return True
    from .. import _lib as L
    if torch.cuda.is_available():
        torch.cuda.synchronize()            # nothing in flight may add to the counter after it is cleared
    lost = int(err[0]) if err is not None else 0
    if err is not None:
        err.zero_()
    _ChainState.note_fault()
    _ChainState.dirty = False
    _ChainState.reported_epoch = _ChainState.epoch     # everything enqueued so far has completed (synchronised above)
    raise L.TecoganHipError(
        'chained SRNet launch (training): %s timed out waiting for a neighbour tile; the results of this '
        'step are INVALID and its optimiser step was DROPPED on every rank (weights and Adam moments '
        'untouched).  Later steps run one launch per layer%s.'
        % (f'{lost} workgroup(s) of this rank' if lost else 'workgroups of another rank',
           f'; the chained launches are tried again after {_ChainState.rearm_wait} clean iterations'
           if _ChainState.rearm_first > 0 else ''))


def chain_epoch():
    """The epoch of the last chained launch enqueued by this process (see chain_check)."""
    return _ChainState.epoch


def srnet_body(tape, srnet, lr, tran):
    """conv_in + nb residual blocks (tecogan_nets.py:108-116, :141-143) of one frame: one launch
    forward, one launch for the whole reverse sweep of these 1 + 2*nb layers; the (dZ, X) pairs of
    the weight gradients are deferred exactly as the per-layer nodes defer them."""
    import ctypes
    from .. import _lib as L
    conv_in = srnet.conv_in['0']
    blocks = [(rb.conv['0'], rb.conv['2']) for rb in srnet.resblocks]
    layers = [conv_in] + [c for pair in blocks for c in pair]
    nb, nl = len(blocks), 1 + 2 * len(blocks)
    n, c_lr, h, w = lr.shape
    c_tran, nf = tran.shape[1], conv_in.cout
    fw = (L.PackedLayer * nl)()
    layout = 16 if _ChainState.parts(n, h, w) == 4 else 64
    # all 2 x (1 + 2nb) packs (forward, data gradient) of the body by two launches, cached on the network
    # until a parameter changes
    ver = (tuple(_ver(m.weight) for m in layers), layout, c_lr)      # every layer: a partial load / freeze must re-pack too

    def build():
        ws = [m.weight.detach() for m in layers]
        f = ops.chain_pack([(wt, 0, wt.shape[1], 0) for wt in ws], layout)
        # (the data gradient of conv_in only reaches `tran`: input channels [c_lr, c_lr + c_tran))
        d = ops.chain_pack([(ws[0], c_lr, c_tran, 2)] + [(wt, 0, wt.shape[1], 2) for wt in ws[1:]], layout)
        return f, d
    keep, dgp = _CACHE.get(srnet, ('body',), ver, build)
    for i, m in enumerate(layers):
        fw[i].w, fw[i].b = keep[i].data_ptr(), m.bias.data_ptr()
    acts = torch.empty(nl, n, nf, h, w, dtype=torch.float32, device=lr.device)
    flags, err, epoch = _ChainState.buffers(nl + 1, n, h, w, lr.device)
    st = ops._stream()
    L.check(L.lib().tg_srnet_body_fwd(fw, layout, nb, lr.data_ptr(), c_lr, tran.data_ptr(), c_tran, acts.data_ptr(),
                                      n, nf, h, w, flags.data_ptr(), err.data_ptr(), epoch,
                                      _ChainState.poll_limit, st), 'tg_srnet_body_fwd')
    out = acts[nl - 1]
    if tape is None:
        return out
    dz = torch.empty(nl, n, nf, h, w, dtype=torch.float32, device=lr.device)
    tape.reserved[id(out)] = dz[nl - 1]    # the gradient of the body's output lives in the block's last slot

    def bwd():
        g = tape.pop_grad(out)
        tape.reserved.pop(id(out), None)
        if g is None:
            return
        dg = (L.PackedLayer * nl)()
        hold = dgp
        for i in range(nl):
            dg[i].w = dgp[i].data_ptr()
        if g.data_ptr() != dz[nl - 1].data_ptr():
            dz[nl - 1].copy_(g)   # (a producer that could not write in place)
        d_tran = torch.empty(n, c_tran, h, w, dtype=torch.float32, device=g.device)
        fl, er, ep = _ChainState.buffers(nl + 1, n, h, w, g.device)
        L.check(L.lib().tg_srnet_body_bwd(dg, layout, nb, acts.data_ptr(), dz.data_ptr(), d_tran.data_ptr(),
                                          c_tran, n, nf, h, w, fl.data_ptr(), er.data_ptr(), ep,
                                          _ChainState.poll_limit, ops._stream()),
                'tg_srnet_body_bwd')
        tape.keep.append(hold)
        if conv_in.weight.requires_grad:
            gw = _grad_buf(conv_in.weight)
            tape.defer_wgrad(('w', id(conv_in), 0), dz[0], lr, gw, 0)
            tape.defer_wgrad(('w', id(conv_in), 1), dz[0], tran, gw, c_lr, bias=_grad_buf(conv_in.bias))
            # the residual-block convs' weight AND bias gradients are taken from the per-frame blocks by
            # one launch when the tape is flushed (conv_in's bias gradient rides on its own launch above)
            tape.defer_body(id(srnet), layers, acts, dz)
        tape.add_grad(tran, d_tran)
    tape.record(bwd)
    tape.keep.append(keep)
    return out


def conv3x3_small(tape, layer, x, act=NONE, up_src=None, up_mode=ops.UP_NONE, up_scale=1, res=None):
    """cout <= 4 head (flow[2] / conv_out).  `up_src` is data (no gradient).  `res` (data as well): the
    already up-sampled residual, added by the small-cout kernel's 4-row form (waves split the input
    channels); shapes it does not take go to the MFMA kernel (3 of 32 output columns used)."""
    w, b = layer.weight, layer.bias
    cout, cin = w.shape[0], w.shape[1]
    if res is not None and cin <= 64 and ops.conv3x3_small_res_ok(x, res):
        y = ops.conv3x3_small(x, w, b, act, res=res)
    elif res is not None and act in (NONE, RELU, LRELU):
        pk, ocb = layer.packed()
        y = ops.conv3x3(x, pk, b, cin, cout, ocb, act, res=res, ksplit=1)
    else:
        y = ops.conv3x3_small(x, w, b, act, up_src=up_src, up_mode=up_mode, up_scale=up_scale)
    if tape is None:
        return y

    def bwd():
        g = tape.pop_grad(y)
        if g is None:
            return
        dz = ops.act_bwd(g, y, act, out=g) if act != NONE else g
        if w.requires_grad:
            tape.defer_wgrad(('w', id(layer), 0), dz, x, _grad_buf(w), 0, bias=_grad_buf(b))
        fuse = id(x) in tape.relu_outputs        # x = relu(...): deliver dZ of that layer directly
        if ops.conv3x3_fewin_ok(dz, cin) and (not fuse or x.data_ptr() % 16 == 0):
            # few channels in, many out: the VALU kernel (the MFMA one pads K = 9 cout to a 72-deep chunk)
            wd = _CACHE.get(layer, ('dgw',), _ver(w),
                            lambda: w.detach().transpose(0, 1).flip(2, 3).contiguous())
            tape.add_grad(x, ops.conv3x3_fewin(dz, wd, relu_mask=x if fuse else None), masked=fuse)
        else:
            pkd = _CACHE.get(layer, ('dg', 0), _ver(w),
                             lambda: ops.pack_conv3x3_dgrad(w.detach().contiguous()))
            tape.add_grad(x, ops.conv3x3(dz, pkd[0], None, cout, cin, pkd[3], ksplit=1,
                                         relu_mask=x if fuse else None), masked=fuse)
    tape.record(bwd)
    return y


# ---------------------------------------------------------------------------
# ConvTranspose2d(k3,s2,p1,op1): forward = sub-pixel phase kernel; backward via
# the space-to-depth embedding (oy = 2*iy - 1 + ky  <=>  ky -> (phase, tap)).
# ---------------------------------------------------------------------------
_KT = {0: (1, 0), 1: (0, 1), 2: (1, 1)}          # convT: ky -> (py, ty)
_IDX = {}


def _embed_index(kind, a, b, device):
    """Flat gather indices of the space-to-depth embeddings, built once per shape:
    `fwd` maps every element of the embedded 3x3 weight to its source element (or to a
    trailing zero slot), `inv` maps every element of the original weight to its slot in the
    embedded tensor.  One index_select then replaces 9 / 16 strided slice copies."""
    key = (kind, a, b, str(device))
    ent = _IDX.get(key)
    if ent is None:
        table = _KT if kind == 'convt' else _K4
        k = len(table)
        src = torch.arange(a * b * k * k, dtype=torch.int64).view(a, b, k, k)
        if kind == 'convt':          # (ci, co, 3, 3) -> (ci, 4co, 3, 3)
            emb = torch.full((a, 4 * b, 3, 3), a * b * k * k, dtype=torch.int64)
            for ky, (py, ty) in table.items():
                for kx, (px, tx) in table.items():
                    ph = py * 2 + px
                    emb[:, ph * b:(ph + 1) * b, ty, tx] = src[:, :, ky, kx]
        else:                        # (co, ci, 4, 4) -> (co, 4ci, 3, 3)
            emb = torch.full((a, 4 * b, 3, 3), a * b * k * k, dtype=torch.int64)
            for ky, (py, ty) in table.items():
                for kx, (px, tx) in table.items():
                    ph = py * 2 + px
                    emb[:, ph * b:(ph + 1) * b, ty, tx] = src[:, :, ky, kx]
        fwd = emb.reshape(-1)
        inv = torch.empty(a * b * k * k, dtype=torch.int64)
        valid = fwd < a * b * k * k
        inv[fwd[valid]] = torch.nonzero(valid).reshape(-1)
        ent = _IDX[key] = (fwd.to(device), inv.to(device))
    return ent


def _convt_embed(wt):
    """(ci, co, 3, 3) -> conv weight (cout_op = ci, cin_op = 4*co, 3, 3) acting on s2d(dY)."""
    ci, co = wt.shape[:2]
    fwd, _ = _embed_index('convt', ci, co, wt.device)
    return ops.index_gather(wt.contiguous(), fwd).view(ci, 4 * co, 3, 3)   # out-of-range slot = 0


def convt3x3s2(tape, layer, x, act=RELU):
    w, b = layer.weight, layer.bias            # (ci, co, 3, 3)
    ci, co = w.shape[:2]
    pk, _ = layer.packed()
    y = ops.convt3x3s2(x, pk, b, co, act)
    if tape is None:
        return y
    if act == RELU:
        tape.relu_outputs.add(id(y))

    def bwd():
        g = tape.pop_grad(y)
        if g is None:
            return
        premasked = act == RELU and id(y) not in tape.unmasked       # every contribution came masked
        dz = ops.act_bwd(g, y, act, out=g) if (act != NONE and not premasked) else g
        fuse = id(x) in tape.relu_outputs
        nx, _, hx, wx = x.shape
        direct = co <= 64 and ci <= 64 and dz.is_contiguous() and x.is_contiguous()
        s = None
        if ops.conv3x3s2_supported(nx, co, ci, hx, wx):
            # small frames: the gradient taken directly as a stride-2 conv of dZ (K = 9 co) -- 11 us
            # instead of 27 us for the 32-chunk phased form on s2d(dZ)
            wk = _CACHE.get(layer, ('ctd',), _ver(w), lambda: ops.pack_conv3x3(w.detach().contiguous(), ocb=64)[0])
            buf = tape.reserved.get(id(x)) if tape.grad(x) is None else None
            tape.add_grad(x, ops.conv3x3s2(dz, wk, co, ci, relu_mask=x if fuse else None, out=buf), masked=fuse)
        else:
            s = ops.space_to_depth(dz, 2)                          # (n, 4co, h, w)
            we = _CACHE.get(layer, ('cte',), _ver(w), lambda: ops.pack_conv3x3(_convt_embed(w.detach())))
            if co % 8 == 0:    # phase py = 1 owns tap rows {0, 1}, py = 0 only {1}: 9 of 36 taps are non-zero
                tape.add_grad(x, ops.conv3x3_phased(s, we[0], 4 * co, ci, we[3], 1, co, ops.TAPS_1, ops.TAPS_01,
                                                    relu_mask=x if fuse else None), masked=fuse)
            else:
                tape.add_grad(x, ops.conv3x3(s, we[0], None, 4 * co, ci, we[3], ksplit=1,
                                             relu_mask=x if fuse else None), masked=fuse)
        if w.requires_grad:
            if direct:
                # dW straight from dZ (tg_wgrad3x3_convt_multi): no s2d copy, no embedded gradient to gather back
                tape.defer_wgrad(('ctw', id(layer)), x, dz, _grad_buf(w), convt=True, post=_grad_buf(b))
            else:
                if s is None:
                    s = ops.space_to_depth(dz, 2)

                def post(ge):                                      # G[ci][(ph,co)][ty][tx]
                    _, inv = _embed_index('convt', ci, co, ge.device)
                    ops.index_gather(ge, inv, out=_grad_buf(w), accumulate=True)
                tape.defer_wgrad(('ct', id(layer)), x, s, None, 0, post, phased=(co, ops.TAPS_1, ops.TAPS_01))
                tape.defer_bias(_grad_buf(b), dz)
    tape.record(bwd)
    return y


# ---------------------------------------------------------------------------
# Conv2d(k4, s2, p1, no bias) of the discriminator blocks through s2d:
#   2*oy - 1 + ky = 2*(oy + ty - 1) + py
# ---------------------------------------------------------------------------
_K4 = {0: (1, 0), 1: (0, 1), 2: (1, 1), 3: (0, 2)}   # ky -> (py, ty)


def direct_conv4():
    """TG_CONV4_DIRECT=0: the embedded form everywhere (lab / tests).  Read per call, so a test can toggle it."""
    return os.environ.get('TG_CONV4_DIRECT', '1') != '0'


def _conv4_embed(w4):
    """(co, ci, 4, 4) -> (co, 4*ci, 3, 3) acting on s2d(x, 2)."""
    co, ci = w4.shape[:2]
    fwd, _ = _embed_index('conv4', co, ci, w4.device)
    return ops.index_gather(w4.contiguous(), fwd).view(co, 4 * ci, 3, 3)


def conv4x4s2(tape, holder, x, need_dx=True):
    """holder.weight: (co, ci, 4, 4).  Returns (n, co, h/2, w/2).
    Forward and data gradient run on the direct K = 16 ci kernels (tg_conv4x4s2_fwd / _dgrad, round 4) wherever
    `tg_conv4x4s2_supported` says so: ci, co multiples of 64 and an input 64 or more wide (a multiple of 64) on the
    row-tile kernels, 32 / 16 wide on the small-map kernels (folded pixel tiles, split-K + a summing launch).  Every
    other shape and the weight gradient use the embedding into a phase-masked 3x3 conv on space_to_depth(x, 2) --
    that copy is then made in backward, only when the weights take a gradient."""
    w = holder.weight
    co, ci = w.shape[:2]
    n, _, h_, w_ = x.shape
    direct = direct_conv4() and ops.conv4x4s2_supported(n, ci, co, h_, w_)
    sparse = ci % 8 == 0 and co > 32
    if direct:
        pk4 = _CACHE.get(holder, ('c4x',), _ver(w), lambda: ops.pack_conv4x4s2(w.detach().contiguous()))
        y = ops.conv4x4s2(x, pk4[0], co)
        s_held = [None]
    else:
        s_held = [ops.space_to_depth(x, 2)]
        pk = _CACHE.get(holder, ('c4f',), _ver(w), lambda: ops.pack_conv3x3(_conv4_embed(w.detach())))
        # phase coordinate 1 owns tap rows {0, 1}, coordinate 0 rows {1, 2}: 4 of 9 taps per channel
        if sparse:
            y = ops.conv3x3_phased(s_held[0], pk[0], 4 * ci, co, pk[3], 1, ci, ops.TAPS_12, ops.TAPS_01)
        else:
            y = ops.conv3x3(s_held[0], pk[0], None, 4 * ci, co, pk[3])
    if tape is None:
        return y

    def bwd():
        g = tape.pop_grad(y)
        if g is None:
            return
        if w.requires_grad:
            def post(ge):
                _, inv = _embed_index('conv4', co, ci, ge.device)
                ops.index_gather(ge, inv, out=_grad_buf(w), accumulate=True)
            s = s_held[0] if s_held[0] is not None else ops.space_to_depth(x, 2)
            tape.defer_wgrad(('c4', id(holder)), g, s, None, 0, post, phased=(ci, ops.TAPS_12, ops.TAPS_01))
        if need_dx:
            a = tape.act_outputs.get(id(x))
            fuse = a is not None and tape.grad(x) is None
            if direct:
                # x = act(conv(.)): the epilogue applies act'(x) (one pass over the tensor instead of three)
                fuse = fuse and a in (ops.ACT_RELU, ops.ACT_LRELU02)
                gx = ops.conv4x4s2_dgrad(g, pk4[1], ci, act_y=x if fuse else None, act=a if fuse else ops.ACT_NONE)
                tape.add_grad(x, gx, act_applied=fuse)
                return
            pkd = _CACHE.get(holder, ('c4d',), _ver(w),
                             lambda: ops.pack_conv3x3_dgrad(_conv4_embed(w.detach())))
            if ci % 64 == 0:    # rot180 taps: coordinate 1 owns rows {1, 2}, coordinate 0 rows {0, 1}
                ds = ops.conv3x3_phased(g, pkd[0], co, 4 * ci, pkd[3], 2, ci, ops.TAPS_01, ops.TAPS_12)
            else:
                ds = ops.conv3x3(g, pkd[0], None, co, 4 * ci, pkd[3], ksplit=1)
            if fuse:
                # x = act(conv(.)): the depth-to-space pass applies act'(x) on the way out (one pass over
                # the HR tensor instead of three)
                gx, fused = ops.depth_to_space(ds, 2, act_y=x, act=a)
                tape.add_grad(x, gx, act_applied=fused)
            else:
                tape.add_grad(x, ops.depth_to_space(ds, 2))
    tape.record(bwd)
    return y


# ---------------------------------------------------------------------------
# pointwise / gather ops
# ---------------------------------------------------------------------------
def maxpool2(tape, x):
    y = ops.maxpool2(x)
    if tape is not None:
        def bwd():
            g = tape.pop_grad(y)
            if g is not None:
                tape.add_grad(x, ops.maxpool2_bwd(x, g))
        tape.record(bwd)
    return y


def upsample(tape, x, scale, up_mode, mul=1.0):
    y = ops.upsample(x, scale, up_mode, mul)
    if tape is not None:
        def bwd():
            g = tape.pop_grad(y)
            if g is not None:
                tape.add_grad(x, ops.upsample_bwd(g, scale, up_mode, mul))
        tape.record(bwd)
    return y


def backward_warp(tape, x, flow, need_dimg=True, need_dflow=True, dflow_out=None, s2d=1):
    """`dflow_out`: a callable returning the buffer the flow gradient is written into (a slice of
    a larger gradient tensor the caller owns, allocated when backward reaches it) instead of
    the gradient being deposited on the tape.  `s2d` > 1: the result is space_to_depth(warp, s2d) -- the
    unroll's warp -> space_to_depth pair (tecogan_nets.py:208-212) as one launch each way."""
    y = ops.backward_warp(x, flow) if s2d == 1 else ops.backward_warp_s2d(x, flow, s2d)
    if tape is not None and (need_dimg or need_dflow):
        def bwd():
            g = tape.pop_grad(y)
            if g is None:
                return
            cur = tape.grad(x) if need_dimg else None
            if cur is not None and cur.is_contiguous() and id(x) not in tape.act_applied:
                # x has a gradient already (its own loss term): the scatter adds into it -- no memset of a new
                # tensor, no accumulation pass
                tape.unmasked.add(id(x))
                _, dflow = ops.backward_warp_bwd(x, flow, g, True, need_dflow,
                                                 dflow_out() if dflow_out is not None else None, dimg_acc=cur, s2d=s2d)
                dimg = None
            else:
                dimg, dflow = ops.backward_warp_bwd(x, flow, g, need_dimg, need_dflow,
                                                    dflow_out() if dflow_out is not None else None, s2d=s2d)
            if need_dimg and dimg is not None:
                tape.add_grad(x, dimg)
            if need_dflow and dflow_out is None:
                tape.add_grad(flow, dflow)
        tape.record(bwd)
    return y


def view(tape, x, shape):
    """Reshape of a contiguous tensor (gradients are keyed by tensor identity, so the
    re-viewed tensor needs its own node)."""
    y = x.view(shape)
    if tape is not None:
        def bwd():
            g = tape.pop_grad(y)
            if g is not None:
                tape.add_grad(x, g.view(x.shape))
        tape.record(bwd)
    return y


def channel_norm(tape, x, mean, std):
    """(x - mean[c]) / std[c]  (vgg_nets.py:29)."""
    y = ops.channel_norm(x, mean, std)
    if tape is not None:
        def bwd():
            g = tape.pop_grad(y)
            if g is not None:
                tape.add_grad(x, ops.channel_norm(g, None, std))
        tape.record(bwd)
    return y


def space_to_depth(tape, x, scale):
    y = ops.space_to_depth(x, scale)
    if tape is not None:
        def bwd():
            g = tape.pop_grad(y)
            if g is not None:
                tape.add_grad(x, ops.depth_to_space(g, scale))
        tape.record(bwd)
    return y


def _distributed():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def bn_lrelu(tape, bn, x, need_dx=True, sync=None, groups=1):
    """BatchNorm2d (train mode) + LeakyReLU(0.2); bn holds weight/bias/running stats.
    Under torch.distributed (or sync=True) the statistics are global (SyncBatchNorm).
    groups > 1: x stacks that many INDEPENDENT batches along n (the critic's real + fake pair pass):
    batch statistics per group, the running statistics updated group after group -- exactly what the
    reference's separate passes do -- and, data parallel, ONE collective per layer for all groups."""
    if sync is None:
        sync = _distributed()
    n = x.shape[0]
    if groups > 1:
        assert n % groups == 0
        per = n // groups
        if sync:
            y, stats = ops.sync_bn_lrelu_train_fwd_groups(x, groups, bn.weight, bn.bias, bn.running_mean,
                                                          bn.running_var)
        else:
            y, stats = torch.empty_like(x), []
            for g in range(groups):
                _, mean, invstd = ops.bn_lrelu_train_fwd(x[g * per:(g + 1) * per], bn.weight, bn.bias,
                                                         bn.running_mean, bn.running_var,
                                                         out=y[g * per:(g + 1) * per])
                stats.append((mean, invstd, None))
        for _ in range(groups):
            bn.count_pass() if hasattr(bn, 'count_pass') else bn.num_batches_tracked.add_(1)
        if tape is not None:
            def bwd_groups():
                g_ = tape.pop_grad(y)
                if g_ is None:
                    return
                train = bn.weight.requires_grad
                gw = _grad_buf(bn.weight) if train else None
                gb = _grad_buf(bn.bias) if train else None
                if sync:
                    dx = ops.sync_bn_lrelu_train_bwd_groups(x, y, g_, groups, bn.weight, stats, gw, gb, need_dx)
                else:
                    dx = torch.empty_like(x) if need_dx else None
                    for g in range(groups):
                        sl = slice(g * per, (g + 1) * per)
                        ops.bn_lrelu_train_bwd(x[sl], y[sl], g_[sl], bn.weight, stats[g][0], stats[g][1], gw, gb,
                                               need_dx, dx_out=dx[sl] if need_dx else None)
                if need_dx:
                    tape.add_grad(x, dx)
            tape.record(bwd_groups)
        return y
    if sync:
        y, mean, invstd, count = ops.sync_bn_lrelu_train_fwd(x, bn.weight, bn.bias, bn.running_mean,
                                                             bn.running_var)
    else:
        y, mean, invstd = ops.bn_lrelu_train_fwd(x, bn.weight, bn.bias, bn.running_mean,
                                                 bn.running_var)
    bn.count_pass() if hasattr(bn, 'count_pass') else bn.num_batches_tracked.add_(1)
    if tape is not None:
        def bwd():
            g = tape.pop_grad(y)
            if g is None:
                return
            train = bn.weight.requires_grad
            gw = _grad_buf(bn.weight) if train else None
            gb = _grad_buf(bn.bias) if train else None
            if sync:
                dx = ops.sync_bn_lrelu_train_bwd(x, y, g, bn.weight, mean, invstd, count, gw, gb,
                                                 need_dx)
            else:
                dx = ops.bn_lrelu_train_bwd(x, y, g, bn.weight, mean, invstd, gw, gb, need_dx)
            if need_dx:
                tape.add_grad(x, dx)
        tape.record(bwd)
    return y


def linear1(tape, lin, x, need_dx=True):
    y = ops.linear1_fwd(x, lin.weight, lin.bias)
    if tape is not None:
        def bwd():
            g = tape.pop_grad(y)
            if g is None:
                return
            train = lin.weight.requires_grad
            dx = ops.linear1_bwd(x, lin.weight, g, _grad_buf(lin.weight) if train else None,
                                 _grad_buf(lin.bias) if train else None, need_dx)
            if need_dx:
                tape.add_grad(x, dx)
        tape.record(bwd)
    return y
