"""CPU checks of the host-side training logic that needs no kernel: the
space-to-depth weight embeddings that let the transposed convs and the
discriminator's 4x4/stride-2 convs reuse the 3x3/stride-1 MFMA kernels
(models/train_graph.py), verified against torch's own convs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tecogan_pytorch_amd  # noqa: F401
from tecogan_pytorch_amd.models import train_graph as TG
from oracle import tecogan_oracle as O


def rs(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).uniform(-1, 1, shape).astype(np.float32))


def _embed_on_host(kind, w):
    """Applies the product's slot table (TG._embed_index) with torch on the CPU; the product applies
    the same table with tg_index_gather on the GPU (tests/test_hip_train_ops.py checks that)."""
    a, b = w.shape[:2]
    fwd, _ = TG._embed_index(kind, a, b, w.device)
    flat = torch.cat([w.reshape(-1), w.new_zeros(1)])
    return flat.index_select(0, fwd).view(a, 4 * b, 3, 3)


def test_conv4x4s2_equals_conv3x3_on_space_to_depth():
    x, w = rs(1, (2, 5, 12, 16)), rs(2, (7, 5, 4, 4))
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    we = _embed_on_host('conv4', w)
    assert we.shape == (7, 20, 3, 3)
    out = F.conv2d(O.space_to_depth(x, 2), we, None, padding=1)
    assert (out - ref).abs().max() <= 1e-5
    # exactly 16 of the 36 embedded taps per (co, ci) are populated
    assert int((we != 0).sum()) == 7 * 5 * 16


def test_convt_data_gradient_equals_conv3x3_on_space_to_depth():
    """dX of ConvTranspose2d(k3,s2,p1,op1) = conv3x3(s2d(dY), embed(W))."""
    x = rs(1, (2, 6, 5, 7)).requires_grad_(True)
    w = rs(2, (6, 4, 3, 3))
    y = F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1)
    dy = rs(3, tuple(y.shape))
    y.backward(dy)
    we = _embed_on_host('convt', w)
    assert we.shape == (6, 16, 3, 3)
    dx = F.conv2d(O.space_to_depth(dy, 2), we, None, padding=1)
    assert (dx - x.grad).abs().max() <= 1e-5


def test_embedded_weight_gradient_extraction_roundtrip():
    """The (ky,kx) -> (phase, tap) maps are bijections onto the populated entries."""
    for table, k in ((TG._KT, 3), (TG._K4, 4)):
        seen = set(table.values())
        assert len(seen) == k and all(0 <= py <= 1 and 0 <= t <= 2 for py, t in seen)


def test_tape_accumulates_and_orders():
    """Tape semantics without any kernel: closures run in reverse, grads keyed by identity."""
    tape = TG.Tape()
    order = []
    a, b = torch.zeros(1), torch.zeros(1)
    tape.record(lambda: order.append('first'))
    tape.record(lambda: order.append('second'))
    g = torch.ones(1)
    tape.grads[id(a)] = g
    assert tape.grad(a) is g and tape.grad(b) is None
    assert tape.pop_grad(a) is g and tape.grad(a) is None
    tape.backward()
    assert order == ['second', 'first'] and tape.nodes == []


def test_adam_state_dict_round_trip():
    """Optimiser-state checkpoint (the resume half the reference leaves as a TODO): moments,
    step count and the learning rate survive a save / load; mismatched networks are refused."""
    import pytest
    from tecogan_pytorch_amd.models.optim import Adam
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    a = Adam(ps, lr=1e-4, betas=(0.9, 0.99))
    a.steps = 7
    a.state[id(ps[0])] = (torch.full((3, 4), 0.5), torch.full((3, 4), 0.25))
    sd = a.state_dict()
    assert sd['steps'] == 7 and sd['state'][1] is None
    qs = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))]
    b = Adam(qs, lr=1.0)
    b.load_state_dict(sd)
    # the learning rate in force at the checkpoint is restored; the other hyper-parameters are
    # the CURRENT configuration's (a resumed run follows its yml)
    assert b.steps == 7 and b.param_groups[0]['lr'] == 1e-4 and b.param_groups[0]['betas'] == (0.9, 0.999)
    m, v = b.state[id(qs[0])]
    assert torch.equal(m, torch.full((3, 4), 0.5)) and torch.equal(v, torch.full((3, 4), 0.25))
    assert id(qs[1]) not in b.state
    with pytest.raises(ValueError):
        Adam(qs[:1], lr=1.0).load_state_dict(sd)
    with pytest.raises(ValueError):
        Adam([torch.nn.Parameter(torch.zeros(4, 3)), qs[1]], lr=1.0).load_state_dict(sd)


def test_chain_check_reports_a_fault_once_and_flags_the_iterations_behind_it():
    """iteration k + 1 is already stamped when iteration k's fault is raised; its later resolve must not
    raise a second time with a misleading text -- chain_check(epoch=...) returns True (update dropped, already
    reported) for every iteration stamped up to the epoch the raise covered."""
    import torch
    from tecogan_pytorch_amd import _lib
    from tecogan_pytorch_amd.models import train_graph as TG
    S = TG._ChainState
    saved = (S.err, S.epoch, S.disabled, S.dirty, S.reported_epoch)
    try:
        S.err, S.epoch, S.disabled, S.dirty, S.reported_epoch = torch.zeros(16, dtype=torch.int32), 7, False, True, 0
        assert TG.chain_check(0.0, counter=False, epoch=6) is False            # clean iteration
        with pytest.raises(_lib.TecoganHipError, match='another rank'):
            TG.chain_check(1.0, counter=False, epoch=6)                         # iteration k: raises, covers epochs <= 7
        assert S.disabled and not S.dirty and S.reported_epoch == 7
        assert TG.chain_check(1.0, counter=False, epoch=7) is True              # iteration k + 1: dropped, not raised again
        S.epoch = 9
        with pytest.raises(_lib.TecoganHipError):
            TG.chain_check(1.0, counter=False, epoch=9)                         # a NEW fault still raises
        S.err[0] = 3
        with pytest.raises(_lib.TecoganHipError, match='3 workgroup'):
            TG.chain_check(0.0, counter=True)                                   # the synchronous form reads the counter
        assert int(S.err[0]) == 0
    finally:
        S.err, S.epoch, S.disabled, S.dirty, S.reported_epoch = saved


def test_log_queue_keeps_the_entries_behind_a_faulting_one():
    """_drain_log_queue resolves one entry at a time; an entry whose checks raise is consumed, the entries
    behind it stay queued (with their pinned buffers) for the next call."""
    import torch
    from collections import OrderedDict
    from tecogan_pytorch_amd.models.base_model import BaseModel

    class Ev:
        def query(self): return True
        def synchronize(self): pass
    m = BaseModel.__new__(BaseModel)
    m._log_queue, m._log_pinned, m._pending_log = [], [], None
    m._log_dict, m.running_log_dict, m.log_decay = OrderedDict(), OrderedDict(), 0.5

    def ok(v):
        return OrderedDict(x=v[0])

    def bad(v):
        raise RuntimeError('fault')
    m._log_queue = [(Ev(), torch.tensor([1.0]), ok), (Ev(), torch.tensor([2.0]), bad), (Ev(), torch.tensor([3.0]), ok)]
    with pytest.raises(RuntimeError, match='fault'):
        m._drain_log_queue()
    assert len(m._log_queue) == 1 and m.running_log_dict['x'] == 1.0
    m._drain_log_queue()
    assert not m._log_queue and m.running_log_dict['x'] == 2.0 and len(m._log_pinned) == 3
