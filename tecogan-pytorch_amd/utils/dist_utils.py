"""Rank helpers with the reference's names (codes/utils/dist_utils.py) plus the
two collectives the inference path needs.  One process per GPU; backend "nccl"
is RCCL on ROCm, "gloo" is used by the CPU tests.  The inference data path has
NO collective: sequences are independent and sharded round-robin
(codes/main.py:169: `for idx in range(rank, num_seq, world_size)`); only
timing (max over ranks) and metric sums (reduce to rank 0,
metric_calculator.py:99-102) cross ranks."""
import functools
import os

import torch
import torch.distributed as dist


def init_dist(opt, local_rank, backend='nccl', device=None):
    """env:// rendezvous as under torch.distributed.run (dist_utils.py:8-24).  `device`
    overrides the default ('cuda' for nccl, 'cpu' for gloo): a gloo group with device 'cuda'
    stages the exchanged tensors through the host (two test ranks on one GPU)."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl' or device == 'cuda':
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend)
    rank, world_size = get_dist_info()
    opt.update({'dist': True, 'device': device or ('cuda' if backend == 'nccl' else 'cpu'),
                'local_rank': local_rank, 'world_size': world_size, 'rank': rank})


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def master_only(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        rank, _ = get_dist_info()
        if rank == 0:
            return func(*args, **kwargs)
    return wrapper


def shard_indices(num_items, rank=None, world_size=None):
    """Sequence indices this rank processes (main.py:93,169 round-robin)."""
    if rank is None or world_size is None:
        rank, world_size = get_dist_info()
    return list(range(rank, num_items, world_size))


def max_over_ranks(value, device='cpu'):
    """Wall-clock style scalar: MAX over ranks (bench.py timing contract)."""
    rank, world = get_dist_info()
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def assert_ranks_agree(values, what, device='cpu'):
    """Every rank passes the same small vector of INTEGERS (library version, kernel capabilities, option switches);
    ONE all-gather compares them and every rank raises the same error when they differ.  The training step's exchange
    pattern depends on these (a rank that pairs D's real / fake pass issues 19 collectives per step, one that does not
    issues 27; a rank whose chained launches are unavailable stamps no fault slot): a mixed build would otherwise
    surface as a hang inside the first gradient bucket.  Reference: DDP checks parameter shapes across ranks at
    construction (base_model.py:130-136 wraps with DistributedDataParallel, which verifies them)."""
    rank, world = get_dist_info()
    vals = [int(v) for v in values]
    if world == 1:
        return vals
    t = torch.tensor(vals, dtype=torch.int64, device=device)
    if t.is_cuda and dist.get_backend() == 'gloo':
        t = t.cpu()
    got = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(got, t)
    rows = [g.tolist() for g in got]
    if any(r != rows[0] for r in rows):
        bad = {i: r for i, r in enumerate(rows) if r != rows[0]}
        raise RuntimeError(f'ranks disagree on {what}: rank 0 has {rows[0]}, differing ranks {bad} -- '
                           f'a mixed build / environment; refusing to start (the exchange patterns would not match)')
    return vals


def reduce_sum_to_master(values, device='cpu'):
    """Per-sequence metric vectors: SUM-reduce to rank 0 (each sequence is
    non-zero on exactly one rank), metric_calculator.py:68-117."""
    rank, world = get_dist_info()
    t = torch.as_tensor(values, dtype=torch.float64, device=device).clone()
    if world > 1:
        dist.reduce(t, dst=0)
    return t


# ---------------------------------------------------------------------------
# Transport of the training step's exchanges.  Default: torch.distributed (backend "nccl" IS
# RCCL on ROCm; collectives are enqueued on RCCL's own stream and ordered against the
# compute stream by events, so `async_op=True` overlaps them with compute).  A gloo group
# (CPU tests, or two test ranks sharing one GPU -- RCCL refuses duplicate devices) stages
# device tensors through the host.  TECOGAN_COMM=c_abi routes the same calls through the
# C-ABI communicator (include/tecogan_hip.h tg_comm_*, tg_allreduce_sum_f32,
# tg_allgather_f32); the 128-byte unique id is shipped over the existing process group.
# ---------------------------------------------------------------------------
class _Done:
    def wait(self):
        return True


_C_COMM = None


def _c_comm():
    """Lazily created C-ABI communicator (one per process)."""
    global _C_COMM
    if _C_COMM is None:
        import ctypes
        from .. import _lib as L
        lib = L.lib()
        rank, world = get_dist_info()
        ident = [None]
        if rank == 0:
            buf = (ctypes.c_uint8 * 128)()
            L.check(lib.tg_comm_get_unique_id(buf), 'tg_comm_get_unique_id')
            ident[0] = bytes(buf)
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(ident[0])
        handle = ctypes.c_void_p()
        L.check(lib.tg_comm_init_rank(buf, world, rank, ctypes.byref(handle)), 'tg_comm_init_rank')
        _C_COMM = handle
    return _C_COMM


def destroy_c_comm():
    global _C_COMM
    if _C_COMM is not None:
        from .. import _lib as L
        L.lib().tg_comm_destroy(_C_COMM)
        _C_COMM = None


def _use_c_abi(t):
    return os.environ.get('TECOGAN_COMM', '') == 'c_abi' and t.is_cuda and t.dtype == torch.float32


def _staged(t):
    """gloo cannot address device memory: exchange a host copy."""
    return t.is_cuda and dist.get_backend() == 'gloo'


# collectives issued by this process (bench.py --gpus N reports them per training step: "rccl_comm_count")
COMM_COUNTS = {'all_reduce': 0, 'all_gather': 0, 'bytes': 0}


def all_reduce_sum_(t, async_op=False):
    """In-place SUM of `t` over ranks.  Returns an object with .wait() (already complete
    unless async_op on an RCCL group: then wait() orders the CURRENT stream after the
    collective without blocking the host)."""
    rank, world = get_dist_info()
    if world > 1:
        COMM_COUNTS['all_reduce'] += 1
        COMM_COUNTS['bytes'] += t.numel() * t.element_size()
    if _use_c_abi(t):
        from .. import _lib as L
        L.check(L.lib().tg_allreduce_sum_f32(_c_comm(), t.data_ptr(), t.numel(),
                                             torch.cuda.current_stream().cuda_stream),
                'tg_allreduce_sum_f32')
        return _Done()
    if world == 1:
        return _Done()
    if _staged(t):
        h = t.detach().cpu()
        dist.all_reduce(h)
        t.copy_(h)
        return _Done()
    if async_op:
        return dist.all_reduce(t, async_op=True)
    dist.all_reduce(t)
    return _Done()


def all_gather_flat(t):
    """(world, t.numel()) tensor holding every rank's `t`, rank order."""
    rank, world = get_dist_info()
    flat = t.reshape(-1)
    if world > 1:
        COMM_COUNTS['all_gather'] += 1
        COMM_COUNTS['bytes'] += flat.numel() * flat.element_size()
    if _use_c_abi(t):
        from .. import _lib as L
        out = torch.empty(max(world, 1) * flat.numel(), dtype=t.dtype, device=t.device)
        L.check(L.lib().tg_allgather_f32(_c_comm(), flat.data_ptr(), out.data_ptr(), flat.numel(),
                                         torch.cuda.current_stream().cuda_stream), 'tg_allgather_f32')
        return out.view(max(world, 1), -1)
    if world == 1:
        return flat.view(1, -1)
    if _staged(t):
        h = flat.detach().cpu()
        out = torch.empty(world * h.numel(), dtype=h.dtype)
        _gather_list(out, h, world)
        return out.to(t.device).view(world, -1)
    out = torch.empty(world * flat.numel(), dtype=t.dtype, device=t.device)
    if dist.get_backend() == 'gloo':
        _gather_list(out, flat.contiguous(), world)
    else:
        dist.all_gather_into_tensor(out, flat.contiguous())
    return out.view(world, -1)


def _gather_list(out, h, world):
    parts = [torch.empty_like(h) for _ in range(world)]
    dist.all_gather(parts, h)
    torch.cat(parts, out=out)


def broadcast_(t, src=0):
    """Rank `src`'s values into every rank's `t` (DDP's initial parameter / buffer sync,
    torch/nn/parallel/distributed.py _sync_module_states, reached from base_model.py:130-136)."""
    rank, world = get_dist_info()
    if world == 1:
        return t
    if _staged(t):
        h = t.detach().cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)
    return t


class GradBucket:
    """One network's gradient exchange: flatten -> all-reduce(SUM) -> mean written back.
    start() launches the collective (asynchronously on an RCCL group) and returns at once;
    finish() orders the current stream after it and scatters the mean into the .grad tensors.
    10.4 MB (G) / 3.3 MB (D): latency bound over xGMI, so ONE collective per network."""

    def __init__(self, grads, flat=None):
        self.grads = grads
        self.flat = flat          # the grads are already views of this buffer (no copies)
        self.work = None
        self._owned = None

    def start(self):
        rank, world = get_dist_info()
        if (world == 1 and not os.environ.get('TECOGAN_COMM')) or not self.grads:
            return self
        buf = self.flat
        if buf is None:
            buf = self._owned = torch.cat([g.reshape(-1) for g in self.grads])
        self.work = all_reduce_sum_(buf, async_op=True)
        return self

    def finish(self, scale_fn=None):
        if self.work is None:
            return
        self.work.wait()
        rank, world = get_dist_info()
        if self.flat is not None:
            if world > 1:
                if scale_fn is not None:
                    scale_fn(self.flat, None, world)     # in-place mean of the bucket
                else:
                    self.flat.div_(world)
        else:
            off = 0
            for g in self.grads:
                k = g.numel()
                src = self._owned[off:off + k]
                if scale_fn is not None:
                    scale_fn(g, src, world)
                else:
                    g.copy_(src.view_as(g)).div_(world)
                off += k
        self.work = self._owned = None


def allreduce_mean_(grads, scale_fn=None):
    """DDP gradient exchange for one network, blocking form: GradBucket.start + finish.
    `scale_fn(dst, src_flat_slice_or_None, world)` does dst = src / world on the device
    (tg_div_scalar: the IEEE division DDP performs); default = torch ops (CPU tensors in the gloo tests)."""
    rank, world = get_dist_info()
    if world == 1 or not grads:
        return
    GradBucket(grads).start().finish(scale_fn)
