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


def init_dist(opt, local_rank, backend='nccl'):
    """env:// rendezvous as under torch.distributed.run (dist_utils.py:8-24)."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend)
    rank, world_size = get_dist_info()
    opt.update({'dist': True, 'device': 'cuda' if backend == 'nccl' else 'cpu',
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


def reduce_sum_to_master(values, device='cpu'):
    """Per-sequence metric vectors: SUM-reduce to rank 0 (each sequence is
    non-zero on exactly one rank), metric_calculator.py:68-117."""
    rank, world = get_dist_info()
    t = torch.as_tensor(values, dtype=torch.float64, device=device).clone()
    if world > 1:
        dist.reduce(t, dst=0)
    return t


def allreduce_mean_(grads, scale_fn=None):
    """DDP gradient exchange for one network: ONE flat fp32 bucket, all-reduce(SUM), mean
    written back in place.  With backend nccl this is a single RCCL ring/tree over xGMI
    (10.4 MB for G, 3.3 MB for D -- latency bound, so one large collective beats the
    reference's many DDP buckets).  `scale_fn(dst, src_flat_slice, a)` does dst = a * src
    on the device (HIP axpy); default = torch ops (CPU tensors in the gloo tests)."""
    rank, world = get_dist_info()
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    inv = 1.0 / world
    off = 0
    for g in grads:
        k = g.numel()
        src = flat[off:off + k]
        if scale_fn is not None:
            scale_fn(g, src, inv)
        else:
            g.copy_(src.view_as(g)).mul_(inv)
        off += k
