"""world_size-2 gloo test of the N > 1 path: round-robin clip sharding with no
data-path collective, max-over-ranks timing and metric reduce to rank 0."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_seq, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tecogan_pytorch_amd.utils import dist_utils as D
    opt = {}
    D.init_dist(opt, rank, backend='gloo')
    assert opt['rank'] == rank and opt['world_size'] == world and opt['dist']
    mine = D.shard_indices(num_seq)
    # each rank "processes" its own sequences: metric = 10 + idx, others stay 0
    vals = [0.0] * num_seq
    for i in mine:
        vals[i] = 10.0 + i
    red = D.reduce_sum_to_master(vals)
    tmax = D.max_over_ranks(1.0 + rank)
    # gradient bucket: rank r holds (r+1) * base; the mean over 2 ranks is 1.5 * base
    base = [torch.arange(6, dtype=torch.float32).view(2, 3), torch.ones(5), torch.full((1,), 4.0)]
    grads = [(rank + 1) * b.clone() for b in base]
    D.allreduce_mean_(grads)
    gsum = [g.tolist() for g in grads]
    calls = []
    D.master_only(lambda: calls.append(rank))()
    # start-up agreement: identical vectors pass, a differing rank makes EVERY rank raise
    agree = D.assert_ranks_agree([101, 4, 1, 0, 2589440, 819648], 'test vector') == [101, 4, 1, 0, 2589440, 819648]
    try:
        D.assert_ranks_agree([101, 4 if rank == 0 else 0, 1], 'chain capability')
        mismatch = 'not raised'
    except RuntimeError as e:
        mismatch = str(e)
    out[rank] = (mine, red.tolist(), tmax, calls, gsum, agree, mismatch)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_and_reductions():
    world, num_seq = 2, 7
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, num_seq, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert sorted(a[0] + b[0]) == list(range(num_seq))          # every clip exactly once
    assert set(a[0]).isdisjoint(b[0])
    assert a[0] == [0, 2, 4, 6] and b[0] == [1, 3, 5]           # main.py:169 order
    assert a[1] == [10.0 + i for i in range(num_seq)]           # rank 0 holds the full table
    assert a[2] == 2.0 and b[2] == 2.0                          # MAX over ranks everywhere
    assert a[3] == [0] and b[3] == []                           # master_only
    expect = [[[0.0, 1.5, 3.0], [4.5, 6.0, 7.5]], [1.5] * 5, [6.0]]
    assert a[4] == expect and b[4] == expect                    # flat-bucket mean, shapes kept
    assert a[5] and b[5]
    for m in (a[6], b[6]):                                      # both ranks refuse, naming the differing rank
        assert 'ranks disagree on chain capability' in m and '{1: [101, 0, 1]}' in m, m


def test_gradient_bucket_sizes_are_the_documented_ones():
    """The flat gradient buckets the data-parallel step all-reduces (10.36 MB G,
    3.28 MB D): the Adam layout's arithmetic (64-float alignment per tensor + one block holding the fault slot) on the
    shipped TecoGAN networks -- what `bench.py --gpus N` prints as allreduce_G / allreduce_D bytes."""
    from tecogan_pytorch_amd.models.networks import FRNet, SpatioTemporalDiscriminator
    from tecogan_pytorch_amd.models.optim import Adam

    def bucket_bytes(net):
        tot = sum((p.numel() + Adam.ALIGN - 1) // Adam.ALIGN * Adam.ALIGN for p in net.parameters())
        return 4 * (tot + Adam.ALIGN)
    g = FRNet(3, 3, 64, 10, 'BD', 4)
    d = SpatioTemporalDiscriminator(3, 128, 3, 'BD', 4)
    assert sum(p.numel() for p in g.parameters()) == 1745506 + 843587       # FNet + SRNet
    assert bucket_bytes(g) == 10357504, bucket_bytes(g)
    assert bucket_bytes(d) == 3278336, bucket_bytes(d)


def test_single_process_defaults():
    from tecogan_pytorch_amd.utils import dist_utils as D
    assert D.get_dist_info() == (0, 1)
    assert D.shard_indices(5) == [0, 1, 2, 3, 4]
    assert D.max_over_ranks(3.5) == 3.5
