"""GPU: the data-parallel training step under 2 ranks (ADVICE high/medium).

Two ranks run VSRGANModel.train() with per-rank seeds and per-rank batch shards and must
(a) start from rank 0's weights (DDP's construction-time broadcast), (b) stay bit-identical
to each other after two iterations (flat-bucket gradient mean, SyncBatchNorm statistics,
fused adaptive-D decision), and (c) reproduce the single-process run on the concatenated
batch: running BatchNorm statistics to 1e-3 relative (+2e-5 abs), weights to a few Adam sign flips,
rank-0-reduced log values to 1e-3 relative.

`gloo` variant: both ranks share the ONE GPU of the test box (RCCL refuses two ranks on one
device); device tensors are staged through the host by utils/dist_utils, everything else is
the production path.  `nccl` variant: real RCCL over xGMI, skipped with < 2 GPUs."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, '_dist_train_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _run(world, backend, tmp_path, tag, extra_env=None):
    port = _free_port()
    outs = [str(tmp_path / f'{tag}_r{r}.pt') for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), port, outs[r], backend],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            logs.append(out.decode('utf-8', 'replace')[-3000:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f'rank {r} failed:\n{logs[r] if r < len(logs) else ""}'
    return [torch.load(o, map_location='cpu') for o in outs]


def _check_two_ranks(res2, res1):
    a, b = res2
    # (a) identical start = rank 0's seed-0 initialisation = the single-process model
    for k in a['after_init']:
        assert torch.equal(a['after_init'][k], b['after_init'][k]), ('init differs across ranks', k)
        assert torch.equal(a['after_init'][k], res1['after_init'][k]), ('init differs from 1-rank', k)
    # (b) replicas stay bit-identical
    for k in a['final']:
        assert torch.equal(a['final'][k], b['final'][k]), ('replicas diverged', k)
    # (c) equals the single-process run on the whole batch
    for k, v in res1['final'].items():
        w = a['final'][k]
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(w) == 6, k
        elif 'running_' in k:
            assert torch.allclose(v, w, rtol=1e-3, atol=2e-5), (k, (v - w).abs().max())
        elif v.dtype.is_floating_point:
            d = (v - w).abs().max().item()
            assert d <= 2.5e-4, (k, d)            # 2 Adam steps of 5e-5: a few sign flips at most
    for it in range(2):
        l1, l2 = res1['logs'][it]['local'], a['logs'][it]['reduced']
        assert a['logs'][it]['local']['distance'] == b['logs'][it]['local']['distance']
        assert a['logs'][it]['local']['n_upd_D'] == b['logs'][it]['local']['n_upd_D'] == l1['n_upd_D']
        for k, v in l1.items():
            tol = 1e-2 if (k in ('l_gan_G', 'p_fake_G') or it > 0) else 1e-3
            assert abs(l2[k] - v) <= tol * abs(v) + 5e-4, (it, k, l2[k], v)


def test_two_ranks_gloo_on_one_gpu_equal_single_process(tmp_path):
    res1 = _run(1, 'none', tmp_path, 'w1')[0]
    res2 = _run(2, 'gloo', tmp_path, 'w2')
    _check_two_ranks(res2, res1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs (RCCL over xGMI)')
@pytest.mark.parametrize('comm', ['torch', 'c_abi'])
def test_two_ranks_rccl_equal_single_process(tmp_path, comm):
    res1 = _run(1, 'none', tmp_path, 'w1')[0]
    res2 = _run(2, 'nccl', tmp_path, 'w2' + comm, {'TECOGAN_COMM': comm} if comm == 'c_abi' else None)
    _check_two_ranks(res2, res1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs (RCCL over xGMI)')
def test_two_ranks_rccl_with_chained_body_launches(tmp_path):
    """The first real DDP + chained-launch combination: at the REDS crop 128 (2 x 32 x 32 LR
    frames, the 16 x 16 x 4 row-chain form) every rank's SRNet body runs as persistent chained launches forward
    and backward while RCCL exchanges gradients / SyncBN statistics between the steps.  Replicas must stay
    bit-identical, no launch may have tripped its fail-safe, and the result must match the single-process run
    (chained as well) on the whole batch."""
    env = {'TG_TEST_CROP': '128'}
    res1 = _run(1, 'none', tmp_path, 'c1', env)[0]
    res2 = _run(2, 'nccl', tmp_path, 'c2', env)
    for r in [res1] + res2:
        assert r['chained_launches'] > 0 and not r['chain_disabled'], (r['chained_launches'], r['chain_disabled'])
    _check_two_ranks(res2, res1)


def test_worker_at_crop128_runs_the_chained_body_launches(tmp_path):
    """One-GPU half of the test above (the two-GPU half is skipped on a one-GPU box): the worker at the REDS crop
    really goes through the chained launches, and none trips its fail-safe."""
    r = _run(1, 'none', tmp_path, 'c1', {'TG_TEST_CROP': '128'})[0]
    assert r['chained_launches'] >= 2 * 2 * 7 and not r['chain_disabled'], (r['chained_launches'], r['chain_disabled'])


def test_c_abi_communicator_world1():
    """tg_comm_* / tg_allreduce_sum_f32 / tg_allgather_f32 through RCCL with one rank: binds
    librccl at run time, creates a communicator from a unique id, reduces in place."""
    import ctypes
    from tecogan_pytorch_amd import _lib as L
    lib = L.lib()
    ident = (ctypes.c_uint8 * 128)()
    L.check(lib.tg_comm_get_unique_id(ident), 'tg_comm_get_unique_id')
    assert lib.tg_comm_library_origin().decode() != ''
    comm = ctypes.c_void_p()
    L.check(lib.tg_comm_init_rank(ident, 1, 0, ctypes.byref(comm)), 'tg_comm_init_rank')
    assert lib.tg_comm_world(comm) == 1 and lib.tg_comm_rank(comm) == 0
    x = torch.arange(1 << 20, dtype=torch.float32, device='cuda') * 0.5
    ref = x.clone()
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.tg_allreduce_sum_f32(comm, x.data_ptr(), x.numel(), st), 'tg_allreduce_sum_f32')
    out = torch.empty_like(x)
    L.check(lib.tg_allgather_f32(comm, x.data_ptr(), out.data_ptr(), x.numel(), st), 'tg_allgather_f32')
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and torch.equal(out, ref)
    with pytest.raises(L.TecoganHipError):
        L.check(lib.tg_comm_init_rank(ident, 2, 5, ctypes.byref(ctypes.c_void_p())), 'bad rank')
    L.check(lib.tg_comm_destroy(comm), 'tg_comm_destroy')


def test_sync_bn_statistics_survive_a_large_mean_offset():
    """|mean| >> std.  The SyncBN path (Chan merge of centred per-rank
    sums) must equal the two-pass fused kernel, not lose the variance to cancellation."""
    import tecogan_pytorch_amd.ops as ops
    g = torch.Generator().manual_seed(3)
    x = (1000.0 + 0.05 * torch.randn(4, 8, 32, 32, generator=g)).cuda()
    gamma, beta = torch.ones(8, device='cuda'), torch.zeros(8, device='cuda')
    rm1, rv1 = torch.zeros(8, device='cuda'), torch.ones(8, device='cuda')
    rm2, rv2 = rm1.clone(), rv1.clone()
    y1, m1, i1 = ops.bn_lrelu_train_fwd(x, gamma, beta, rm1, rv1)
    y2, m2, i2, cnt = ops.sync_bn_lrelu_train_fwd(x, gamma, beta, rm2, rv2)
    assert cnt == 4 * 32 * 32
    assert torch.equal(m1, m2) and torch.equal(i1, i2) and torch.equal(y1, y2)
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    var = x.double().var(dim=(0, 2, 3), unbiased=False).float()
    assert torch.allclose(1.0 / i2 ** 2 - 1e-5, var.cuda(), rtol=2e-3)
    # two simulated ranks: gathered statistics of the two halves merge to the global ones
    from tecogan_pytorch_amd import _lib as L
    lib, st = L.lib(), torch.cuda.current_stream().cuda_stream
    halves = torch.empty(2, 16, device='cuda')
    for r in range(2):
        xr = x[2 * r:2 * r + 2].contiguous()
        L.check(lib.tg_bn_local_stats(xr.data_ptr(), halves[r].data_ptr(), 2, 8, 1024, st), 'local')
    mean, invstd = torch.empty(8, device='cuda'), torch.empty(8, device='cuda')
    L.check(lib.tg_bn_merge_stats(halves.data_ptr(), 2, 2.0 * 1024, 1e-5, 0.1, mean.data_ptr(),
                                  invstd.data_ptr(), None, None, 8, st), 'merge')
    assert torch.allclose(mean, m1, rtol=0, atol=1e-4)
    assert torch.allclose(invstd, i1, rtol=1e-3)


ROOT = os.path.dirname(HERE)


def _bench_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(kw)
    return env


def test_plain_bench_invocation_brings_up_two_ranks():
    """plain `python bench.py --gpus 2 ...` (no launcher around it) spawns its own two ranks
    and the line says n_gpus 2.  On a one-GPU box this is the rehearsal form (both ranks on cuda:0 over gloo); with
    two GPUs it is the real thing over RCCL.  The gradient buckets on the line are the sizes DESIGN.md quotes."""
    import json
    env = _bench_env() if torch.cuda.device_count() >= 2 else _bench_env(TG_BENCH_REHEARSAL='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
                        '--clips', '2', '--train-steps', '2', '--cpu-frames', '0', '--no-roofline', '--no-secondary',
                        '--no-parity-check'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    js = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith('{')]
    assert len(js) == 1, r.stdout[-2000:]
    line = json.loads(js[0])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['launched_by'].startswith('bench.py')
    assert line['value'] > 0 and line['steps'] == 4
    t = line['train_ddp']
    assert 'error' not in t, t
    assert t['n_gpus'] == 2 and t['process_group']['world_seen_by_rank0'] == 2
    assert t['allreduce_G']['bytes'] == 10357504 and t['allreduce_D']['bytes'] == 3278336
    assert t['allreduce_G']['ms_per_call'] > 0 and t['allreduce_D']['ms_per_call'] > 0


@pytest.mark.skipif(torch.cuda.device_count() >= 8, reason='the box really has 8 GPUs')
def test_plain_bench_invocation_refuses_more_gpus_than_present():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1'],
                       env=_bench_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith('{')]
    assert '--gpus 8 requested but' in r.stderr
