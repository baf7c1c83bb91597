"""bench.py's own rank launcher: plain `python bench.py --gpus N` must bring up N ranks by
itself, relay exactly one JSON line whose n_gpus is the number of ranks the process group joined, and fail loudly
-- never print a 1-GPU line -- when the box cannot give it N GPUs or a rank dies.

No GPU here: TG_BENCH_LAUNCH_PROBE=1 makes every child join a gloo group on the CPU and stop after the
launcher-level fields (the GPU-side rehearsal of the full line is tests/test_dist_gpu.py).
Reference counterpart: train.sh:42-53 (`torch.distributed.launch --nproc_per_node`), codes/utils/dist_utils.py:8-24.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'TG_BENCH_REHEARSAL')}
    env.update(kw)
    return env


def _json_lines(out):
    return [ln for ln in out.splitlines() if ln.lstrip().startswith('{')]


@pytest.mark.parametrize('world', [2, 4])
def test_plain_invocation_spawns_n_ranks(world):
    r = subprocess.run([sys.executable, BENCH, '--gpus', str(world), '--steps', '3', '--warmup', '1'],
                       env=_env(TG_BENCH_LAUNCH_PROBE='1'), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    js = _json_lines(r.stdout)
    assert len(js) == 1, r.stdout
    line = json.loads(js[0])
    assert line['n_gpus'] == world and line['ranks_seen'] == world and line['distinct_devices'] == world
    assert line['launched_by'] == 'bench.py' and line['steps'] == 3 and line['warmup'] == 1


def test_a_dead_rank_fails_the_launch_and_prints_no_line():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2'], env=_env(TG_BENCH_LAUNCH_PROBE='1', TG_BENCH_PROBE_FAIL_RANK='1'),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert 'rank 1 exited with code 7' in r.stderr


def test_more_gpus_than_the_box_has_is_refused():
    """This container has no GPU: `--gpus 8` must exit non-zero with a clear message and no JSON line."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--steps', '2', '--warmup', '1'], env=_env(),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert '--gpus 8 requested but' in r.stderr


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '4'], env=_env(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0',
                                                                       MASTER_ADDR='127.0.0.1', MASTER_PORT='29999'),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and _json_lines(r.stdout) == []
    assert '--gpus 4 but WORLD_SIZE 2' in r.stderr


def test_torch_distributed_run_form_still_works():
    """The driver's documented N > 1 form: RANK / WORLD_SIZE come from torch.distributed.run, bench.py must NOT spawn again."""
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29541', BENCH, '--gpus', '2'],
                       env=_env(TG_BENCH_LAUNCH_PROBE='1'), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    js = _json_lines(r.stdout)
    assert len(js) == 1
    line = json.loads(js[0])
    assert line['n_gpus'] == 2 and line['launched_by'] == 'torch.distributed.run'
