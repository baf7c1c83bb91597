#!/usr/bin/env python
"""Benchmark of the MI355X-native TecoGAN/FRVSR recurrent frame (FRNet.step).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1
either the driver launches it under torch.distributed.run (RANK / WORLD_SIZE /
MASTER_* in the environment), or -- plain `python bench.py --gpus N` with no
RANK set -- this file brings up its own N ranks (launch_ranks(): one child per
GPU, the reference's `torch.distributed.launch --nproc_per_node N` of
train.sh:42-53 / codes/utils/dist_utils.py:8-24).  Rank 0 prints ONE JSON line
whose `n_gpus` is the number of ranks the process group actually joined
(`ranks_seen`: a sum all-reduce of ones), asserted equal to --gpus; a --gpus N
request on a box with fewer than N GPUs exits non-zero and prints no line.

Workload = BASELINE.json configs[1]: TecoGAN 4xSR BD generator-only inference,
synthetic 3x134x320 LR clip (the shape the reference's published 27 FPS is
quoted on, profile.sh / main.py:210-264).  A "step" is one recurrent frame
(FNet -> pad/upsample/warp/space_to_depth -> SRNet -> uint8 quantise) of a
K-frame clip per GPU run through FRNet.infer_sequence (true recurrence; FNet of
frame t+1 overlapped with SRNet of frame t on a second HIP stream), uniform
random LR frames resident in HBM, random-init weights (seeded), fp32 end to end.
The reference's own profile protocol (independent random frames through step(),
sync per frame) is reported beside it as fps_step_protocol_*.
Clips are independent, so N GPUs run N clips with no data-path collective
(weak scaling); value = N*K frames / max-over-ranks wall time.

Extra objects on the same line:
  roofline      dominant kernel class (fp32-MFMA conv3x3): algorithmic FLOPs of
                its launches in one frame / their summed duration measured with
                HIP events around a masked replay of exactly those launches.
  cpu_baseline  the CPU oracle (torch-CPU restatement of the reference path,
                bit-checked against the reference in the authoring container)
                timed on this host's cores on a bounded sample (N == 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--clips', type=int, default=10,
                    help='how many times the K-step timed region is repeated (median reported)')
    ap.add_argument('--no-train-leg', action='store_true',
                    help='skip the data-parallel TecoGAN training leg (BASELINE configs[2]/[3])')
    ap.add_argument('--train-steps', type=int, default=10)
    ap.add_argument('--lr-size', default='3x134x320')
    ap.add_argument('--scale', type=int, default=4)
    ap.add_argument('--degradation', default='BD')
    ap.add_argument('--cpu-frames', type=int, default=16,
                    help='max frames of the CPU baseline sample (0 disables)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-live-pmc', action='store_true',
                    help='do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='single-stream clip inference only (for kernel-trace profiles whose per-kernel '
                         'durations are not inflated by the FNet/SRNet stream overlap)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the secondary protocols (step protocol, 4-clip batch): counter passes then '
                         'see only the launches of the headline workload')
    ap.add_argument('--no-parity-check', action='store_true',
                    help='skip the untimed parity check (profiling passes: only the launches of the timed workload)')
    ap.add_argument('--aten-frames', type=int, default=30,
                    help='frames of the ATen/MIOpen-on-GPU context baseline (0 disables)')
    return ap.parse_args()


def _die(msg, code=2):
    print(f'bench.py: {msg}', file=sys.stderr, flush=True)
    sys.exit(code)


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) with no RANK in the environment: start N children of this
    very command line, one rank per GPU, on a free local port; relay rank 0's stdout (the one JSON
    line), pass every rank's stderr through, and return non-zero as soon as ANY child fails (the
    others are then terminated: a rank that lost its peer would sit in a collective for ever).
    The reference's counterpart is `python -m torch.distributed.launch --nproc_per_node N`
    (train.sh:42-53) + init_dist (codes/utils/dist_utils.py:8-24)."""
    import socket
    import subprocess
    n = args.gpus
    probe = os.environ.get('TG_BENCH_LAUNCH_PROBE') == '1'
    rehearsal = os.environ.get('TG_BENCH_REHEARSAL') == '1'
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not probe and not rehearsal and have < n:
        _die(f'--gpus {n} requested but {have} GPU(s) visible on this box: refusing to print a line for fewer ranks')
    if rehearsal and have < 1:
        _die('TG_BENCH_REHEARSAL=1 needs one GPU')
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(n), 'LOCAL_WORLD_SIZE': str(n),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': '0',
                    'TG_BENCH_LAUNCHED_BY': 'bench.py'})
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading
    lines = []
    th = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    th.start()
    rc, live = 0, set(range(n))
    deadline = time.time() + float(os.environ.get('TG_BENCH_LAUNCH_TIMEOUT', '3600'))     # a hung rank must not hang the caller for ever
    while live and rc == 0:
        if time.time() > deadline:
            rc = 4
            print(f'bench.py: ranks {sorted(live)} still running after TG_BENCH_LAUNCH_TIMEOUT; stopping them', file=sys.stderr, flush=True)
            break
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0:
                    rc = c if c > 0 else 1
                    print(f'bench.py: rank {r} exited with code {c}; stopping the other ranks', file=sys.stderr, flush=True)
        time.sleep(0.05)
    for r in live:                         # only the PIDs started here
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    th.join(timeout=10)
    for ln in lines:                       # whatever else rank 0 wrote to stdout stays visible, on stderr
        if not ln.lstrip().startswith('{'):
            sys.stderr.write(ln)
    if rc == 0:
        js = [ln for ln in lines if ln.lstrip().startswith('{')]
        if len(js) != 1:
            _die(f'rank 0 printed {len(js)} JSON lines, expected exactly one', 3)
        seen = json.loads(js[0]).get('n_gpus')
        if seen != n:
            _die(f'the line says n_gpus {seen}, --gpus was {n}', 3)
        sys.stdout.write(js[0] if js[0].endswith('\n') else js[0] + '\n')
        sys.stdout.flush()
    return rc


def ranks_seen_by_collective(dist, dev):
    """How many ranks the process group REALLY holds: every rank contributes 1 to a sum all-reduce
    (RCCL under backend nccl), plus the set of distinct (host, GPU) pairs from an all-gather."""
    one = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(one)
    ident = [None] * dist.get_world_size()
    me = (os.uname().nodename,
          str(getattr(torch.cuda.get_device_properties(dev), 'uuid', None) or torch.cuda.current_device())
          if dev.type == 'cuda' else f'cpu{dist.get_rank()}')
    dist.all_gather_object(ident, me)
    return int(round(one.item())), len(set(ident))


def launch_probe(args):
    """TG_BENCH_LAUNCH_PROBE=1: the launcher's plumbing on a box without a GPU (tests/test_bench_launch_cpu.py):
    each child joins a gloo group on the CPU and rank 0 prints the launcher-level fields only."""
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if os.environ.get('TG_BENCH_PROBE_FAIL_RANK') == str(rank):
        _die(f'rank {rank}: injected failure', 7)
    dist.init_process_group(backend='gloo')
    seen, distinct = ranks_seen_by_collective(dist, torch.device('cpu'))
    assert seen == dist.get_world_size() == world == args.gpus, (seen, world, args.gpus)
    if rank == 0:
        print(json.dumps({'probe': True, 'n_gpus': seen, 'ranks_seen': seen, 'distinct_devices': distinct,
                          'steps': args.steps, 'warmup': args.warmup,
                          'launched_by': os.environ.get('TG_BENCH_LAUNCHED_BY', 'torch.distributed.run')}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def kernel_table(net, plan, bufs, reps=20):
    """Per kernel class: launches/frame, algorithmic flops & bytes, measured ms
    per frame (HIP events around a masked replay on the launch stream)."""
    from tecogan_pytorch_amd import _lib as L
    lib = L.lib()
    lr_c, lr_p, hr_p, out = bufs
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    nk = lib.tg_frnet_plan_kinds()
    for k in range(nk):
        nl, fl, by = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        L.check(lib.tg_frnet_plan_kind_stats(plan.handle, k, ctypes.byref(nl), ctypes.byref(fl),
                                             ctypes.byref(by)), 'kind_stats')
        name = lib.tg_frnet_kind_name(k).decode()
        if nl.value == 0 or name.startswith('quantize'):
            continue
        mask = 1 << k

        def run(r):
            L.check(lib.tg_frnet_replay(plan.handle, lr_c.data_ptr(), lr_p.data_ptr(),
                                        hr_p.data_ptr(), out.data_ptr(), mask, r, stream),
                    'tg_frnet_replay')
        # one full step first: earlier class replays ran on stale inputs and left e.g. a
        # garbage flow field behind, and the gather kernels are data dependent
        L.check(lib.tg_frnet_replay(plan.handle, lr_c.data_ptr(), lr_p.data_ptr(), hr_p.data_ptr(),
                                    out.data_ptr(), (1 << nk) - 1, 1, stream), 'tg_frnet_replay')
        run(3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run(reps)                 # all replays enqueued by ONE C call: no host gap per replay
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rows.append(dict(kernel=name, launches=nl.value, gflop=fl.value / 1e9,
                         mbytes=by.value / 1e6, ms_per_frame=ms,
                         tflops=(fl.value / 1e12) / (ms / 1e3) if fl.value else None,
                         gbs=(by.value / 1e9) / (ms / 1e3)))
    return rows


def _csrc_hashes():
    """sha1 of every kernel source: tools/summarize_pmc.py stamps them into profiles/pmc_traffic.json
    ("_sources") so that a traffic figure measured on OTHER kernel code is reported as stale."""
    import hashlib
    d = os.path.join(ROOT, 'tecogan-pytorch_amd', 'csrc')
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            out[f] = hashlib.sha1(open(os.path.join(d, f), 'rb').read()).hexdigest()[:16]
    return out


def pmc_traffic_stale():
    """True when the committed counter passes were collected on different kernel sources (or carry no stamp)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            stamp = json.load(f).get('_sources')
    except Exception:
        return True
    return stamp != _csrc_hashes()


def pmc_traffic(kernel, tag=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, written by tools/summarize_pmc.py); None if absent.  The launches of
    the other workloads of the counter run are kept under tagged keys ("... [2xBI]", "... [train128]")."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            table = json.load(f)
        want = kernel.replace(' ', '').rstrip('>')
        for k, v in table.items():
            if k.startswith('_'):
                continue
            name, _, ktag = k.partition(' [')
            if (ktag.rstrip(']') or None) != tag:
                continue
            if name.replace(' ', '').replace('tg::', '').startswith(want):
                return v
    except Exception:
        pass
    return None


def live_pmc_traffic(lr_size, scale, deg, timeout_s=150):
    """HBM bytes per launch of every tg:: kernel of the frame, MEASURED IN THIS RUN: two child invocations of this script
    under `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE in passes of their own, with --kernel-trace only, as
    MI355X_MICROARCH.md (section HBM / rocprofv3 PMC) prescribes -- on a 1-clip single-stream run of the same workload.
    Returns {kernel symbol without arguments: bytes per launch} or None (no rocprofv3, a pass failed or timed out: the
    line then falls back to the committed passes).  Values are the raw counters (KB -> B), the same convention as
    profiles/pmc_traffic.json: FETCH_SIZE under-counts 16-byte-per-lane streaming reads by 2x on gfx950 and WRITE_SIZE
    is uncalibrated (the guide), so ratios between rounds are exact and absolutes are a lower bound."""
    import csv
    import shutil
    import signal
    import subprocess
    import tempfile
    rp = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rp):
        return None
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', '6', '--warmup', '2', '--clips', '1', '--lr-size', lr_size,
           '--scale', str(scale), '--degradation', deg, '--no-roofline', '--no-pipeline', '--no-secondary',
           '--no-parity-check', '--no-train-leg', '--cpu-frames', '0', '--aten-frames', '0', '--no-live-pmc']
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    per = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='tg_pmc_', dir='/tmp')
        try:
            proc = subprocess.Popen([rp, '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--'] + cmd,
                                    cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)        # exactly the group this call started
                proc.wait()
                return None
            path = None
            for root_, _, files in os.walk(d):
                if 'pmc_counter_collection.csv' in files:
                    path = os.path.join(root_, 'pmc_counter_collection.csv')
            if rc != 0 or path is None:
                return None
            acc = {}
            with open(path) as f:
                for r in csv.DictReader(f):
                    if 'tg::' not in r['Kernel_Name'] or r['Counter_Name'] != ctr:
                        continue
                    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tg::', '').replace(' ', '')
                    a_ = acc.setdefault(name, [0.0, 0])
                    a_[0] += float(r['Counter_Value']); a_[1] += 1
            for name, (tot, n) in acc.items():
                per.setdefault(name, {})[ctr] = tot / n * 1024.0
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {k: v['FETCH_SIZE'] + v['WRITE_SIZE'] for k, v in per.items() if len(v) == 2} or None


def _live_lookup(table, kernel):
    if not table:
        return None
    want = kernel.replace(' ', '').rstrip('>')
    for k, v in table.items():
        if k.startswith(want):
            return v
    return None


def cpu_baseline(sd, scale, deg, c, h, w, max_frames, max_seconds):
    """Reference-protocol FPS of the CPU oracle: fresh rand inputs per frame
    (generated outside the timer), eval/no_grad, FPS = frames / sum(step time)
    (codes/main.py:249-262 minus the CUDA sync).  torch's default of one thread per logical
    core oversubscribes a 42 880-pixel convolution on a 128/256-thread host (round 1: 0.93
    frames/s on 128 threads, slower than 8 cores), so the thread count is swept first
    (2 frames each) and the best one is used for the sample."""
    from oracle import tecogan_oracle as O
    torch.manual_seed(1)
    default_threads = torch.get_num_threads()

    def frame():
        a = [torch.rand(1, c, h, w), torch.rand(1, c, h, w), torch.rand(1, c, scale * h, scale * w)]
        t0 = time.perf_counter()
        O.frnet_step(sd, a[0], a[1], a[2], scale, deg)
        return time.perf_counter() - t0
    sweep = {}
    with torch.no_grad():
        cands = sorted({t for t in (8, 16, 32, 64, default_threads) if t <= (os.cpu_count() or 8)})
        for nt in cands:
            torch.set_num_threads(nt)
            frame()                                   # warm-up (thread pool / oneDNN primitives)
            sweep[nt] = 2.0 / (frame() + frame())
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        frame()
        tot, frames = 0.0, 0
        while frames < max_frames and tot < max_seconds:
            tot += frame()
            frames += 1
    torch.set_num_threads(default_threads)
    return dict(value=frames / tot, unit='frames/s', cores=best, kind='port',
                sample=f'{frames} frames of the same {c}x{h}x{w} workload, oracle/tecogan_oracle.py '
                       f'(torch-CPU fp32, oneDNN), nproc={os.cpu_count()}, best of a thread sweep',
                thread_sweep_fps={str(k): round(v, 3) for k, v in sweep.items()})


def aten_gpu_baseline(sd, scale, deg, c, h, w, frames, dev):
    """Context only: the same frame through stock PyTorch-ROCm ops (ATen / MIOpen fp32 on this
    GPU) -- i.e. what the unmodified reference's op path does on MI355X -- using the oracle's
    restatement on device tensors, reference protocol (sync every frame, main.py:249-262)."""
    from oracle import tecogan_oracle as O
    sd = {k: v.to(dev) for k, v in sd.items()}
    tot = 0.0
    try:
        with torch.no_grad():
            def run():
                a = [torch.rand(1, c, h, w, device=dev), torch.rand(1, c, h, w, device=dev),
                     torch.rand(1, c, scale * h, scale * w, device=dev)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _aten_step(O, sd, a[0], a[1], a[2], scale, deg)
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            for _ in range(15):        # MIOpen's find runs inside the first calls of every conv shape
                run()                  # (a fresh box measured 10 frames/s with 5 warm-up frames, 43-49 warm)
            for _ in range(frames):
                tot += run()
    except Exception as e:       # context number only: never fail the bench for it
        return {'error': repr(e)[:200]}
    return {'value': frames / tot, 'unit': 'frames/s', 'kind': 'ATen/MIOpen fp32 on the same GPU',
            'sample': f'{frames} frames, sync every frame'}


def _aten_step(O, sd, lr_curr, lr_prev, hr_prev, scale, deg):
    """FRNet.step with torch.nn.functional ops exactly as the reference composes them."""
    import torch.nn.functional as F
    fs = O._sub(sd, 'fnet.')
    out = torch.cat([lr_curr, lr_prev], 1)
    for enc in ('encoder1', 'encoder2', 'encoder3'):
        out = F.leaky_relu(F.conv2d(out, fs[enc + '.0.weight'], fs[enc + '.0.bias'], padding=1), 0.2)
        out = F.leaky_relu(F.conv2d(out, fs[enc + '.2.weight'], fs[enc + '.2.bias'], padding=1), 0.2)
        out = F.max_pool2d(out, 2, 2)
    for dec in ('decoder1', 'decoder2', 'decoder3'):
        out = F.leaky_relu(F.conv2d(out, fs[dec + '.0.weight'], fs[dec + '.0.bias'], padding=1), 0.2)
        out = F.leaky_relu(F.conv2d(out, fs[dec + '.2.weight'], fs[dec + '.2.bias'], padding=1), 0.2)
        out = F.interpolate(out, scale_factor=2, mode='bilinear', align_corners=False)
    out = F.leaky_relu(F.conv2d(out, fs['flow.0.weight'], fs['flow.0.bias'], padding=1), 0.2)
    flow = torch.tanh(F.conv2d(out, fs['flow.2.weight'], fs['flow.2.bias'], padding=1)) * 24
    h, w = lr_curr.shape[2:]
    flow = F.pad(flow, (0, w - w // 8 * 8, 0, h - h // 8 * 8), 'reflect')

    def up(x):
        if deg == 'BI':
            return F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False)
        k = sd['upsample_func.kernels']
        n, c, hh, ww = x.shape
        x = F.pad(x.reshape(n * c, 1, hh, ww), (1, 2, 1, 2), mode='replicate')
        o = F.conv2d(x, k.view(scale, 1, 4, 1)).permute(0, 2, 1, 3).reshape(n * c, 1, scale * hh, ww + 3)
        o = F.conv2d(o, k.view(scale, 1, 1, 4)).permute(0, 2, 3, 1).reshape(n, c, scale * hh, scale * ww)
        return o
    hr_flow = scale * up(flow)
    n, c, H, W = hr_prev.shape
    iu = torch.linspace(-1.0, 1.0, W).view(1, 1, 1, W).expand(n, -1, H, -1)
    iv = torch.linspace(-1.0, 1.0, H).view(1, 1, H, 1).expand(n, -1, -1, W)
    grid = torch.cat([iu, iv], 1).to(hr_flow.device)          # host mesh + H2D, as net_utils.py:62-64
    grid = (grid + torch.cat([hr_flow[:, 0:1] / ((W - 1.0) / 2.0),
                              hr_flow[:, 1:2] / ((H - 1.0) / 2.0)], 1)).permute(0, 2, 3, 1)
    warped = F.grid_sample(hr_prev, grid, mode='bilinear', padding_mode='border', align_corners=True)
    s2d = warped.reshape(n, c, H // scale, scale, W // scale, scale).permute(0, 3, 5, 1, 2, 4) \
        .reshape(n, scale * scale * c, H // scale, W // scale)
    ss = O._sub(sd, 'srnet.')
    out = F.relu(F.conv2d(torch.cat([lr_curr, s2d], 1), ss['conv_in.0.weight'], ss['conv_in.0.bias'],
                          padding=1))
    b = 0
    while f'resblocks.{b}.conv.0.weight' in ss:
        t = F.relu(F.conv2d(out, ss[f'resblocks.{b}.conv.0.weight'], ss[f'resblocks.{b}.conv.0.bias'],
                            padding=1))
        out = F.conv2d(t, ss[f'resblocks.{b}.conv.2.weight'], ss[f'resblocks.{b}.conv.2.bias'],
                       padding=1) + out
        b += 1
    for u in ([0, 2] if scale == 4 else [0]):
        out = F.relu(F.conv_transpose2d(out, ss[f'conv_up.{u}.weight'], ss[f'conv_up.{u}.bias'],
                                        stride=2, padding=1, output_padding=1))
    out = F.conv2d(out, ss['conv_out.weight'], ss['conv_out.bias'], padding=1)
    return out + up(lr_curr)


def copy_ceiling(dev, nbytes, blocks, threads, rotate_bytes=0, reps=48):
    """A float4 copy of `nbytes` (read once + written once = 2 nbytes of traffic) with the warp launch's own grid
    (tg_copy_ceiling): what a plain streaming kernel of that size reaches on this GPU.  rotate_bytes > 0: the launches
    rotate over enough buffer pairs that none finds its input in L2 / Infinity Cache (the batched warp line's protocol);
    0: one buffer pair, re-used back to back (the one-frame line's protocol: the previous frame was written a moment
    ago and is read from the caches)."""
    from tecogan_pytorch_amd import _lib as L
    from tecogan_pytorch_amd import ops
    nbytes = int(nbytes) // 16 * 16
    npair = max(1, int(rotate_bytes // (2 * nbytes)) + 1) if rotate_bytes else 1
    bufs = [(torch.rand(nbytes // 4, device=dev), torch.empty(nbytes // 4, device=dev)) for _ in range(npair)]
    lib = L.lib()

    def go(i):
        a, b = bufs[i % npair]
        L.check(lib.tg_copy_ceiling(a.data_ptr(), b.data_ptr(), nbytes, blocks, threads, ops._stream()), 'tg_copy_ceiling')
    for i in range(npair + 2):
        go(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        go(i)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    return {'GBps': 2 * nbytes / us * 1e-3, 'avg_launch_us': us, 'bytes_read_plus_written': 2 * nbytes,
            'grid': [blocks, threads], 'buffer_pairs': npair,
            'what': 'float4 grid-stride copy, same grid as the warp launch, back-to-back launches between HIP events '
                    '(so a launch boundary is inside each sample, as in the batched warp line)'}


def warp_batched_roofline(dev, h, w, scale, deg, clips=8, reps=48):
    """The fused flow-upsample + warp + space_to_depth kernel on `clips` independent clips
    in one launch (the serving configuration: one frame step of `clips` streams).  At n=1
    the kernel moves one frame and is launch/ramp-bound; this line shows what the same
    kernel reaches when the launch is large enough to cover the ramp.  The flow is a camera
    motion -- a pan of a few HR pixels plus 1 % zoom and a small roll, different per clip --
    i.e. smooth like real optical flow, not constant.  Launches rotate over enough buffer
    sets (> 512 MB) that none finds its inputs in L2 / Infinity Cache."""
    from tecogan_pytorch_amd import ops
    mode = ops.UP_MODE[deg]
    per_set = clips * (2 * 3 * scale * scale * h * w + 2 * h * w) * 4
    nsets = max(2, int(600e6 // per_set) + 1)
    g = torch.Generator(device='cpu').manual_seed(7)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) - h / 2,
                            torch.arange(w, dtype=torch.float32) - w / 2, indexing='ij')
    sets = []
    for _ in range(nsets):
        pan = (torch.rand(clips, 2, 1, 1, generator=g) - 0.5) * 2.0       # LR px: +-1 (= +-4 HR px)
        zoom = 0.01 * (torch.rand(clips, 1, 1, 1, generator=g) - 0.5) * 2
        roll = 0.005 * (torch.rand(clips, 1, 1, 1, generator=g) - 0.5) * 2
        fx = pan[:, 0:1] + zoom * xs - roll * ys
        fy = pan[:, 1:2] + zoom * ys + roll * xs
        flow = torch.cat([fx, fy], 1).to(dev).contiguous()
        prev = torch.rand(clips, 3, scale * h, scale * w, generator=g).to(dev).contiguous()
        sets.append((flow, prev, torch.empty(clips, scale * scale * 3, h, w, device=dev)))
    for fl, pv, out in sets:
        ops.flowup_warp_s2d(fl, pv, h, w, scale, mode, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        fl, pv, out = sets[i % nsets]
        ops.flowup_warp_s2d(fl, pv, h, w, scale, mode, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    # algorithmic bytes: read HR frame once + LR flow once, write the s2d tensor once
    mbytes = per_set / 1e6
    gbs = mbytes / us * 1e3
    nseg = -(-scale * w // 256)
    tiles = nseg * h * clips                       # the launcher's grid: one block per (LR row, 256-column segment, clip)
    cc = copy_ceiling(dev, per_set // 2, tiles, 256 if tiles > 2048 and scale == 4 else (512 if scale == 4 else 256),
                      rotate_bytes=600e6, reps=reps)
    return {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': gbs / HBM_PEAK_GBS, 'copy_ceiling': cc, 'frac_of_copy_ceiling': gbs / cc['GBps'],
            'traffic': None, 'clips_per_launch': clips,
            'avg_launch_us': us, 'algorithmic_mbytes_per_launch': mbytes,
            'flow': 'camera motion: pan +-4 HR px, zoom +-1 %, roll +-0.005 rad, per clip',
            'note': 'back-to-back launches on one stream over %d rotating buffer sets, HIP '
                    'events; same kernel as roofline_warp' % nsets}


def config5_leg(dev, gen, frames):
    """BASELINE configs[4]: TecoGAN 2x BI generator-only inference of a 3x268x640 LR clip (same HR
    size as the headline): clip rate (median of 3 clips) and the dominant conv class -- SRNet's 21
    full-resolution layers as ONE chained Winograd launch -- against the fp32-MFMA peak and against
    the Winograd form's own ceiling."""
    from tecogan_pytorch_amd.models.networks import FRNet
    from tecogan_pytorch_amd import _lib as L
    c, h, w, s = 3, 268, 640, 2
    torch.manual_seed(0)
    net = FRNet(c, c, 64, 10, 'BI', s).to(dev).eval()
    clip = torch.rand(frames, c, h, w, generator=gen).to(dev)
    with torch.no_grad():
        for _ in range(2):
            net.infer_sequence(clip, dev, return_device_tensor=True)
            torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            net.infer_sequence(clip, dev, return_device_tensor=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        net.check_faults()
        plan = net._get_plan(1, h, w, dev)
        bufs = (torch.rand(1, c, h, w, generator=gen).to(dev), torch.rand(1, c, h, w, generator=gen).to(dev),
                torch.rand(1, c, s * h, s * w, generator=gen).to(dev), torch.empty(1, c, s * h, s * w, device=dev))
        rows = kernel_table(net, plan, bufs, reps=10)
        net.check_faults()
    gf, _ = net.profile((c, h, w))
    out = {'workload': f'TecoGAN 2xSR BI generator-only inference, synthetic {frames}-frame 3x268x640 LR clip -> '
                       f'3x536x1280 uint8 (BASELINE configs[4])',
           'value': frames / sorted(ts)[1], 'unit': 'frames/s', 'ms_per_step': 1e3 * sorted(ts)[1] / frames,
           'steps': frames, 'algorithmic_gflop_per_frame': gf['FNet'] + gf['SRNet'],
           'launches_per_frame': L.lib().tg_frnet_plan_launches(plan.handle)}
    mf = [r for r in rows if r['kernel'].startswith(('conv3x3_mfma', 'conv3x3_wino'))]
    if mf:
        dom = max(mf, key=lambda r: r['ms_per_frame'])
        ach = dom['tflops']
        wino = dom['kernel'].startswith('conv3x3_wino')
        out['roofline'] = {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS,
                           'unit': 'TFLOP/s', 'launches_per_frame': dom['launches'],
                           'avg_launch_us': 1e3 * dom['ms_per_frame'] / dom['launches'],
                           'algorithmic_gflop_per_launch': dom['gflop'] / dom['launches'],
                           'traffic': pmc_traffic(dom['kernel'], '2xBI'),
                           'traffic_source': 'profiles/pmc_traffic.json (committed rocprofv3 --pmc passes, tools/gpu_pmc.sh); NOT measured in this run'}
        if wino:     # algorithmic FLOPs may exceed the pipe's peak in this form: the statement about the
            #          matrix pipe is the executed fraction, the one about the form its own ceiling
            out['roofline']['frac_vs_winograd_ceiling'] = ach / (MFMA_F32_PEAK_TFLOPS * 2.25)
            out['roofline']['frac'] = ach * 16.0 / 36.0 / MFMA_F32_PEAK_TFLOPS
            out['roofline']['frac_is'] = ('executed MFMA FLOPs (16/36 of the algorithmic ones) / peak; the '
                                          'algorithmic rate `achieved` may exceed `peak` in the Winograd form')
        else:
            out['roofline']['frac'] = ach / MFMA_F32_PEAK_TFLOPS
    return out


def parity_check(dev, c, h, w, s, deg):
    """Outside every timed region: (a) FRNet.step with the procedural parity weights on the seeded inputs of
    tests/golden/fullsize.npz -- a digest the REFERENCE produced (tests/golden/make_golden.py) -- through the
    same kernels the bench times (the LDS-resident SRNet body at 134x320); (b) the bench path itself,
    infer_sequence(pipeline=True), against the frame-by-frame step() chain on a 4-frame clip (uint8: at most one
    level on <= 0.2 % of a frame -- the 8-pair batched flow pass picks the Winograd form for layers the
    one-pair pass runs in the direct form, so the two are equal up to fp32 summation order, not bit for bit)."""
    import numpy as np
    from procedural_weights import generator_state_dict
    from tecogan_pytorch_amd.models.networks import FRNet
    out = {}
    try:
        tagk = {('BD', 4, 134, 320): 'A', ('BI', 2, 268, 640): 'E'}.get((deg, s, h, w))
        net = FRNet(c, c, 64, 10, deg, s)
        net.load_state_dict(generator_state_dict(scale=s, degradation=deg), strict=True)
        net = net.to(dev).eval()

        def rs(seed, shape):
            return torch.from_numpy(np.random.RandomState(seed).uniform(0.0, 1.0, shape).astype(np.float32)).to(dev)
        with torch.no_grad():
            if tagk is not None:
                g = np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize.npz'))
                o = net.step(rs(100, (1, c, h, w)), rs(101, (1, c, h, w)), rs(102, (1, c, s * h, s * w)))
                flat = o.double().cpu().reshape(-1)
                idx = torch.from_numpy(g[f'full_{tagk}_sample_idx'])
                e_s = float(np.abs(flat[idx].float().numpy() - g[f'full_{tagk}_samples']).max())
                e_m = abs(flat.mean().item() - float(g[f'full_{tagk}_mean']))
                out['step_vs_reference_digest'] = {'max_abs_err_257_samples': e_s, 'mean_err': e_m,
                                                   'ok': bool(e_s <= 3e-4 and e_m <= 1e-5)}
            clip = torch.rand(4, c, h, w, generator=torch.Generator().manual_seed(5)).to(dev)
            u8 = net.infer_sequence(clip, dev, pipeline=True, return_device_tensor=True)
            hr = torch.zeros(1, c, s * h, s * w, device=dev)
            prev = torch.zeros(1, c, h, w, device=dev)
            worst, differ = 0, 0.0
            for t in range(clip.shape[0]):
                hr = net.step(clip[t:t + 1], prev, hr)
                prev = clip[t:t + 1]
                q = (hr[0] * 255.0).round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0)
                d = (q.int() - u8[t].int()).abs()
                worst = max(worst, int(d.max()))
                differ = max(differ, float((d > 0).float().mean()))
            torch.cuda.synchronize()
            net.check_faults()
            out['pipelined_clip_vs_step_chain'] = {'max_uint8_levels': worst, 'max_fraction_differing': differ,
                                                   'ok': bool(worst <= 1 and differ <= 2e-3)}
        out['ok'] = all(v['ok'] for v in out.values() if isinstance(v, dict))
    except Exception as e:       # reported, never fatal: the parity tests proper are tests/ -m gpu
        out = {'ok': False, 'error': repr(e)[:300]}
    return out


def _barrier(dist, local_rank):
    if dist.get_backend() == 'nccl':
        dist.barrier(device_ids=[local_rank])
    else:
        dist.barrier()


def ddp_train_leg(args, dev, rank, world, local_rank, dist_on):
    """BASELINE configs[3] (and configs[2] at N = 1): the full TecoGAN training step --
    prepare_training_data (on-device BD) + VSRGANModel.train() (G forward/BPTT, 3 D passes,
    adaptive D update, 2 fused Adam steps) -- data parallel over the ranks of this launch:
    per-rank batch 2 x 10 frames (-> 19 with ping-pong) at the REDS crop 128, seeds 0 + rank,
    gradients of G (and of D when it updates) averaged by ONE flat all-reduce each over RCCL,
    SyncBatchNorm statistics and the fused adaptive-D scalar exchanged per step.  Reports
    MAX-over-ranks ms/step, clips/s, and the all-reduce of the two gradient buckets timed on
    its own (HIP events around 20 back-to-back calls): ms per call and ring bus bandwidth
    2 (N-1)/N x bytes / t.  Context for the north_star's multi-GPU split; not `value`."""
    import torch.distributed as dist
    from tecogan_pytorch_amd.models import define_model

    def opt_for(crop):
        return {
            'scale': 4, 'dist': dist_on and world > 1, 'device': 'cuda', 'rank': rank, 'world_size': world,
            'is_train': True,
            'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5}, 'train': {'crop_size': crop}},
            'model': {'name': 'TecoGAN',
                      'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10},
                      'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3}},
            'train': {'tempo_extent': 10, 'ckpt_dir': '/tmp',
                      'generator': {'lr': 5e-5, 'betas': [0.9, 0.999]},
                      'discriminator': {'update_policy': 'adaptive', 'update_threshold': 1e9,
                                        'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                      'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                      'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                      'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'},
                      'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
            'logger': {'decay': 0.99},
        }

    def run(crop, steps, warm=8):
        torch.manual_seed(0 + rank)                       # base_utils.py:46
        m = define_model(opt_for(crop))                   # broadcasts rank 0's weights under DDP
        if dist_on and world > 1:
            m.exchange_timing = {}                        # events around the waits for both gradient exchanges
        gen = torch.Generator().manual_seed(1 + rank)
        data = [{'gt': torch.rand(2, 10, 3, crop + 8, crop + 8, generator=gen).to(dev)} for _ in range(2)]
        for i in range(warm):
            m.prepare_training_data(data[i % 2]); m.train()
        if getattr(m, 'exchange_timing', None) is not None:
            m.exchange_timing.clear()
        if dist_on:
            _barrier(dist, local_rank)
        from tecogan_pytorch_amd.utils import dist_utils as DU
        c0 = dict(DU.COMM_COUNTS)
        import gc
        gc.collect()                              # (a collection of this long-lived process inside 10 steps is 5 % of them)
        nupd0 = getattr(m, 'cnt_upd_D', 0.0)
        blocks = []
        for _ in range(3):                        # three blocks of `steps` iterations, the median block is reported
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):                # (the log is NOT read per step: like a run that prints every 100 iterations)
                m.prepare_training_data(data[i % 2]); m.train()
            m.sync_log()                          # the last iteration's scalars and fault check
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / steps)
        nupd = int(getattr(m, 'cnt_upd_D', 0.0) - nupd0) // 3
        dt = sorted(blocks)[1]
        m.train_blocks_ms = [1e3 * b for b in blocks]
        m.comm_per_step = {k: (DU.COMM_COUNTS[k] - c0[k]) / (3 * steps) for k in c0}
        if dist_on:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return m, dt, nupd

    out = {}
    m, dt, nupd = run(128, args.train_steps)
    from tecogan_pytorch_amd.utils import dist_utils as DU
    if dist_on:
        # readiness checks of the first real multi-GPU run: every rank must see the world the driver asked for
        seen = dist.get_world_size()
        assert seen == world == args.gpus or os.environ.get('TG_BENCH_REHEARSAL') == '1' and seen == world, \
            f'rank {rank}: process group of {seen} ranks, --gpus {args.gpus}, WORLD_SIZE {world}'
        out['process_group'] = {'world_seen_by_rank0': seen, 'backend': dist.get_backend(),
                                'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version())
                                if dist.get_backend() == 'nccl' else None,
                                'transport': os.environ.get('TECOGAN_COMM', 'torch.distributed'),
                                # what every rank compared at model construction (BaseModel.check_ranks_agree: one
                                # all-gather; a mixed build refuses to start instead of hanging in the first bucket)
                                'agreement_vector': m.agreement_vector(),
                                'agreement_vector_is': '[tg_version, chained-launch parts at the training shape, '
                                                       'pair_pass, c_abi transport, G bucket floats, D bucket floats]'}
        if os.environ.get('TECOGAN_COMM', '') == 'c_abi':
            import ctypes
            from tecogan_pytorch_amd import _lib as L_
            seen_c, rank_c = ctypes.c_int(-1), ctypes.c_int(-1)
            if L_.lib().tg_comm_query(DU._c_comm(), ctypes.byref(seen_c), ctypes.byref(rank_c)) == 0:
                out['process_group']['ranks_seen'] = seen_c.value          # ncclCommCount: RCCL's own answer
                assert seen_c.value == world and rank_c.value == rank, (seen_c.value, rank_c.value, world, rank)
        cps = getattr(m, 'comm_per_step', {})
        out['rccl_comm_count'] = {
            'all_reduce_per_step': cps.get('all_reduce'), 'all_gather_per_step': cps.get('all_gather'),
            'payload_bytes_per_step': cps.get('bytes'),
            'what': '2 flat gradient buckets (G, D) + 1 fused adaptive-D scalar pair + SyncBatchNorm: 4 layers x '
                    '(1 statistics all-gather forward + 1 sum all-reduce backward) for the D update (real and fake pass '
                    'run as ONE pair pass since round 4: 8 instead of 16) and the same for the generator\'s pass through D'}
        et = getattr(m, 'exchange_timing', None)
        if et:
            torch.cuda.synchronize()
            out['exchange_wait_ms_per_step'] = {
                k: sum(a.elapsed_time(b) for a, b in v) / max(1, len(v)) for k, v in et.items()}
            out['exchange_wait_note'] = ('time the COMPUTE stream stalled for a gradient exchange (events around the '
                                         'wait + mean kernel): D\'s all-reduce is launched before the D-independent '
                                         'generator losses and should be ~0 when it hides under them; G\'s is '
                                         'blocking by construction (nothing is left to overlap it with)')
    out.update({
        'workload': 'BASELINE configs[3]: TecoGAN 4xSR BD GAN training step, per-GPU batch 2 x 10 -> 19 '
                    'frames, crop 128 (REDS yml shape), synthetic U[0,1) GT, random-init weights, fp32; '
                    'prepare_training_data + train() with the discriminator updated EVERY step '
                    '(adaptive policy with threshold +inf: the full G + D step incl. both gradient exchanges)',
        'n_gpus': world, 'ms_per_step': 1e3 * dt, 'clips_per_s': world * 2 / dt,
        'hr_frames_per_s': world * 2 * 19 / dt, 'd_updates': nupd, 'steps': args.train_steps,
        'ms_per_step_blocks': getattr(m, 'train_blocks_ms', None), 'statistic': 'median of three blocks of `steps` iterations',
        'scaling': 'weak'})
    # the two gradient buckets on their own
    for name, optim in (('G', m.optim_G), ('D', m.optim_D)):
        buf = optim.flat_grad if optim.flat_grad is not None else torch.zeros(1 << 20, device=dev)
        nbytes = buf.numel() * 4
        if dist_on and world > 1:
            for _ in range(3):
                dist.all_reduce(buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            out[f'allreduce_{name}'] = {'bytes': nbytes, 'ms_per_call': ms,
                                        'bus_GBps': 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9}
        else:
            out[f'allreduce_{name}'] = {'bytes': nbytes, 'ms_per_call': None, 'bus_GBps': None,
                                        'note': 'single rank: no exchange'}
    del m
    if world == 1:
        m2, dt2, nupd2 = run(256, args.train_steps)
        out['config2_crop256'] = {
            'workload': 'BASELINE configs[2]: same step, 2 x 10 -> 19 frames, crop 256 (Vimeo yml shape)',
            'ms_per_step': 1e3 * dt2, 'hr_frames_per_s': 2 * 19 / dt2, 'd_updates': nupd2}
        del m2
    # (no torch.cuda.empty_cache() here: it segfaults under an RCCL process group on this stack)
    return out


def main():
    args = parse()
    if args.gpus < 1:
        _die(f'--gpus {args.gpus}')
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args))          # plain `python bench.py --gpus N`: bring up the N ranks here
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        # never print a line whose n_gpus is not what --gpus asked for
        _die(f'rank {rank}: --gpus {args.gpus} but WORLD_SIZE {world} in the environment')
    if os.environ.get('TG_BENCH_LAUNCH_PROBE') == '1' and 'RANK' in os.environ:
        return launch_probe(args)
    # under torch.distributed.run (RANK/WORLD_SIZE set) always go through the process group,
    # so the N = 1 launch exercises exactly the code the N > 1 launches use
    dist_on = world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # Rehearsal switch for a one-GPU box (tools/rehearse_multi_gpu.sh): TG_BENCH_REHEARSAL=1 puts
        # every rank on cuda:0 and exchanges through gloo (host staged), which walks every N > 1
        # code path of this file except RCCL itself.  Never set by the driver.
        if os.environ.get('TG_BENCH_REHEARSAL') == '1':
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend='gloo')
            # N processes on ONE GPU: the chained launches assume the process owns the device (every
            # workgroup a launch waits for must get a slot while the waiters spin); with several processes
            # time-sharing the CUs their fail-safe can trip (measured at 2 ranks: ~400 workgroups timed
            # out, the plan fell back -- INTEGRATION.md).  The rehearsal walks the per-layer paths instead.
            os.environ['TG_WINO_CHAIN'] = '0'
            os.environ['TG_WINO_RES'] = '0'
            from tecogan_pytorch_amd.models.networks.tecogan_nets import SRNet
            SRNet.chain_body = False
        else:
            if local_rank >= torch.cuda.device_count():
                _die(f'rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible')
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank if dist_on else 0)
    ranks_seen, distinct_gpus = 1, 1
    if dist_on:
        # the ranks RCCL itself carries (a sum all-reduce of ones) and the distinct GPUs behind them: `n_gpus` of the
        # line is THIS number, and it must be what --gpus asked for
        ranks_seen, distinct_gpus = ranks_seen_by_collective(dist, dev)
        if ranks_seen != args.gpus or dist.get_world_size() != args.gpus:
            _die(f'rank {rank}: the process group carries {ranks_seen} ranks (world {dist.get_world_size()}), --gpus {args.gpus}')
        if distinct_gpus != args.gpus and os.environ.get('TG_BENCH_REHEARSAL') != '1':
            _die(f'rank {rank}: {args.gpus} ranks on {distinct_gpus} distinct GPU(s)')

    from tecogan_pytorch_amd.models.networks import FRNet
    from tecogan_pytorch_amd import _lib as L
    L.lib()   # loud failure if the HIP library is missing

    c, h, w = [int(v) for v in args.lr_size.split('x')]
    s, deg = args.scale, args.degradation
    torch.manual_seed(0)                      # same random-init weights on every rank
    net = FRNet(c, c, 64, 10, deg, s).to(dev).eval()

    # per-rank clip data (base_utils.py:46: seed + rank)
    gen = torch.Generator(device='cpu').manual_seed(1234 + rank)
    pool = []
    for _ in range(4):
        pool.append([torch.rand(1, c, h, w, generator=gen).to(dev),
                     torch.rand(1, c, h, w, generator=gen).to(dev),
                     torch.rand(1, c, s * h, s * w, generator=gen).to(dev)])
    outs = [torch.empty(1, c, s * h, s * w, device=dev) for _ in range(2)]

    def barrier():
        if dist_on:
            _barrier(dist, local_rank)

    # ---- headline: clip inference (FRNet.infer_sequence) ------------------------------
    # a K-frame synthetic LR clip, resident in HBM; true recurrence (hr_prev = previous
    # output), uint8 quantisation on the device, FNet(t+1) overlapped with warp+SRNet(t)
    # on a second HIP stream; the uint8 result stays on the device (no D2H in the timer).
    clip = torch.rand(args.steps, c, h, w, generator=gen).to(dev)
    wclip = torch.rand(max(args.warmup, 2), c, h, w, generator=gen).to(dev)
    with torch.no_grad():
        pipe = not args.no_pipeline
        net.infer_sequence(wclip, dev, pipeline=pipe, return_device_tensor=True)     # W warm-up steps
        torch.cuda.synchronize()
        # plus two untimed clips of the timed shape: the first clips of a process also pay for the
        # allocator, the event ring and the runtime's stream -> hardware-queue assignment
        # (DESIGN.md section 9); steady state is reached from the second clip on
        for _ in range(2):
            net.infer_sequence(clip, dev, pipeline=pipe, return_device_tensor=True)
            torch.cuda.synchronize()
        # The timed region is EXACTLY K steps (one K-frame clip) between barrier + synchronize
        # brackets; it is repeated `--clips` times (default 10) and the MEDIAN region is the
        # reported one (min / max beside it): a single 20-60 ms region is at the mercy of one
        # scheduling hiccup, and box-to-box spread was already 2 % in round 1.
        times = []
        for _ in range(max(1, args.clips)):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net.infer_sequence(clip, dev, pipeline=pipe, return_device_tensor=True)  # exactly K steps
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)   # this rank's K steps; MAX over ranks below
            net.check_faults()                       # fail-safe of chained launches (a host read, after the sync)
            barrier()                                # closing bracket (its own latency is not a step)
            torch.cuda.synchronize()

        # ---- secondary protocols (rank-local, not part of `value`) -----------------------
        sec = {}
        if not args.no_pipeline and not args.no_secondary:
            net.infer_sequence(wclip, dev, pipeline=False, return_device_tensor=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            net.infer_sequence(clip, dev, pipeline=False, return_device_tensor=True)
            torch.cuda.synchronize()
            sec['fps_clip_single_stream'] = args.steps / (time.perf_counter() - t1)
            # one stream, flows batched: the 8-pair FNet passes enqueued on the SAME stream ahead of their frames
            # (no concurrency of any kind; the per-frame-FNet figure above keeps the reference's loop shape)
            net.infer_sequence(wclip, dev, pipeline='one_stream', return_device_tensor=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            net.infer_sequence(clip, dev, pipeline='one_stream', return_device_tensor=True)
            torch.cuda.synchronize()
            sec['fps_clip_one_stream_batched_flow'] = args.steps / (time.perf_counter() - t1)
            # with host I/O, as the reference's loop has it (tecogan_nets.py:273-279 moves every
            # frame H2D and the uint8 result D2H): LR clip in pinned host memory, uploaded batch by
            # batch, uint8 HR frames downloaded to pinned memory on a copy stream while the next
            # batch computes; one synchronisation at the end.  PCIe-inclusive: never `value`.
            clip_host = clip.cpu().pin_memory()
            net.infer_sequence(clip_host, dev)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                net.infer_sequence(clip_host, dev)
                ts.append(time.perf_counter() - t1)
            sec['fps_with_h2d_d2h'] = args.steps / sorted(ts)[1]
            nstep = min(args.steps, 60)
            for i in range(4):
                net.step(*pool[i % 4], out=outs[i & 1])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(nstep):               # independent random frames, no sync between
                net.step(*pool[i % 4], out=outs[i & 1])
            torch.cuda.synchronize()
            sec['fps_step_protocol_no_sync'] = nstep / (time.perf_counter() - t1)
            # several independent clips per GPU through one launch list (plan batch n = 4):
            # clip-level data parallelism INSIDE the GPU fills the tile-quantisation holes
            nb = 4
            bpool = [torch.rand(nb, c, h, w, generator=gen).to(dev) for _ in range(2)] + \
                    [torch.rand(nb, c, s * h, s * w, generator=gen).to(dev)]
            bout = torch.empty(nb, c, s * h, s * w, device=dev)
            for _ in range(3):
                net.step(bpool[0], bpool[1], bpool[2], out=bout)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nbs = max(4, nstep // nb)
            for _ in range(nbs):
                net.step(bpool[0], bpool[1], bpool[2], out=bout)
            torch.cuda.synchronize()
            sec['fps_step_protocol_4_clips_batched'] = nb * nbs / (time.perf_counter() - t1)
            # the same four clips through the pipelined clip inference (true recurrence per clip,
            # batched FNet on the side stream, batched uint8 output): the serving configuration
            clips4 = torch.rand(nb, args.steps, c, h, w, generator=gen).to(dev)
            net.infer_sequence(clips4, dev, return_device_tensor=True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                net.infer_sequence(clips4, dev, return_device_tensor=True)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            net.check_faults()
            sec['fps_4_clips_pipelined'] = nb * args.steps / sorted(ts)[1]
            del clips4
            for kc in (2, 8):                    # the other serving points DESIGN.md quotes
                ck = torch.rand(kc, args.steps, c, h, w, generator=gen).to(dev)
                net.infer_sequence(ck, dev, return_device_tensor=True)
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    t1 = time.perf_counter()
                    net.infer_sequence(ck, dev, return_device_tensor=True)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t1)
                net.check_faults()
                sec[f'fps_{kc}_clips_pipelined'] = kc * args.steps / sorted(ts)[1]
                del ck
            # reference protocol: synchronise after every frame (main.py:257-259)
            tsync = 0.0
            nsync = min(args.steps, 30)
            for i in range(nsync):
                t1 = time.perf_counter()
                net.step(*pool[i % 4], out=outs[i & 1])
                torch.cuda.synchronize()
                tsync += time.perf_counter() - t1
            sec['fps_step_protocol_sync_every_frame'] = nsync / tsync

    # ---- BASELINE configs[4]: 2x BI at 3x268x640 (the other up-sampler, warp stride 2, and the shape
    # whose SRNet runs as ONE chained Winograd launch), under the same clock as the headline ------
    cfg5 = None
    if rank == 0 and not args.no_secondary and (s, deg, (c, h, w)) == (4, 'BD', (3, 134, 320)):
        try:
            cfg5 = config5_leg(dev, gen, min(args.steps, 60))
        except Exception as e:          # a context leg: never lose the headline line to it
            cfg5 = {'error': repr(e)[:300]}
    if dist_on:
        t = torch.tensor(times, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)     # per region: the slowest rank
        times = t.tolist()
    elapsed = sorted(times)[len(times) // 2] if len(times) % 2 else \
        0.5 * (sorted(times)[len(times) // 2 - 1] + sorted(times)[len(times) // 2])
    train_leg = None
    if not args.no_train_leg:
        try:
            train_leg = ddp_train_leg(args, dev, rank, world, local_rank, dist_on)
        except Exception as e:          # a context leg: never lose the headline line to it
            train_leg = {'error': repr(e)[:300]}

    result = None
    if rank == 0:
        gf, _ = net.profile((c, h, w))
        plan = net._get_plan(1, h, w, dev)
        cfg_name = ('configs[1]' if (s, deg, (c, h, w)) == (4, 'BD', (3, 134, 320)) else
                    'configs[4]' if (s, deg, (c, h, w)) == (2, 'BI', (3, 268, 640)) else 'shape, not a named config')
        result = {
            'metric': 'HR frames/sec/GPU at 4xSR 3x134x320 LR; Vid4 PSNR vs reference',
            'value': world * args.steps / elapsed,
            'unit': 'frames/s',
            'n_gpus': ranks_seen, 'steps': args.steps, 'warmup': args.warmup,
            'ranks_seen': ranks_seen, 'distinct_gpus': distinct_gpus,
            'launched_by': (os.environ.get('TG_BENCH_LAUNCHED_BY', 'torch.distributed.run') if dist_on else 'plain process')
                           + (' (TG_BENCH_REHEARSAL: every rank on cuda:0, gloo)' if os.environ.get('TG_BENCH_REHEARSAL') == '1' else ''),
            'ms_per_step': 1e3 * elapsed / args.steps,
            'timed_regions': len(times), 'ms_per_step_min': 1e3 * min(times) / args.steps,
            'ms_per_step_max': 1e3 * max(times) / args.steps, 'statistic': 'median of the timed regions',
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'TecoGAN {s}xSR {deg} generator-only inference '
                                   f'(FRNet.infer_sequence) of a synthetic {args.steps}-frame '
                                   f'{c}x{h}x{w} LR clip -> {c}x{s*h}x{s*w} HR uint8, 1 clip/GPU, '
                                   f'random-init weights (BASELINE {cfg_name}); a step = one '
                                   f'recurrent frame',
                       'parallelism': f'clip-sharded x{world}, no data-path collective',
                       'algorithmic_gflop_per_frame': gf['FNet'] + gf['SRNet'],
                       'launches_per_frame': L.lib().tg_frnet_plan_launches(plan.handle)},
            'pipelined': not args.no_pipeline,
            'published_reference': '27 FPS on 1x GTX 1080 Ti (README benchmark.png); other hardware, '
                                   'not a baseline for vs_baseline',
        }
        result.update(sec)
        if not args.no_parity_check:
            pc = parity_check(dev, c, h, w, s, deg)
            result['parity_check'] = 'ok' if pc.get('ok') else 'FAILED'
            result['parity_check_detail'] = pc
        if train_leg is not None:
            result['train_ddp'] = train_leg
        if cfg5 is not None:
            result['config5_2xBI'] = cfg5
        if not args.no_roofline:
            with torch.no_grad():
                rows = kernel_table(net, plan, (*pool[0], outs[0]))
            dom = max(rows, key=lambda r: r['ms_per_frame'])
            mf = [r for r in rows if r['kernel'].startswith(('conv3x3_mfma', 'conv3x3_wino'))]
            dom_mf = max(mf, key=lambda r: r['ms_per_frame'])
            ach = dom_mf['tflops']
            result['roofline'] = {
                'bound': 'mfma', 'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / MFMA_F32_PEAK_TFLOPS, 'frac_executed': ach / MFMA_F32_PEAK_TFLOPS,
                'frac_algorithmic': ach / MFMA_F32_PEAK_TFLOPS,       # (a direct form executes what it counts)
                'traffic': pmc_traffic(dom_mf['kernel']),
                'traffic_source': 'profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this '
                                  'workload, tools/gpu_pmc.sh); NOT measured in this run',
                'kernel': dom_mf['kernel'], 'launches_per_frame': dom_mf['launches'],
                'avg_launch_us': 1e3 * dom_mf['ms_per_frame'] / dom_mf['launches'],
                'algorithmic_gflop_per_launch': dom_mf['gflop'] / dom_mf['launches'],
            }
            result['roofline']['traffic_stale'] = pmc_traffic_stale()
            ct_gflop = 0.0                 # direct (non-Winograd) FLOPs inside the dominant launch
            if dom_mf['kernel'] == 'conv3x3_wino_resident_kernel':
                nlay = 1 + 2 * 10
                result['roofline']['layers_per_launch'] = nlay
                plain_ct = [r for r in rows if r['kernel'] == 'convt3x3s2_mfma_kernel' and r['launches'] > 0]
                if s == 4 and not plain_ct:
                    # SRNet's first ConvTranspose2d runs as the launch's tail (direct fp32 MFMA products)
                    ct_gflop = 2.0 * 64 * 9 * 64 * h * w / 1e9
                    result['roofline']['tail'] = {'layer': 'ConvTranspose2d(64, 64, 3, 2, 1, 1) + ReLU on the resident blocks',
                                                  'gflop': ct_gflop, 'form': 'direct fp32 MFMA (executed = algorithmic)',
                                                  'separate_launch': 'TG_WINO_RES_CT=0'}
                result['roofline']['avg_layer_us'] = 1e3 * dom_mf['ms_per_frame'] / nlay
                result['roofline']['avg_layer_us_is'] = 'launch time / 21 conv layers' + (' (the tail included)' if ct_gflop else '')
                result['roofline']['note'] = ('ONE persistent launch per frame: SRNet conv_in + 20 residual-block convs on '
                                              'LDS-resident 8x24-pixel blocks (tg_conv3x3_wino_res.hip); per-layer launches of '
                                              'the same arithmetic: TG_WINO_RES=0')
            if dom_mf['kernel'].startswith('conv3x3_wino'):
                # `achieved` counts the ALGORITHMIC FLOPs of the 3x3 convolution (2*9*cin*cout per
                # pixel, the contract's definition); the Winograd F(2x2,3x3) form executes 16 of
                # every 36 of those multiplies on the matrix cores, so the share of the MFMA peak
                # the kernel actually occupies is frac * 16/36.
                result['roofline']['form'] = ('Winograd F(2x2,3x3), fp32 MFMA 16x16x4: 16 of 36 algorithmic '
                                              'multiplies are executed')
                # (a transposed-conv tail inside the launch is direct: its FLOPs are executed one for one)
                g_all = dom_mf['gflop'] / dom_mf['launches']
                ex_share = ((g_all - ct_gflop) * 16.0 / 36.0 + ct_gflop) / g_all
                result['roofline']['mfma_executed_tflops'] = ach * ex_share
                result['roofline']['mfma_executed_frac'] = ach * ex_share / MFMA_F32_PEAK_TFLOPS
                # against the form's own ceiling (every MFMA issue slot busy = 157.3 x 36/16 algorithmic)
                result['roofline']['frac_vs_winograd_ceiling'] = result['roofline']['mfma_executed_frac']
                # `frac` is the share of the fp32-MFMA peak the kernel actually occupies (executed FLOPs): the
                # algorithmic rate of a Winograd kernel may exceed the peak, and no line should print > 1
                # ONE stable pair of keys: `frac_executed` and `frac_algorithmic`; `frac` IS
                # `frac_executed`, for good (round 3 printed the algorithmic number under `frac`: trend readers beware)
                result['roofline']['frac_algorithmic'] = ach / MFMA_F32_PEAK_TFLOPS
                result['roofline']['frac_executed'] = result['roofline']['mfma_executed_frac']
                result['roofline']['frac'] = result['roofline']['frac_executed']
                result['roofline']['frac_is'] = ('executed MFMA FLOPs (16/36 of the algorithmic ones) / peak; `achieved` is the '
                                                 'ALGORITHMIC rate of the 3x3 convolutions (SURVEY 8d) and may exceed `peak` in this form')
            warp = [r for r in rows if r['kernel'].startswith('flowup_warp')]
            if warp:
                wk = warp[0]
                result['roofline_warp'] = {
                    'bound': 'hbm', 'achieved': wk['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': wk['gbs'] / HBM_PEAK_GBS, 'traffic': pmc_traffic(wk['kernel']),
                    'traffic_source': 'profiles/pmc_traffic.json (committed PMC passes); NOT measured in this run',
                    'kernel': wk['kernel'], 'avg_launch_us': 1e3 * wk['ms_per_frame'],
                    'algorithmic_mbytes_per_launch': wk['mbytes']}
                # what a plain copy of the same bytes with the same grid reaches (back-to-back launches between two events,
                # exactly how the kernel table times the warp launch: a launch boundary is inside every sample of both)
                cc1 = copy_ceiling(dev, int(wk['mbytes'] * 1e6) // 2, -(-s * w // 256) * h, 512 if s == 4 else 256)
                result['roofline_warp']['copy_ceiling'] = cc1
                result['roofline_warp']['frac_of_copy_ceiling'] = wk['gbs'] / cc1['GBps']
                result['roofline_warp_batched'] = warp_batched_roofline(
                    dev, h, w, s, deg)
            # roofline.traffic MEASURED IN THIS RUN (it used to come from the committed passes only): two
            # rocprofv3 --pmc child passes of ~15 s each; the committed value stays beside it
            if world == 1 and not dist_on and not args.no_live_pmc:
                live = live_pmc_traffic(args.lr_size, s, deg)
                for key in ('roofline', 'roofline_warp'):
                    ro = result.get(key)
                    if ro is None:
                        continue
                    lv = _live_lookup(live, ro['kernel'])
                    ro['traffic_committed'] = ro['traffic']
                    if lv is not None:
                        ro['traffic'] = lv
                        ro['traffic_source'] = ('measured in THIS run: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (own passes, '
                                                '--kernel-trace only) over a 1-clip single-stream child run of this script; '
                                                'raw counters KB -> B per launch (MI355X_MICROARCH.md: FETCH_SIZE under-counts '
                                                '16-byte-per-lane reads 2x on gfx950, WRITE_SIZE uncalibrated); '
                                                '`traffic_committed` = profiles/pmc_traffic.json')
                    else:
                        ro['traffic_live'] = 'unavailable (no rocprofv3, or a counter pass failed / timed out): committed passes used'
            result['kernels'] = rows
            result['gpu_ms_per_frame_sum_of_kernels'] = sum(r['ms_per_frame'] for r in rows)
            result['slowest_kernel_class'] = dom['kernel']
        # context baselines only in a plain process (MIOpen's find segfaults next to an RCCL process
        # group on this stack; the driver's N = 1 run is a plain process)
        if world == 1 and not dist_on and args.aten_frames > 0:
            sd_dev = {k: v.detach() for k, v in net.state_dict().items()}
            # the reference's profile mode sets cudnn.benchmark = True (main.py:216), which
            # on ROCm is MIOpen's exhaustive find; report the better of both settings
            best = None
            for bm in (False, True):
                torch.backends.cudnn.benchmark = bm
                r = aten_gpu_baseline(sd_dev, s, deg, c, h, w, args.aten_frames, dev)
                r['cudnn_benchmark'] = bm
                if best is None or r.get('value', 0) > best.get('value', 0):
                    best = r
            torch.backends.cudnn.benchmark = False
            result['aten_gpu_baseline'] = best
            if 'value' in result['aten_gpu_baseline'] and 'fps_step_protocol_sync_every_frame' in result:
                result['vs_aten_gpu'] = (result['fps_step_protocol_sync_every_frame'] /
                                         result['aten_gpu_baseline']['value'])
        if world == 1 and args.cpu_frames > 0:
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            result['cpu_baseline'] = cpu_baseline(sd, s, deg, c, h, w, args.cpu_frames,
                                                  args.cpu_seconds)
            result['gpu_vs_cpu'] = result['value'] / result['cpu_baseline']['value']
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result), flush=True)
    if dist_on:
        _barrier(dist, local_rank)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
