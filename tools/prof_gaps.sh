# Idle time between consecutive kernels of the single-stream clip inference (rocprofv3 kernel trace)
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o kt -- python $REPO/bench.py --steps 40 --warmup 5 --clips 2 --no-roofline --no-pipeline --no-secondary --no-train-leg --cpu-frames 0 --aten-frames 0 > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name 'kt_kernel_trace.csv' | head -1)
python - <<PY
import csv, collections
rows = sorted(csv.DictReader(open('$f')), key=lambda r: int(r['Start_Timestamp']))
# keep the last 40-frame clip: find the last 40 convout_tail launches
tails = [i for i, r in enumerate(rows) if 'convout_tail' in r['Kernel_Name']]
lo, hi = tails[-21], tails[-1]          # 20 whole frames
seg = rows[lo + 1:hi + 1]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
span = int(seg[-1]['End_Timestamp']) - int(rows[lo]['End_Timestamp'])
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows[lo:hi], rows[lo + 1:hi + 1])]
print('20 frames: launches', len(seg), 'per frame', len(seg) / 20, ' busy us/frame', busy / 20e3, ' span us/frame', span / 20e3,
      ' idle us/frame', (span - busy) / 20e3)
gs = sorted(gaps)
print('gap between consecutive kernels: median %.2f us  mean %.2f us  p90 %.2f us  max %.1f us' % (
    gs[len(gs) // 2] / 1e3, sum(gs) / len(gs) / 1e3, gs[int(0.9 * len(gs))] / 1e3, gs[-1] / 1e3))
PY
