# Idle gaps of the GPU inside ONE TecoGAN training step (crop 128): where the queue ran dry (rocprofv3 kernel trace)
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CROP=${1:-128}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ptg -o kt -- python $REPO/tools/bench_train.py --crop $CROP --steps 6 --warmup 3 --force-d > /tmp/ptg.log 2>&1
f=$(find /tmp/ptg -name 'kt_kernel_trace.csv' | head -1)
python - <<PY
import csv
rows = sorted(csv.DictReader(open('$f')), key=lambda r: int(r['Start_Timestamp']))
# steps are delimited by the Adam launches of the generator (2 adam_kernel per step: D then G)
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
ends = adam[1::2]                       # G's Adam = end of a step
lo, hi = ends[-3], ends[-1]             # two whole steps
seg = rows[lo + 1:hi + 1]
span = (int(seg[-1]['End_Timestamp']) - int(rows[lo]['End_Timestamp'])) / 2e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 2e3
print('per step: span %.1f us, busy %.1f us, idle %.1f us, launches %d' % (span, busy, span - busy, len(seg) // 2))
gaps = []
prev = rows[lo]
for r in seg:
    g = (int(r['Start_Timestamp']) - int(prev['End_Timestamp'])) / 1e3
    gaps.append((g, prev['Kernel_Name'].split('(')[0][-40:], r['Kernel_Name'].split('(')[0][-40:]))
    prev = r
big = sorted(gaps, reverse=True)[:14]
print('largest gaps (us) over two steps:')
for g, a, b in big:
    print('  %7.1f  %s -> %s' % (g, a, b))
import collections
h = collections.Counter()
for g, a, b in gaps:
    h['>=50' if g >= 50 else '10-50' if g >= 10 else '3-10' if g >= 3 else '1-3' if g >= 1 else '<1'] += g
print('idle us per step by gap size:', {k: round(v / 2, 1) for k, v in h.items()})
PY
