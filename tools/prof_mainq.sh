# Pipelined clip inference: gaps between consecutive kernels of the MAIN queue (the one that runs the resident launch), by kernel pair.
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -o kt -- python $REPO/bench.py --steps 60 --warmup 10 --clips 2 --no-roofline --no-secondary --no-parity-check --no-train-leg --cpu-frames 0 --aten-frames 0 > /tmp/pq.log 2>&1
f=$(find /tmp/pq -name 'kt_kernel_trace.csv' | head -1)
python - <<PY
import csv, collections
rows = sorted(csv.DictReader(open('$f')), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.split('(')[0].replace('void tg::', '').replace('tg::', '')
    return n[:34]
qs = collections.Counter(r['Queue_Id'] for r in rows if 'resident' in r['Kernel_Name'])
mq = qs.most_common(1)[0][0]
main = [r for r in rows if r['Queue_Id'] == mq]
tails = [i for i, r in enumerate(main) if 'convout_tail' in r['Kernel_Name']]
lo, hi = tails[-41], tails[-1]
seg = main[lo:hi + 1]
pair = collections.defaultdict(list)
for a, b in zip(seg, seg[1:]):
    pair[(short(a['Kernel_Name']), short(b['Kernel_Name']))].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
dur = collections.defaultdict(list)
for r in seg[1:]:
    dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['End_Timestamp'])) / 40e3
print('main queue', mq, ': 40 frames, span us/frame %.1f' % span)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('  kernel %-36s n/frame %.2f  avg %.1f us  per frame %.1f us' % (k, len(v) / 40, sum(v) / len(v), sum(v) / 40))
for k, v in sorted(pair.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('  gap %-34s -> %-34s n %3d  median %.1f  mean %.1f  max %.1f  per frame %.1f us' % (k[0], k[1], len(v), v2[len(v) // 2], sum(v) / len(v), v2[-1], sum(v) / 40))
PY
