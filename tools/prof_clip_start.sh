# Frame-to-frame intervals (start of the resident launch) of ONE 20-frame clip between two synchronisations, with the flow passes.
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pcs -o kt -- python $REPO/bench.py --steps 20 --warmup 5 --clips 3 --no-roofline --no-secondary --no-parity-check --no-train-leg --cpu-frames 0 --aten-frames 0 > /tmp/pcs.log 2>&1
f=$(find /tmp/pcs -name 'kt_kernel_trace.csv' | head -1)
python - <<PY
import csv
rows = sorted(csv.DictReader(open('$f')), key=lambda r: int(r['Start_Timestamp']))
res = [r for r in rows if 'resident' in r['Kernel_Name']]
last = res[-20:]
t0 = int(last[0]['Start_Timestamp'])
# first kernel of the clip = the first launch after the previous clip's last tail
prev_end = max(int(r['End_Timestamp']) for r in rows if int(r['End_Timestamp']) < t0 and 'convout_tail' in r['Kernel_Name'])
first = min(int(r['Start_Timestamp']) for r in rows if int(r['Start_Timestamp']) > prev_end)
print('gap previous clip end -> first kernel of this clip: %.1f us;  first kernel -> first resident launch: %.1f us' % ((first - prev_end) / 1e3, (t0 - first) / 1e3))
print('resident start-to-start (us):', [round((int(b['Start_Timestamp']) - int(a['Start_Timestamp'])) / 1e3) for a, b in zip(last, last[1:])])
end = max(int(r['End_Timestamp']) for r in rows if int(r['Start_Timestamp']) >= first)
print('clip span %.1f us = %.1f us per frame' % ((end - first) / 1e3, (end - first) / 20e3))
fn = [r for r in rows if int(r['Start_Timestamp']) >= first and ('wino_kernel' in r['Kernel_Name'] or 'small_ks' in r['Kernel_Name'])]
q = {}
for r in rows:
    if int(r['Start_Timestamp']) >= first:
        q.setdefault(r['Queue_Id'], []).append(r)
for k, v in q.items():
    print('queue', k, 'launches', len(v), 'first %.1f us  last end %.1f us' % ((int(v[0]['Start_Timestamp']) - first) / 1e3, (int(v[-1]['End_Timestamp']) - first) / 1e3),
          ' first kernel', v[0]['Kernel_Name'][:50])
sk = [r for r in rows if int(r['Start_Timestamp']) >= first and 'small_ks_kernel<2>' in r['Kernel_Name']]
print('flow passes end at (us):', [round((int(r['End_Timestamp']) - first) / 1e3) for r in sk])
PY
