export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o kt -- python $REPO/bench.py --steps 60 --warmup 10 --clips 3 --no-roofline --no-secondary --no-train-leg --cpu-frames 0 --aten-frames 0 > /tmp/pp.log 2>&1
tail -1 /tmp/pp.log | cut -c1-200
f=$(find /tmp/pp -name 'kt_kernel_stats.csv' | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open('$f')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:18]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} us {100*int(r['TotalDurationNs'])/tot:5.1f}%")
PY
