# kernel-level times of the Winograd lab (rocprofv3 --kernel-trace --stats); raw traces deleted
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wprof -o kt -- python $REPO/tools/wino_lab.py "$@" > /tmp/wprof.log 2>&1
grep -E "TIME|err" /tmp/wprof.log
f=$(find /tmp/wprof -name 'kt_kernel_stats.csv' | head -1)
python - <<PY
import csv
for r in csv.DictReader(open('$f')):
    if 'conv3x3' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us min {int(r['MinNs'])/1e3:7.2f} max {int(r['MaxNs'])/1e3:7.2f}")
PY
