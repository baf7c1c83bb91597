# per-launch durations of one batched FNet pass (the last of 6), in launch order.  Usage: bash tools/fnet_layers.sh [batch]
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NB=${1:-8}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/fl_$NB -o kt -- python $REPO/tools/fnet_layers.py $NB > /tmp/fl_$NB.log 2>&1
python - <<PY
import csv, glob
tr = glob.glob('/tmp/fl_$NB/**/kt_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'pack' not in r['Kernel_Name'] and 'at::' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
per = len(rows) // 6
last = rows[-per:]
t0 = int(last[0]['Start_Timestamp'])
tot = 0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.1f} us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} x {r['Workgroup_Size_X']:4s} lds {r.get('LDS_Block_Size','?'):6s} {r['Kernel_Name'][:80]}")
print('batch $NB: launches', per, 'sum us', round(tot, 1), 'span us', (int(last[-1]['End_Timestamp']) - t0) / 1e3, 'per frame', round(tot / $NB, 1))
PY
