export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_w -o kt -- python $REPO/bench.py --steps 8 --warmup 2 --cpu-frames 0 --aten-frames 0 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$REPO/gpurun_out/prof_w/kt_kernel_trace.csv')))
w=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows if 'flowup_warp' in r['Kernel_Name']]
# find runs of consecutive warp launches (replay): gaps
prev=None; out=[]
names=[r['Kernel_Name'][:30] for r in rows]
idx=[i for i,r in enumerate(rows) if 'flowup_warp' in r['Kernel_Name']]
runs=[]
for a,b in zip(idx,idx[1:]):
    if b==a+1:
        s0,e0=int(rows[a]['Start_Timestamp']),int(rows[a]['End_Timestamp'])
        s1=int(rows[b]['Start_Timestamp'])
        runs.append((e0-s0, s1-e0))
print('consecutive warp launches: n=',len(runs))
print('dur us (first 10):',[round(d/1000,1) for d,g in runs[:10]])
print('gap us (first 10):',[round(g/1000,1) for d,g in runs[:10]])
PY
