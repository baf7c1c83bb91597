export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o kt -- python $REPO/tools/bench_train.py --crop 256 --steps 4 --warmup 2 --force-d > $REPO/gpurun_out/prof_train.log 2>&1
tail -1 $REPO/gpurun_out/prof_train.log | cut -c1-200
f=$REPO/gpurun_out/prof_train/kt_kernel_stats.csv
head -22 $f | cut -c1-170
python - <<PY
import csv
rows=list(csv.DictReader(open('$REPO/gpurun_out/prof_train/kt_kernel_trace.csv')))
tot=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
span=int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp'])
print('launches',len(rows),'sum kernel ms',tot/1e6,'span ms',span/1e6)
PY
