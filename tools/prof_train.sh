# rocprofv3 kernel trace of the TecoGAN training step; prints the class totals and every
# at::native / ATen kernel left in the step.  Raw traces are deleted (gpurun_out is size-capped).
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CROP=${1:-256}
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o kt -- python $REPO/tools/bench_train.py --crop $CROP --steps 4 --warmup 2 --force-d > $REPO/gpurun_out/prof_train.log 2>&1
tail -1 $REPO/gpurun_out/prof_train.log | cut -c1-200
f=$(find $REPO/gpurun_out/prof_train -name 'kt_kernel_stats.csv' | head -1)
head -22 $f | cut -c1-170
python - <<PY
import csv, glob, collections
tr = glob.glob('$REPO/gpurun_out/prof_train/**/kt_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('launches', len(rows), 'sum kernel ms', tot / 1e6, 'span ms', span / 1e6)
aten = collections.Counter(); aten_ns = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    if 'at::' in k or 'at_cuda' in k or 'Cijk' in k or 'miopen' in k.lower():
        key = k[:110]
        aten[key] += 1; aten_ns[key] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('ATen / library launches', sum(aten.values()), 'of', len(rows), ' time ms', sum(aten_ns.values()) / 1e6,
      '(all 6 steps incl. warmup + setup)')
for k, v in aten.most_common(25):
    print(f'{v:6d} {aten_ns[k]/1e3:9.1f} us  {k}')
# conv classes by grid size (which layer shapes they are)
g = collections.Counter(); gns = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    if 'conv3x3' in k or 'wgrad3x3' in k or 'convt3x3' in k or 'conv4' in k:
        key = (k.split('(')[0][-60:], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', ''))
        g[key] += 1; gns[key] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('conv launches by (kernel, grid, block): count, avg us, total ms')
for key, v in sorted(g.items(), key=lambda kv: -gns[kv[0]])[:28]:
    print(f'{v:6d} {gns[key]/v/1e3:8.1f} {gns[key]/1e6:8.2f}  {key}')
PY
cp $f $REPO/gpurun_out/train_crop${CROP}_kernel_stats.csv
rm -rf $REPO/gpurun_out/prof_train
