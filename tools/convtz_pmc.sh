#!/bin/bash
# PMC anatomy of the Z-mode transposed conv's forms (tools/convtz_lab.py): two counter passes, per-kernel averages.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmcz_$i -o pmc -- python $REPO/tools/convtz_lab.py > $OUT/pmcz_$i.log 2>&1
  python - $OUT/pmcz_$i <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'convt3x3s2' not in k: continue
    acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print('   %-34s n=%3d mean %.4g' % (c, len(v), sum(v) / len(v)))
PY
  rm -rf $OUT/pmcz_$i
done
