#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
export TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_convt_lab.so
for rows in 4 2; do
echo "== rows $rows"; TG_CONVTZ_ROWS=$rows timeout 600 python bench.py --steps 40 --warmup 10 --no-train-leg --cpu-frames 0 --aten-frames 0 --no-parity-check --no-secondary > $OUT/v10_bench_$rows.json 2>/dev/null; python - <<P
import json
j=json.load(open('gpurun_out/v10_bench_$rows.json'))
print(j['value'])
for r in j.get('kernels',[]):
    if 'convt' in r['kernel'] or 'tail' in r['kernel']: print('  ',r['kernel'], r['launches'], round(r['ms_per_frame']*1e3,1),'us', r['tflops'] and round(r['tflops'],1))
P
done
