#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "warp or resident or winograd or step_vs_reference or fullsize" 2>&1 | tail -4
echo "== lab"; timeout 120 python tools/wino_res_lab.py 2>&1 | grep -E "per-layer|identical" | tail -2
echo "== bench"; timeout 600 python bench.py --steps 40 --warmup 10 --no-train-leg --cpu-frames 0 --aten-frames 0 > $OUT/v8_bench.json 2> $OUT/v8_bench.err; tail -3 $OUT/v8_bench.err; python - <<'P'
import json
j=json.load(open('gpurun_out/v8_bench.json'))
print(j['value'], j.get('parity_check'), j.get('fps_clip_single_stream'))
print('warp', j['roofline_warp']['frac'], j['roofline_warp']['avg_launch_us'], 'batched', j['roofline_warp_batched']['frac'], j['roofline_warp_batched']['avg_launch_us'])
for r in j.get('kernels',[]): print('  ',r['kernel'], r['launches'], round(r['ms_per_frame']*1e3,1),'us', r['tflops'] and round(r['tflops'],1))
P
