#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; cd $REPO
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_parity_long.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python bench.py --cpu-frames 0 --aten-frames 0 --no-train-leg > $OUT/b15.json 2>/dev/null
python - <<PY
import json
d = json.load(open('$OUT/b15.json'))
print('value', d['value'], 'single', d.get('fps_clip_single_stream'))
for k in ('roofline_warp', 'roofline_warp_batched'):
    print(k, {a: d[k][a] for a in ('frac', 'avg_launch_us')})
PY
