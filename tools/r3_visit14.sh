#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
echo "== bench"; timeout 900 python bench.py --cpu-frames 0 --aten-frames 0 > $OUT/b14.json 2>/dev/null
python - <<PY
import json
d = json.load(open('$OUT/b14.json'))
for k in ('value', 'fps_clip_single_stream', 'fps_2_clips_pipelined', 'fps_4_clips_pipelined', 'fps_8_clips_pipelined'):
    print(k, d.get(k))
for k in ('roofline_warp', 'roofline_warp_batched'):
    print(k, {a: d[k][a] for a in ('frac', 'avg_launch_us')})
print('config5', d['config5_2xBI']['value'])
t = d.get('train_ddp', {})
print('train', t.get('ms_per_step'), t.get('config2_crop256', {}).get('ms_per_step'))
PY
