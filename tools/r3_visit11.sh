#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== new gpu tests"; timeout 900 python -m pytest tests/test_r3_parity.py tests/test_hip_parity.py tests/test_hip_train_ops.py -m gpu -q -x -k "bi_two or spatial_discriminator_backward or trained_activation or srnet_body" 2>&1 | tail -15
echo "== bench default"; timeout 900 python bench.py > $OUT/r03_bench.json 2> $OUT/r03_bench.err; tail -3 $OUT/r03_bench.err | cut -c1-300
python - <<PY
import json
d = json.load(open('$OUT/r03_bench.json'))
for k in ('value', 'ms_per_step', 'fps_clip_single_stream', 'fps_2_clips_pipelined', 'fps_4_clips_pipelined', 'fps_8_clips_pipelined', 'fps_with_h2d_d2h'):
    print(k, d.get(k))
print('roofline', {k: v for k, v in d['roofline'].items() if k not in ('traffic_source', 'form')})
print('warp', d.get('roofline_warp', {}).get('frac'), d.get('roofline_warp_batched', {}).get('frac'))
print('config5', json.dumps(d.get('config5_2xBI'))[:900])
t = d.get('train_ddp', {})
print('train', t.get('ms_per_step'), t.get('config2_crop256', {}).get('ms_per_step'))
PY
