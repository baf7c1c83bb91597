#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; cd $REPO
timeout 1500 python -m pytest tests/test_hip_train_ops.py tests/test_hip_train.py tests/test_hip_parity_long.py tests/test_r3_parity.py tests/test_hip_feat_losses.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
for c in 128 256; do timeout 300 python tools/bench_train.py --crop $c --steps 10 --force-d 2>/dev/null | tail -1 | cut -c1-170; done
bash tools/prof_train.sh 128 2>&1 | grep "s2_oneshot\|1, 2, 1, false"  | cut -c1-140
