#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== train tests"; timeout 1200 python -m pytest tests/test_hip_train_ops.py tests/test_hip_train.py tests/test_hip_parity_long.py tests/test_dist_gpu.py tests/test_hip_feat_losses.py tests/test_disc_variants.py -m gpu -q -x 2>&1 | tail -8
echo "== train"; for c in 128 256; do timeout 300 python tools/bench_train.py --crop $c --steps 10 --force-d 2>/dev/null | tail -1 | cut -c1-200; done
python tools/time_ops.py --crop 256 2>&1 | grep "3, 256, 256\|sum"
