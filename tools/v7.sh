#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== new train tests"; timeout 1200 python -m pytest tests/test_hip_train.py tests/test_hip_train_ops.py -x -q -k "guarded or drops or deep_body or fault_is_reported or two_iterations" 2>&1 | tail -15
echo "== dist"; timeout 900 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -3
