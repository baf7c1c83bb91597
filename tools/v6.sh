#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== bench path + train fullsize"; timeout 1200 python -m pytest tests/test_hip_parity_long.py -x -q -k "bench_path or train_step" 2>&1 | tail -5
cat $OUT/train_grad_rel_l2_crop*.json
