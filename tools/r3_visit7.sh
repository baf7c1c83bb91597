#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== chain tests"; timeout 900 python -m pytest tests/test_hip_train_ops.py -m gpu -q -k "srnet_body" 2>&1 | tail -25
