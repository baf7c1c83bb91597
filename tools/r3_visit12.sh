#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12
