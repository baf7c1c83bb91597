#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_hip_train_ops.py tests/test_hip_parity.py tests/test_hip_train.py -m gpu -q -x 2>&1 | tail -5
echo "== train"; for c in 128 256; do timeout 300 python tools/bench_train.py --crop $c --steps 10 --force-d 2>/dev/null | tail -1 | cut -c1-200; done
echo "== train profile crop 256"; bash tools/prof_train.sh 256 2>&1 | grep -v "^W2026" | head -30 | cut -c1-150
