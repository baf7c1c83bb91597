#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== chain tests"; timeout 1500 python -m pytest tests/test_hip_train_ops.py tests/test_hip_soak.py -x -q -k "chain or body or soak" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
echo "== train tests"; timeout 1500 python -m pytest tests/test_hip_train.py tests/test_r3_parity.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
echo "== long"; timeout 1500 python -m pytest tests/test_hip_parity_long.py -x -q -k "train_step" 2>&1 | grep -E "passed|failed" | tail -3
for c in 128 256; do timeout 300 python tools/bench_train.py --crop $c --steps 10 --force-d 2>/dev/null | tail -1 | cut -c1-200; done
