#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; cd $REPO
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep "passed\|failed\|Error" | tail -3
for c in 128 256; do timeout 300 python tools/bench_train.py --crop $c --steps 10 --force-d 2>/dev/null | tail -1 | cut -c1-170; done
