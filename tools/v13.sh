#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_r4_parity.py tests/test_hip_feat_losses.py tests/test_hip_train.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
