#!/bin/bash
# One short GPU visit while working on the training step: the training-op parity tests, the step at both
# BASELINE training shapes, and the per-call weight-gradient timings.
#   gpurun --timeout 900 -- 'bash tools/gpu_train_check.sh'
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_hip_train_ops.py tests/test_hip_train.py tests/test_r3_parity.py -m gpu -q -x 2>&1 | tail -4
python tools/bench_train.py --crop 256 --steps 6 --warmup 3 --force-d 2>&1 | tail -1 | cut -c1-200
python tools/bench_train.py --crop 128 --steps 6 --warmup 3 --force-d 2>&1 | tail -1 | cut -c1-200
python tools/time_ops.py --crop 256 --ops wgrad3x3_body,wgrad3x3_convt_multi,wgrad3x3,wgrad3x3_multi 2>&1 | tail -30
