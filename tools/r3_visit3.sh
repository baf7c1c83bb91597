#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== failsafe test"; timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "chained_launch_fault" 2>&1 | tail -30
echo "== stream kinds"
timeout 300 python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids
INIT_FIRST=1 timeout 300 python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids
DEBUG_HIP_DYNAMIC_QUEUES=1 timeout 300 python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids
DEBUG_HIP_DYNAMIC_QUEUES=1 INIT_FIRST=1 timeout 300 python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids
echo "== pipe_probe (package default side stream): plain / init first / rccl"
for cfg in "DYN=0" "DYN=0 INIT_FIRST=1" "DYN=0 RCCL=1"; do env $cfg ITERS=9 timeout 200 python tools/pipe_probe.py 2>&1 | grep "^NF"; done
