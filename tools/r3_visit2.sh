#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== failsafe test"; timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "chained_launch_fault" 2>&1 | tail -40
echo "== stream kinds"
timeout 300 python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids
INIT_FIRST=1 timeout 300 python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids
DEBUG_HIP_DYNAMIC_QUEUES=1 timeout 300 python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids
