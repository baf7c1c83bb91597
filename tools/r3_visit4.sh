#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -30
echo "== stream kinds next to RCCL"
RCCL=1 KINDS=torch,plain timeout 300 python tools/stream_probe.py 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl"
RCCL=1 INIT_FIRST=1 KINDS=torch,plain timeout 300 python tools/stream_probe.py 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl"
echo "== pipe_probe (package default side stream): plain / init first / rccl"
for cfg in "DYN=0" "DYN=0 INIT_FIRST=1" "DYN=0 RCCL=1"; do env $cfg ITERS=9 timeout 200 python tools/pipe_probe.py 2>&1 | grep "^NF"; done
