#!/bin/bash
# round-3 visit 1: fail-safe + dedicated stream + baseline training profile at crop 128
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
echo "== pipe probe: plain / HIP initialised first / next to RCCL / DYN=1"
for cfg in "DYN=0" "DYN=0 INIT_FIRST=1" "DYN=0 RCCL=1" "DYN=1" "DYN=0 PIPE=0"; do
  env $cfg ITERS=9 timeout 200 python tools/pipe_probe.py 2>&1 | tail -1
done
echo "== train baseline"; for c in 128 256; do timeout 300 python tools/bench_train.py --crop $c --steps 10 --force-d 2>/dev/null | tail -1 | cut -c1-300; done
echo "== train profile crop 128"; bash tools/prof_train.sh 128 2>&1 | tail -60
