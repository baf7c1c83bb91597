#!/bin/bash
# The chained Winograd launch with agent-scope release / acquire fences around its flags (-DTG_CHAIN_FENCES=1)
# against the shipped form (sc1 stores, acknowledged waitcnt, relaxed flag): us per layer at 4 clips of 134x320.
#   build here:  bash tools/chain_fence_lab.sh build      run on the GPU box:  bash tools/chain_fence_lab.sh
set -euo pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-}" = build ]; then
  cd $REPO/tecogan-pytorch_amd/csrc; OUT=../../tools/_lab_libs; mkdir -p $OUT
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -DTG_CHAIN_FENCES=1 -c tg_conv3x3_wino.hip -o $OUT/tg_wino_fences.o
  OBJS=$(ls tg_*.o | grep -v tg_conv3x3_wino.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_fences.so $OBJS $OUT/tg_wino_fences.o -ldl
  exit 0
fi
cd $REPO
for lib in "" $REPO/tools/_lab_libs/libtecogan_fences.so; do
  echo "== lib '${lib:-shipped}'"
  TECOGAN_HIP_LIB=$lib python tools/wino_chain_probe.py 134 320 4 2>&1 | tail -3
  TECOGAN_HIP_LIB=$lib python tools/wino_chain_probe.py 268 640 1 2>&1 | tail -3
done
