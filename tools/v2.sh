#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
echo "== tests"; timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "resident" 2>&1 | tail -4
echo "== product"; timeout 120 python tools/wino_res_lab.py 2>&1 | grep -E "per-layer|identical" | tail -3
export TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_lab.so
for abl in 0; do
  echo "== abl $abl"; TG_WRES_ABL=$abl WRES_STAMPS=1 timeout 120 python tools/wino_res_lab.py 2>&1 | grep -v amdgpu.ids | grep -E "per-layer|wave  [015] layer  4|wave  8 layer  4|wave 11 layer  4|skew|K-loop end" | tail -9
done
