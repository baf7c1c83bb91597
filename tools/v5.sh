#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
echo "== tests"; timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "resident" 2>&1 | tail -2
for v in "" s8; do
  echo "== variant '$v'"; if [ -n "$v" ]; then export TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_$v.so; fi
  timeout 120 python tools/wino_res_lab.py 2>&1 | grep -E "per-layer|identical" | tail -2
done
