#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
for v in i1 i3; do
  echo "== variant '$v'"; export TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_$v.so
  timeout 120 python tools/wino_res_lab.py 2>&1 | grep -E "per-layer|identical" | tail -2
done
