#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
export TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_lp.so
echo "== tests"; timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "resident_launch_equals" 2>&1 | tail -2
timeout 120 python tools/wino_res_lab.py 2>&1 | grep -E "per-layer|identical" | tail -3
