#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
for v in vm rr; do
echo "== variant $v"; TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_$v.so timeout 600 python -m pytest tests/test_hip_parity.py -q -k "resident_launch_equals" 2>&1 | grep -E "AssertionError|passed|failed" | head
done
