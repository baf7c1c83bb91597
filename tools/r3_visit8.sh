#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
for hw in 64 32; do
HW=$hw python tools/wgrad_lab.py 2>&1 | grep hw=
for abl in 1 2 3 4 8 7; do HW=$hw TG_WGRAD_ABL=$abl TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wgabl$abl.so python tools/wgrad_lab.py 2>&1 | grep hw=; done
done
