#!/bin/bash
# Anatomy of the streaming Z-mode transposed conv: lab builds with -DZS_ABL=bits (timing only), tools/convtz_lab.py.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for abl in ${@:-0 1 2 4 3 7}; do
  TG_LAB_BUILD=1 OUT=$REPO/tools/_lab_libs/zabl$abl EXTRA_FLAGS="-DZS_ABL=$abl" bash tecogan-pytorch_amd/csrc/build.sh > /dev/null 2>&1 || { echo "lab build $abl failed"; continue; }
  echo "ZS_ABL=$abl: $(TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/zabl$abl/libtecogan_lab.so timeout 100 python tools/convtz_lab.py 2>&1 | grep 'form 2' | tail -1)"
done
