#!/bin/bash
# variant builds of the resident kernel: bash tools/build_wres_var.sh NAME "-DFLAG ..."  -> tools/_lab_libs/libtecogan_wres_NAME.so
set -euo pipefail
cd "$(dirname "$0")/../tecogan-pytorch_amd/csrc"
OUT=../../tools/_lab_libs; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -DTG_LAB=1 $(sed -n "s/^\/\/ TG_FILE_FLAGS: *//p" tg_conv3x3_wino_res.hip | head -1) $2 -c tg_conv3x3_wino_res.hip -o $OUT/tg_wres_$1.o
OBJS=$(ls tg_*.o | grep -v tg_conv3x3_wino_res.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_wres_$1.so $OBJS $OUT/tg_wres_$1.o -ldl
