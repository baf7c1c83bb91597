#!/bin/bash
# Ablation builds of libtecogan_hip.so for measurements (selected with TECOGAN_HIP_LIB):
#   warp_abl1 = fused warp kernel without its stores, warp_abl2 = one tap row instead of two,
#   warp_abl4 = no flow loads.   Usage: bash tools/build_lab_libs.sh
set -euo pipefail
cd "$(dirname "$0")/../tecogan-pytorch_amd/csrc"
OUT=../../tools/_lab_libs
mkdir -p $OUT
# every lab object carries -DTG_LAB=1 (the sources #error on a lab switch without it) and the file's own TG_FILE_FLAGS,
# so the ablation numbers come from the same code generation as the product
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -Wno-unused-value -DTG_LAB=1"
ff() { sed -n 's/^\/\/ TG_FILE_FLAGS: *//p' "$1" | head -1; }
for abl in 1 2 4 3; do
  /opt/rocm/bin/hipcc $FLAGS $(ff tg_warp.hip) -DTG_WARP_ABL=$abl -c tg_warp.hip -o $OUT/tg_warp_abl$abl.o &
done
wait
OBJS=$(ls tg_*.o | grep -v tg_warp.o)
for abl in 1 2 4 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_warp_abl$abl.so $OBJS $OUT/tg_warp_abl$abl.o -ldl
done
ls -la $OUT/*.so
# Winograd kernel with its ablation switches (TG_WINO_ABL): TECOGAN_HIP_LIB=tools/_lab_libs/libtecogan_wino_lab.so
/opt/rocm/bin/hipcc $FLAGS $(ff tg_conv3x3_wino.hip) -DTG_WINO_LAB=1 -c tg_conv3x3_wino.hip -o $OUT/tg_conv3x3_wino_lab.o
OBJS=$(ls tg_*.o | grep -v tg_conv3x3_wino.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_wino_lab.so $OBJS $OUT/tg_conv3x3_wino_lab.o -ldl
# whole library with every lab switch compiled in (TG_LAB=1): TECOGAN_HIP_LIB=tools/_lab_libs/libtecogan_lab.so
mkdir -p $OUT/lab_objs
for f in tg_*.hip; do /opt/rocm/bin/hipcc $FLAGS $(ff $f) -c $f -o $OUT/lab_objs/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_lab.so $OUT/lab_objs/*.o -ldl
# LDS-resident SRNet body with its ablation switches (TG_WRES_ABL): TECOGAN_HIP_LIB=tools/_lab_libs/libtecogan_wres_lab.so
/opt/rocm/bin/hipcc $FLAGS $(ff tg_conv3x3_wino_res.hip) -DTG_WRES_LAB=1 -c tg_conv3x3_wino_res.hip -o $OUT/tg_conv3x3_wino_res_lab.o
OBJS=$(ls tg_*.o | grep -v tg_conv3x3_wino_res.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_wres_lab.so $OBJS $OUT/tg_conv3x3_wino_res_lab.o -ldl
