#!/bin/bash
# lab build of the LDS-resident SRNet body with its ablation switches (env TG_WRES_ABL):
#   TECOGAN_HIP_LIB=tools/_lab_libs/libtecogan_wres_lab.so python tools/wino_res_lab.py
set -euo pipefail
cd "$(dirname "$0")/../tecogan-pytorch_amd/csrc"
OUT=../../tools/_lab_libs; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -DTG_WRES_LAB=1 ${EXTRA_FLAGS:-} \
  -Rpass-analysis=kernel-resource-usage -c tg_conv3x3_wino_res.hip -o $OUT/tg_conv3x3_wino_res_lab.o 2>&1 | grep -E "VGPRs:|Spill|Scratch|error" || true
OBJS=$(ls tg_*.o | grep -v tg_conv3x3_wino_res.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_wres_lab.so $OBJS $OUT/tg_conv3x3_wino_res_lab.o -ldl
ls -la $OUT/libtecogan_wres_lab.so
