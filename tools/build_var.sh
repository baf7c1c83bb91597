#!/bin/bash
# variant build of ONE source file: bash tools/build_var.sh tg_warp NAME "-DFLAG ..."  -> tools/_lab_libs/libtecogan_NAME.so
# (every other object comes from the regular build: run tecogan-pytorch_amd/csrc/build.sh first)
set -euo pipefail
cd "$(dirname "$0")/../tecogan-pytorch_amd/csrc"
OUT=../../tools/_lab_libs; mkdir -p $OUT
FF=$(sed -n "s/^\/\/ TG_FILE_FLAGS: *//p" $1.hip | head -1)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -DTG_LAB=1 $FF $3 -c $1.hip -o $OUT/$1_$2.o
OBJS=$(ls tg_*.o | grep -v "^$1.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtecogan_$2.so $OBJS $OUT/$1_$2.o -ldl
