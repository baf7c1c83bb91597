#!/bin/bash
# whole-library variant with extra compiler flags for EVERY file: bash tools/build_all_var.sh NAME "-fno-slp-vectorize" -> tools/_lab_libs/libtecogan_NAME.so
set -euo pipefail
cd "$(dirname "$0")/../tecogan-pytorch_amd/csrc"
OUT=../../tools/_lab_libs/objs_$1; mkdir -p $OUT
PIDS=()
for f in tg_*.hip; do
  FF=$(sed -n "s/^\/\/ TG_FILE_FLAGS: *//p" $f | head -1)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -DTG_LAB=1 $FF $2 -c $f -o $OUT/${f%.hip}.o &
  PIDS+=($!)
done
for p in "${PIDS[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_lab_libs/libtecogan_$1.so $OUT/*.o -ldl
rm -rf $OUT
