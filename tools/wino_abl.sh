# Ablations of the Winograd kernel (needs tools/build_lab_libs.sh first): kernel time with parts switched off
#   ABLS="0 1 2 4 16 31" SHAPE="134 320 64 64" bash tools/wino_abl.sh
export TECOGAN_HIP_LIB=$(pwd)/tools/_lab_libs/libtecogan_wino_lab.so
for a in ${ABLS:-0 1 2 4 16 31}; do echo "ABL $a"; TG_WINO_ABL=$a bash tools/wino_prof.sh ${SHAPE:-134 320 64 64} 2>&1 | grep -E "wino_kernel"; done
