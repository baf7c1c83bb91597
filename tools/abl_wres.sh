# Anatomy of the resident kernel: lab build (tools/build_wres_var.sh lab "-DTG_WRES_LAB=1") with parts switched off
#   bash tools/abl_wres.sh "0 2 4 ..."      (bits: tg_conv3x3_wino_res.hip, WResArgs::abl)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for a in $1; do
  echo "ABL $a: $(TG_WRES_ABL=$a TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_lab.so timeout 120 python $REPO/tools/wino_res_lab.py 2>&1 | grep -E 'resident:' | tail -1 | sed -e 's/.*resident: //')"
done
