# A/B of resident-kernel variant libraries on ONE box, kernel alone (tools/wino_res_lab.py: us per 21-layer launch,
# bit-identity against the per-layer launches):  bash tools/ab_wres.sh name1 name2 ...   (tools/build_wres_var.sh builds them)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "$@"; do
  echo "== $v: $(TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_$v.so timeout 120 python $REPO/tools/wino_res_lab.py 2>&1 | grep -E 'bit-identical|resident:' | sed -e 's/per-layer launches: //' | tr '\n' '|' | cut -c1-420)"
done
done
