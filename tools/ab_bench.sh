# A/B of whole-library variants on ONE box through bench.py's own lines:  bash tools/ab_bench.sh "key expr" lib1 lib2 ...
#   lib = "default" (the shipped library) or a name under tools/_lab_libs/libtecogan_NAME.so (tools/build_var.sh)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
EXPR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  LIBENV=""; [ "$v" != "default" ] && LIBENV="TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_$v.so"
  echo "$v: $(env $LIBENV python $REPO/bench.py --steps 40 --warmup 10 --no-train-leg --no-secondary --no-parity-check --no-live-pmc --cpu-frames 0 --aten-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($EXPR)")"
done
done
