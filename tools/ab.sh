# A/B of resident-kernel variant libraries on ONE box: bash tools/ab.sh name1 name2 ...   (tools/build_wres_var.sh builds them)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "$@"; do
  TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_wres_$v.so python $REPO/bench.py --steps 60 --warmup 10 --no-train-leg --no-secondary --no-parity-check --no-live-pmc --cpu-frames 0 --aten-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1))"
done
done
