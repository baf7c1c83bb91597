# A/B of per-layer / chained Winograd kernel variants on ONE box: per-layer launches at 134x320 (wino_res_lab's first column),
# the batched flow pass (fnet_probe) and the 2xBI clip (the chained launch).  bash tools/ab_wino.sh default NAME ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "$@"; do
  LIBENV=""; [ "$v" != "default" ] && LIBENV="TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_$v.so"
  A=$(env $LIBENV python $REPO/tools/wino_res_lab.py 2>&1 | grep 'per-layer launches' | tail -1 | sed -e 's/   resident.*//')
  B=$(env $LIBENV python $REPO/bench.py --lr-size 3x268x640 --scale 2 --degradation BI --no-train-leg --no-secondary --no-parity-check --no-live-pmc --cpu-frames 0 --aten-frames 0 --clips 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2xBI fps', round(d['value'],1), 'chain us', round(d['roofline']['avg_launch_us'],1))")
  C=$(env $LIBENV python $REPO/bench.py --steps 40 --warmup 10 --no-train-leg --no-secondary --no-parity-check --no-live-pmc --cpu-frames 0 --aten-frames 0 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4xBD fps', round(d['value'],1))")
  echo "$v: $A | $B | $C"
done
done
