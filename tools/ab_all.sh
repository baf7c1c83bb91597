# A/B of whole-library variants on ONE box: the headline bench line, the kernel table's main rows, the 2xBI clip and the
# training step at crop 128 / 256.  bash tools/ab_all.sh default NAME ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "$@"; do
  LIBENV=""; [ "$v" != "default" ] && LIBENV="TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/libtecogan_$v.so"
  A=$(env $LIBENV python $REPO/bench.py --steps 40 --warmup 10 --no-train-leg --no-secondary --no-parity-check --no-live-pmc --cpu-frames 0 --aten-frames 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={r['kernel'].split('<')[0]+('<Z>' if 'Z' in r['kernel'] else ''): round(1e3*r['ms_per_frame'],1) for r in d['kernels']}
print('fps', round(d['value'],1), 'res', k.get('conv3x3_wino_resident_kernel'), 'Z', k.get('convt3x3s2_mfma_kernel<Z>'), 'tail', k.get('convout_tail_kernel'), 'warp', k.get('flowup_warp_s2d_kernel'))")
  B=$(env $LIBENV python $REPO/bench.py --lr-size 3x268x640 --scale 2 --degradation BI --no-train-leg --no-secondary --no-parity-check --cpu-frames 0 --aten-frames 0 --clips 3 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2xBI', round(d['value'],1))")
  C=$(env $LIBENV python $REPO/tools/bench_train.py --crop 128 --steps 10 --force-d 2>/dev/null | tail -1 | python -c "import sys,json; print('t128', round(json.loads(sys.stdin.read())['ms_per_step'],2))")
  D=$(env $LIBENV python $REPO/tools/bench_train.py --crop 256 --steps 10 --force-d 2>/dev/null | tail -1 | python -c "import sys,json; print('t256', round(json.loads(sys.stdin.read())['ms_per_step'],2))")
  echo "$v: $A | $B | $C | $D"
done
done
