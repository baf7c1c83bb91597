# Lab: the first flow pass of a clip (TG_FNET_FIRST_BATCH) at the driver's bench flags and at the defaults
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for fb in ${FBS:-8 4 2}; do
  for st in "20 5" "60 10"; do
    set -- $st
    TG_FNET_FIRST_BATCH=$fb python $REPO/bench.py --gpus 1 --steps $1 --warmup $2 --no-train-leg --no-secondary --no-roofline --cpu-frames 0 --aten-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('first batch $fb  steps $1:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],4), d.get('parity_check'))"
  done
done
