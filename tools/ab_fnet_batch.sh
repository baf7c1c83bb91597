for rep in 1 2; do for b in 6 8 10 12 16; do
  for k in 20 60; do
  echo "TG_FNET_BATCH=$b steps $k: $(TG_FNET_BATCH=$b timeout 120 python bench.py --steps $k --warmup 5 --no-train-leg --cpu-frames 0 --aten-frames 0 --no-live-pmc --no-parity-check --no-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))")"
  done
done; done
