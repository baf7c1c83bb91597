#!/bin/bash
# Weight gradients of the swept frames on a side stream under the rest of the reverse sweep (TG_WGRAD_SIDE = hand-over points)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for rep in 1 2; do for k in 0 1 2 4; do for c in 128 256; do
  echo "TG_WGRAD_SIDE=$k crop $c: $(TG_WGRAD_SIDE=$k timeout 200 python tools/bench_train.py --crop $c --steps 20 --force-d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d.get('ms_per_step', 0), 3), 'ms')")"
done; done; done
