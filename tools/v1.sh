#!/bin/bash
# round-4 visit 1: resident SRNet body -- parity, lab timing, A/B bench
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "resident or winograd_chain_numerics or rule_and_plan" 2>&1 | tail -25
echo "== lab"; timeout 300 python tools/wino_res_lab.py 2>&1 | tail -8
echo "== bench resident"; timeout 400 python bench.py --steps 40 --warmup 10 --no-train-leg --cpu-frames 0 --aten-frames 0 > $OUT/v1_bench_res.json 2> $OUT/v1_bench_res.err; tail -3 $OUT/v1_bench_res.err; cut -c1-400 $OUT/v1_bench_res.json
echo "== bench per-layer"; TG_WINO_RES=0 timeout 400 python bench.py --steps 40 --warmup 10 --no-train-leg --cpu-frames 0 --aten-frames 0 --no-roofline > $OUT/v1_bench_nores.json 2> $OUT/v1_bench_nores.err; tail -3 $OUT/v1_bench_nores.err; cut -c1-400 $OUT/v1_bench_nores.json
