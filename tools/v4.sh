#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
echo "== bench resident"; timeout 400 python bench.py --steps 40 --warmup 10 --no-train-leg --cpu-frames 0 --aten-frames 0 > $OUT/v4_bench_res.json 2> $OUT/v4_bench_res.err; tail -3 $OUT/v4_bench_res.err; cut -c1-300 $OUT/v4_bench_res.json
