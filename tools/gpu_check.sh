#!/bin/bash
# One GPU-box visit: smoke, parity tests, bench, rocprof kernel stats.
# Usage (from the repo root on the GPU box):  bash tools/gpu_check.sh [tag]
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
echo "== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40
echo "== bench"; timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -3 $OUT/bench_$TAG.err; cat $OUT/bench_$TAG.json
echo "== rocprof"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o kt -- python $REPO/bench.py --steps 30 --warmup 5 --no-roofline --cpu-frames 0 --aten-frames 0 > $OUT/prof_$TAG.log 2>&1
cd $REPO
ls -R $OUT/prof_$TAG | head -20
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp "$f" $OUT/kernel_stats_$TAG.csv; head -25 "$f"; }
# same, single stream (per-kernel durations not inflated by the FNet/SRNet overlap)
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1s_$TAG -o kt -- python $REPO/bench.py --steps 30 --warmup 5 --no-roofline --no-pipeline --cpu-frames 0 --aten-frames 0 > $OUT/prof1s_$TAG.log 2>&1
cd $REPO
f=$(find $OUT/prof1s_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp "$f" $OUT/kernel_stats_1stream_$TAG.csv; head -8 "$f" | cut -c1-150; }
