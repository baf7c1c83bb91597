#!/bin/bash
# One GPU-box visit that produces everything committed under profiles/ for a round:
#   bash tools/gpu_round.sh r02      (from the repo root on the GPU box; writes gpurun_out/)
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
echo "== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | sed -n 3,8p
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
# PMC passes FIRST: they stamp profiles/pmc_traffic.json with the current sources, so that the bench line below
# carries roofline.traffic from THIS build (traffic_stale false)
echo "== PMC passes"; bash tools/gpu_pmc.sh $TAG 2>&1 | grep "^pass"
# summarise on the box (the raw per-dispatch tables are too large to travel back)
python tools/summarize_pmc.py $TAG $TAG $OUT && rm -rf $OUT/pmc_${TAG}_*
cp $OUT/pmc_traffic.json $REPO/profiles/pmc_traffic.json 2>/dev/null
echo "== bench (default flags)"; timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -2 $OUT/${TAG}_bench.err; cut -c1-600 $OUT/${TAG}_bench.json
echo "== bench 2xBI (configs[4]) as its own run"; timeout 600 python bench.py --lr-size 3x268x640 --scale 2 --degradation BI --no-train-leg --cpu-frames 0 --aten-frames 0 --clips 5 > $OUT/${TAG}_bench_config5_2xBI.json 2>/dev/null; cut -c1-300 $OUT/${TAG}_bench_config5_2xBI.json
echo "== rocprofv3 kernel stats: pipelined / single stream / training"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o kt -- python $REPO/bench.py --steps 30 --warmup 5 --clips 3 --no-roofline --no-secondary --no-parity-check --no-train-leg --cpu-frames 0 --aten-frames 0 > $OUT/prof_$TAG.log 2>&1
cp $OUT/prof_$TAG/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats_rocprofv3.csv; rm -rf $OUT/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1s_$TAG -o kt -- python $REPO/bench.py --steps 30 --warmup 5 --clips 3 --no-roofline --no-pipeline --no-secondary --no-parity-check --no-train-leg --cpu-frames 0 --aten-frames 0 > $OUT/prof1s_$TAG.log 2>&1
cp $OUT/prof1s_$TAG/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats_single_stream_rocprofv3.csv; rm -rf $OUT/prof1s_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proft_$TAG -o kt -- python $REPO/tools/bench_train.py --crop 256 --steps 4 --warmup 2 --force-d > $OUT/proft_$TAG.log 2>&1
cp $OUT/proft_$TAG/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats_train_rocprofv3.csv; rm -rf $OUT/proft_$TAG
cd $REPO
head -8 $OUT/${TAG}_kernel_stats_single_stream_rocprofv3.csv | cut -c1-160
echo "== training steps"; for c in 256 128; do timeout 300 python tools/bench_train.py --crop $c --steps 10 --force-d 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_train.jsonl; done
timeout 300 python tools/bench_train.py --crop 256 --steps 10 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_train.jsonl
timeout 300 python tools/bench_train.py --crop 256 --steps 10 --model FRVSR 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_train.jsonl
for c in 128 256; do timeout 300 python tools/bench_train.py --crop $c --steps 6 --force-d --feature-crit 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_train.jsonl; done
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proft128_$TAG -o kt -- python $REPO/tools/bench_train.py --crop 128 --steps 4 --warmup 2 --force-d > $OUT/proft128_$TAG.log 2>&1
cp $OUT/proft128_$TAG/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats_train_crop128_rocprofv3.csv; rm -rf $OUT/proft128_$TAG; cd $REPO
cut -c1-220 $OUT/${TAG}_bench_train.jsonl
ls -la $OUT | head -30
