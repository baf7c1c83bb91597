#!/bin/bash
# PMC passes (separate from timing runs).  Usage: bash tools/gpu_pmc.sh tag
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 6 --warmup 2 --clips 1 --no-roofline --no-pipeline --no-secondary --no-parity-check --no-train-leg --cpu-frames 0 --aten-frames 0"
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$i -o pmc -- $CMD > $OUT/pmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$? : $PMC"; ls $OUT/pmc_${TAG}_$i | head
done
# BASELINE configs[4] (2x BI, 268x640): its SRNet runs as the chained Winograd launch -- traffic passes only
CMD5="python $REPO/bench.py --lr-size 3x268x640 --scale 2 --degradation BI --steps 6 --warmup 2 --clips 1 --no-roofline --no-pipeline --no-secondary --no-parity-check --no-train-leg --cpu-frames 0 --aten-frames 0"
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$i -o pmc -- $CMD5 > $OUT/pmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$? : $PMC (2xBI)"
done
# the training step (crop 128): traffic of the chained body launch and the layered weight-gradient launch
CMDT="python $REPO/tools/bench_train.py --crop 128 --steps 3 --warmup 1 --force-d"
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$i -o pmc -- $CMDT > $OUT/pmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$? : $PMC (train crop 128)"
done
