#!/bin/bash
# PMC passes over the fused warp kernel alone (8 clips per launch).  Usage: bash tools/warp_pmc.sh tag
TAG=${1:-warp}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/tools/warp_lab.py --clips 8 --reps 8"
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/wpmc_${TAG}_$i -o pmc -- $CMD > $OUT/wpmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$? : $PMC"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for path in sorted(glob.glob("$OUT/wpmc_${TAG}_*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if 'flowup_warp' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
with open("$OUT/wpmc_${TAG}_summary.csv", 'w') as f:
    f.write('counter,mean_per_launch,launches\n')
    for k in sorted(agg):
        f.write(f'{k},{sum(agg[k]) / len(agg[k]):.1f},{len(agg[k])}\n')
print(open("$OUT/wpmc_${TAG}_summary.csv").read())
PY
