#!/bin/bash
# Last visit of a round: the whole GPU suite, smoke, the bench line at the driver's flags, the self-spawned rehearsals.
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench at the driver's flags"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_flags.json 2>/dev/null; cut -c1-330 $OUT/${TAG}_bench_driver_flags.json
for n in 2 4; do
  echo "== python bench.py --gpus $n (TG_BENCH_REHEARSAL=1: $n ranks on this ONE GPU over gloo; the numbers mean nothing)"
  TG_BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus $n --steps 10 --warmup 3 --clips 3 --train-steps 3 --cpu-frames 0 --no-roofline --no-secondary > $OUT/${TAG}_rehearsal_${n}ranks_one_gpu_selfspawn.json 2> $OUT/${TAG}_rehearsal_${n}ranks.err
  echo "rc=$?"; python -c "
import json,sys
d=json.load(open('$OUT/${TAG}_rehearsal_${n}ranks_one_gpu_selfspawn.json'))
t=d['train_ddp']
print({k:d[k] for k in ('n_gpus','ranks_seen','distinct_gpus','launched_by')}, 'train n_gpus', t.get('n_gpus'), 'G/D bytes', t.get('allreduce_G',{}).get('bytes'), t.get('allreduce_D',{}).get('bytes'), 'comm/step', t.get('rccl_comm_count',{}).get('all_reduce_per_step'), t.get('rccl_comm_count',{}).get('all_gather_per_step'))"
done
echo "== python bench.py --gpus 8 on this box (must refuse)"; python bench.py --gpus 8 --steps 2 --warmup 1; echo "rc=$?"
