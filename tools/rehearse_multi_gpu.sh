# Rehearsal of `bench.py --gpus N` on a box with ONE GPU: N ranks share cuda:0 and exchange through
# gloo (TG_BENCH_REHEARSAL=1).  Walks every N > 1 code path of bench.py (clip sharding, MAX-over-ranks
# timing, the DDP training leg with SyncBN + both gradient buckets + the adaptive-D scalar exchange,
# the all-reduce micro-benchmark) except RCCL itself; the numbers mean nothing (shared GPU).
N=${1:-2}
export TG_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus $N --steps 10 --warmup 3 --clips 3 --train-steps 3 --cpu-seconds 2 2>&1 | tail -3
