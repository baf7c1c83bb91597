cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_hip_train_ops.py -m gpu -q -x -k "stride2" 2>&1 | tail -3
python tools/s2_lab.py
python tools/bench_train.py --crop 256 --steps 6 --warmup 3 --force-d 2>&1 | tail -1 | cut -c1-300
python tools/bench_train.py --crop 128 --steps 6 --warmup 3 --force-d 2>&1 | tail -1 | cut -c1-300
