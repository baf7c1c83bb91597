# kernel trace of the training step: how much of the kernel time overlaps (two streams)?  bash tools/prof_overlap.sh 128
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CROP=${1:-128}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_ov -o kt -- python $REPO/tools/bench_train.py --crop $CROP --steps 4 --warmup 2 --force-d > $REPO/gpurun_out/prof_ov.log 2>&1
python - <<PY
import csv, glob, collections
tr = glob.glob('$REPO/gpurun_out/prof_ov/**/kt_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', ''), r.get('Stream_Id', ''), r['Kernel_Name'][:50]) for r in rows)
ev = ev[len(ev) // 2:]                      # the last steps
tot = sum(e - s for s, e, *_ in ev)
union, cur_s, cur_e = 0, None, None
for s, e, *_ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = ev[-1][1] - ev[0][0]
print('kernels', len(ev), 'sum ms %.3f  union ms %.3f  span ms %.3f' % (tot / 1e6, union / 1e6, span / 1e6))
print('queues', collections.Counter(q for _, _, q, _, _ in ev).most_common(6))
print('streams', collections.Counter(st for _, _, _, st, _ in ev).most_common(6))
side = [i for i, e in enumerate(ev) if e[3] != '0']
if side:
    # the last burst of side-stream kernels and the main-stream kernels around it
    last = side[-1]
    first = last
    while first - 1 in side or (first - 1 >= 0 and any(j in side for j in range(max(0, first - 6), first))):
        first -= 1
        if first not in side and not any(j in side for j in range(max(0, first - 6), first)): break
    t0 = ev[max(0, first - 4)][0]
    for s_, e_, q, st, nm in ev[max(0, first - 4):min(len(ev), last + 5)]:
        print('%9.1f us  +%7.1f us  stream %s  %s' % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, st, nm))
PY
rm -rf $REPO/gpurun_out/prof_ov
