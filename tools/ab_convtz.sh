#!/bin/bash
# A/B of the Z-mode transposed conv's forms through the whole frame (lab build: TG_CONVTZ_FORM = 0 tiled, 1 streaming
# with the dynamic item list, 2 streaming with the static list).   bash tools/ab_convtz.sh   (on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TG_LAB_BUILD=1 OUT=$REPO/tools/_lab_libs/ztest bash tecogan-pytorch_amd/csrc/build.sh > /dev/null 2>&1 || { echo "lab build failed"; exit 1; }
export TECOGAN_HIP_LIB=$REPO/tools/_lab_libs/ztest/libtecogan_lab.so
for rep in 1 2; do
for f in 0 1 2; do
  TG_CONVTZ_FORM=$f timeout 200 python bench.py --steps 40 --warmup 5 --clips 7 --no-train-leg --cpu-frames 0 --aten-frames 0 --no-live-pmc --no-parity-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
z = [r for r in d['kernels'] if 'convt3x3s2' in r['kernel']]
print('form $f: value %.1f fps  single_stream %.1f  4clips %.1f  Z us %s' % (d['value'], d.get('fps_clip_single_stream', 0), d.get('fps_4_clips_pipelined', 0), [round(1e3 * r['ms_per_frame'], 1) for r in z]))"
done; done
