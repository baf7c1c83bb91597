# Lab: the resident launch with / without the transposed-conv tail (TG_WINO_RES_CT), bench kernel table
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for ct in 0 1 0 1; do
TG_WINO_RES_CT=$ct python $REPO/bench.py --steps 60 --warmup 10 --no-train-leg --no-secondary --cpu-frames 0 --aten-frames 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={r['kernel']:r for r in d['kernels']}
print('CT=$ct', round(d['value'],1), 'fps; resident', round(k['conv3x3_wino_resident_kernel']['ms_per_frame']*1e3,1), 'us; convT', round(k.get('convt3x3s2_mfma_kernel',{}).get('ms_per_frame',0)*1e3,1), 'us x', k.get('convt3x3s2_mfma_kernel',{}).get('launches'), '; sum of kernels', round(d['gpu_ms_per_frame_sum_of_kernels']*1e3,1), d['parity_check'])"
done
