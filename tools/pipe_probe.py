import os
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", os.environ.get("DYN", "1"))
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tecogan_pytorch_amd.models.networks import define_generator
opt = {'scale': 4, 'dataset': {'degradation': {'type': 'BD'}},
       'model': {'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10}}}
torch.manual_seed(0)
net = define_generator(opt).cuda().eval()
NF = int(os.environ.get('NF', '60'))
PIPE = os.environ.get('PIPE', '1') == '1'
clip = torch.rand(NF, 3, 134, 320, device='cuda')
mode = os.environ.get('MAIN_HI', '0')
def run():
    if mode == '1':
        hi = torch.cuda.Stream(priority=int(os.environ.get('MAIN_PRIO', '-1'))) if not hasattr(run, 'hi') else run.hi
        run.hi = hi
        hi.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(hi):
            out = net.infer_sequence(clip, 'cuda', pipeline=PIPE, return_device_tensor=True)
        torch.cuda.current_stream().wait_stream(hi)
        return out
    return net.infer_sequence(clip, 'cuda', pipeline=PIPE, return_device_tensor=True)
for _ in range(2): run()
torch.cuda.synchronize()
res = []
for _ in range(int(os.environ.get("ITERS", "5"))):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); res.append(NF / (time.perf_counter() - t0))
print('NF', NF, 'pipe', PIPE, 'side_prio', os.environ.get('TG_SIDE_STREAM_PRIORITY', '-1'), 'main_hi', mode, ' '.join(f'{r:.0f}' for r in res))
