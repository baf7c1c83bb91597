"""Lab: GPU time of the batched flow estimator (FNet-only plan) per frame pair."""
import os, sys, time, torch
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_pytorch_amd import _lib as L
from tecogan_pytorch_amd.models.networks import define_generator
opt = {'scale': 4, 'dataset': {'degradation': {'type': 'BD'}},
       'model': {'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10}}}
torch.manual_seed(0)
net = define_generator(opt).cuda().eval()
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for nb in (1, 4, 8, 16):
    lr = torch.rand(nb + 1, 3, 134, 320, device='cuda')
    plan = net._get_plan(nb, 134, 320, torch.device('cuda'), fnet_only=True)
    def run():
        L.check(lib.tg_frnet_step_phase(plan.handle, 1, 0, lr[1:].data_ptr(), lr[:-1].data_ptr(), None, None,
                                        None, st), 'phase1')
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print(f'FNet batch {nb:2d}: {us:8.1f} us per pass = {us / nb:7.1f} us per frame pair  ({10.511 / (us / nb) * 1e-3 * 1e3:.1f} TFLOP/s)')
