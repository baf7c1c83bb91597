"""Lab: one FNet-only pass per batch size under rocprofv3 --kernel-trace (tools/fnet_layers.sh prints the launches in order)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_pytorch_amd import _lib as L
from tecogan_pytorch_amd.models.networks import define_generator
opt = {'scale': 4, 'dataset': {'degradation': {'type': 'BD'}},
       'model': {'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10}}}
torch.manual_seed(0)
net = define_generator(opt).cuda().eval()
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lr = torch.rand(nb + 1, 3, 134, 320, device='cuda')
plan = net._get_plan(nb, 134, 320, torch.device('cuda'), fnet_only=True)
for _ in range(6):
    L.check(lib.tg_frnet_step_phase(plan.handle, 1, 0, lr[1:].data_ptr(), lr[:-1].data_ptr(), None, None,
                                    None, st), 'phase1')
torch.cuda.synchronize()
