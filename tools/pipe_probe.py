"""Per-clip rates of the pipelined clip inference.  DYN=1 sets DEBUG_HIP_DYNAMIC_QUEUES=1 before HIP
starts (the round-2 setting; since round 3 the side stream has its own hardware queue and the
variable is not needed); INIT_FIRST=1 initialises HIP and creates 3 extra streams before the
package is imported (the situation of a host application with its own GPU work); RCCL=1 also
creates a 1-rank NCCL process group first."""
import os
if os.environ.get("DYN", "0") == "1":
    os.environ["DEBUG_HIP_DYNAMIC_QUEUES"] = "1"
import sys, os, time, torch
if os.environ.get("INIT_FIRST", "0") == "1":
    torch.cuda.init(); _x = torch.zeros(4, device="cuda"); _extra = [torch.cuda.Stream() for _ in range(3)]
    for _s in _extra:
        with torch.cuda.stream(_s): _x += 1
    torch.cuda.synchronize()
if os.environ.get("RCCL", "0") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    _t = torch.ones(8, device="cuda"); dist.all_reduce(_t); torch.cuda.synchronize()
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
print('NF', NF, 'pipe', PIPE, 'DYN', os.environ.get('DEBUG_HIP_DYNAMIC_QUEUES', '-'), 'init_first', os.environ.get('INIT_FIRST', '0'), 'rccl', os.environ.get('RCCL', '0'), 'main_hi', mode, ' '.join(f'{r:.0f}' for r in res))
