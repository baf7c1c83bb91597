import os
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "1")   # must precede the first GPU call
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
mode = sys.argv[1]
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29533')
os.environ.setdefault('RANK','0'); os.environ.setdefault('WORLD_SIZE','1')
torch.cuda.set_device(0)
import torch.distributed as dist
if mode in ('nccl_init', 'nccl_barrier', 'nccl_devid'):
    if mode == 'nccl_devid':
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('nccl')
if mode == 'gloo':
    dist.init_process_group('gloo')
from tecogan_pytorch_amd.models.networks import FRNet
torch.manual_seed(0)
net = FRNet(3,3,64,10,'BD',4).cuda().eval()
clip = torch.rand(60,3,134,320).cuda()
with torch.no_grad():
    net.infer_sequence(clip[:5], 'cuda', return_device_tensor=True)
    torch.cuda.synchronize()
    if mode in ('nccl_barrier', 'nccl_devid', 'gloo'):
        dist.barrier()
        torch.cuda.synchronize()
    for rep in range(int(os.environ.get("ITERS", "2"))):
        t0 = time.perf_counter()
        net.infer_sequence(clip, 'cuda', return_device_tensor=True)
        torch.cuda.synchronize()
        print(mode, 'rep', rep, round(60/(time.perf_counter()-t0),1), 'fps', flush=True)
