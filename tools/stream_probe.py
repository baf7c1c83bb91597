"""Which kind of second stream really runs beside the main stream?  Per-clip rates of the pipelined
clip inference with the side stream being (a) the CU-mask stream of tg_stream_create_dedicated,
(b) a torch pool stream, (c) a plain hipStreamCreateWithFlags(nonblocking) stream, (d) a
high-priority torch stream; plus the batched-FNet pass timed alone on each kind of stream."""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
if os.environ.get("RCCL", "0") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    _t = torch.ones(1 << 20, device="cuda"); dist.all_reduce(_t); torch.cuda.synchronize()
if os.environ.get("INIT_FIRST", "0") == "1":
    torch.cuda.init(); _x = torch.zeros(4, device="cuda"); _extra = [torch.cuda.Stream() for _ in range(3)]
    for _s in _extra:
        with torch.cuda.stream(_s): _x += 1
    torch.cuda.synchronize()
from tecogan_pytorch_amd.models.networks import define_generator
from tecogan_pytorch_amd import ops, _lib as L
opt = {'scale': 4, 'dataset': {'degradation': {'type': 'BD'}},
       'model': {'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10}}}
torch.manual_seed(0)
net = define_generator(opt).cuda().eval()
NF = 60
clip = torch.rand(NF, 3, 134, 320, device='cuda')
dev = torch.device('cuda', 0)
hip = ctypes.CDLL('libamdhip64.so')
def plain_stream():
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
    return torch.cuda.ExternalStream(h.value, device=dev)
def cumask_stream():
    h = ctypes.c_void_p()
    mask = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, mask) == 0
    return torch.cuda.ExternalStream(h.value, device=dev)
kinds = {'cumask': cumask_stream, 'torch': lambda: torch.cuda.Stream(),
         'plain': plain_stream, 'torch_hi': lambda: torch.cuda.Stream(priority=-1)}
def fnet_alone(st):
    plan = net._get_plan(8, 134, 320, dev, fnet_only=True)
    x = torch.rand(9, 3, 134, 320, device='cuda')
    torch.cuda.synchronize()
    def go():
        L.check(L.lib().tg_frnet_step_phase(plan.handle, 1, 0, x[1:].data_ptr(), x[:8].data_ptr(), None, None, None, st.cuda_stream), 'p1')
    for _ in range(3): go()
    st.synchronize(); t0 = time.perf_counter()
    for _ in range(10): go()
    st.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3
for name in os.environ.get('KINDS', 'cumask,torch,plain,torch_hi').split(','):
    st = kinds[name]()
    net._side = st
    ms = fnet_alone(st)
    for _ in range(2): net.infer_sequence(clip, dev, return_device_tensor=True)
    torch.cuda.synchronize()
    res = []
    for _ in range(9):
        t0 = time.perf_counter(); net.infer_sequence(clip, dev, return_device_tensor=True); torch.cuda.synchronize()
        res.append(NF / (time.perf_counter() - t0))
    print(f'{name:9s} DYN={os.environ.get("DEBUG_HIP_DYNAMIC_QUEUES", "-")} init_first={os.environ.get("INIT_FIRST", "0")} rccl={os.environ.get("RCCL", "0")}  FNet x8 alone {ms:.3f} ms;  clips/s:', ' '.join(f'{r:.0f}' for r in res), flush=True)

if os.environ.get("RCCL", "0") == "1":
    dist.destroy_process_group()
