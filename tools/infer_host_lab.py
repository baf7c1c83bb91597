"""Lab: host time of FRNet.infer_sequence's set-up (before the first frame is enqueued) and of the whole enqueue, per clip."""
import os, sys, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_pytorch_amd.models.networks import FRNet
torch.manual_seed(0)
net = FRNet(3, 3, 64, 10, 'BD', 4).cuda().eval()
clip = torch.rand(20, 3, 134, 320).cuda()
for _ in range(3):
    net.infer_sequence(clip, 'cuda', return_device_tensor=True)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.infer_sequence(clip, 'cuda', return_device_tensor=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print('enqueue ms, total ms per 20-frame clip:', [tuple(round(v, 3) for v in t) for t in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    net.infer_sequence(clip, 'cuda', return_device_tensor=True)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
