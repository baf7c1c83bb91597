"""Clip-inference rate for k independent clips advanced in lockstep (FRNet.infer_sequence on a
(k, t, c, h, w) batch).   python tools/multi_clip_probe.py [--clips 1 2 4 8] [--frames 60]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tecogan_pytorch_amd  # noqa: F401  (sets DEBUG_HIP_DYNAMIC_QUEUES before HIP starts)
from tecogan_pytorch_amd.models.networks import FRNet

ap = argparse.ArgumentParser()
ap.add_argument('--clips', type=int, nargs='+', default=[1, 2, 4, 8])
ap.add_argument('--frames', type=int, default=60)
ap.add_argument('--lr', type=int, nargs=2, default=[134, 320])
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = FRNet(3, 3, 64, 10, 'BD', 4).to(dev).eval()
h, w = a.lr
for k in a.clips:
    x = torch.rand(k, a.frames, 3, h, w).to(dev)
    x1 = x[0] if k == 1 else x
    for _ in range(2):
        net.infer_sequence(x1, dev, return_device_tensor=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        net.infer_sequence(x1, dev, return_device_tensor=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[2]
    print(f'clips={k} frames/s={k * a.frames / med:8.1f}  ms/frame-step={1e3 * med / a.frames:.3f}', flush=True)
