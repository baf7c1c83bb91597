"""Lab: time the fused flow-upsample + warp + space_to_depth kernel alone, per kernel form.
Rotates over enough buffer sets that no launch finds its inputs in L2/MALL.
  python tools/warp_lab.py --camera --lr 134 320 --clips 1 8"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tecogan_pytorch_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--clips', type=int, nargs='+', default=[1, 8])
ap.add_argument('--deg', default='BD')
ap.add_argument('--scale', type=int, default=4)
ap.add_argument('--lr', type=int, nargs=2, default=[134, 320])
ap.add_argument('--reps', type=int, default=48)
ap.add_argument('--amp', type=float, default=0.6, help='LR-pixel std of the synthetic flow')
ap.add_argument('--cell', type=int, default=8, help='LR pixels per random flow cell')
ap.add_argument('--camera', action='store_true', help='pan + 1%% zoom + roll instead of the random field')
args = ap.parse_args()
dev = torch.device('cuda:0')
h, w = args.lr
s = args.scale
mode = ops.UP_MODE[args.deg]
g = torch.Generator().manual_seed(3)
for clips in args.clips:
    per_set = clips * (2 * 3 * s * s * h * w + 2 * h * w) * 4
    nsets = max(2, int(600e6 // per_set) + 1)
    sets = []
    for _ in range(nsets):
        fl = torch.randn(clips, 2, h // args.cell + 1, w // args.cell + 1, generator=g) * args.amp
        fl = torch.nn.functional.interpolate(fl, size=(h, w), mode='bilinear')
        if args.camera:
            ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) - h / 2,
                                    torch.arange(w, dtype=torch.float32) - w / 2, indexing='ij')
            pan = (torch.rand(clips, 2, 1, 1, generator=g) - 0.5) * 2.0
            zoom = 0.01 * (torch.rand(clips, 1, 1, 1, generator=g) - 0.5) * 2
            roll = 0.005 * (torch.rand(clips, 1, 1, 1, generator=g) - 0.5) * 2
            fl = torch.cat([pan[:, 0:1] + zoom * xs - roll * ys, pan[:, 1:2] + zoom * ys + roll * xs], 1)
        fl = fl.to(dev).contiguous()
        pv = torch.rand(clips, 3, s * h, s * w, generator=g).to(dev).contiguous()
        sets.append((fl, pv, torch.empty(clips, s * s * 3, h, w, device=dev)))
    if True:
        for i in range(nsets):
            ops.flowup_warp_s2d(sets[i][0], sets[i][1], h, w, s, mode, out=sets[i][2])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.reps):
            fl, pv, out = sets[i % nsets]
            ops.flowup_warp_s2d(fl, pv, h, w, s, mode, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.reps
        print(f"{'camera' if args.camera else 'random'} amp={args.amp} cell={args.cell} "
              f"clips={clips} sets={nsets} {us:8.2f} us/launch  {per_set / us / 1e3:8.1f} GB/s  "
              f"frac={per_set / us / 1e3 / 8000:.3f}", flush=True)
