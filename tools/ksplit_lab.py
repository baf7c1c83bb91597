"""Lab: conv3x3 time vs split-K factor for small problems (training SRNet shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tecogan_pytorch_amd import ops
shapes = [(2, 64, 64, 64, 64), (2, 51, 64, 64, 64), (2, 64, 64, 32, 32), (1, 64, 64, 64, 64),
          (1, 128, 128, 33, 80), (1, 256, 256, 16, 40), (1, 64, 128, 33, 80), (4, 64, 64, 64, 64)]
for n, ci, co, h, w in shapes:
    x = torch.randn(n, ci, h, w, device='cuda')
    wt = torch.randn(co, ci, 3, 3, device='cuda') * 0.05
    b = torch.zeros(co, device='cuda')
    pk, _, _, ocb = ops.pack_conv3x3(wt)
    line = f'n={n} {ci}->{co} @{h}x{w}:'
    for ks in (1, 2, 4, 8):
        if ks > 1 and (ci + 7) // 8 // ks < 1:
            continue
        try:
            for _ in range(5):
                ops.conv3x3(x, pk, b, ci, co, ocb, ops.ACT_RELU, ksplit=ks)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(200):
                ops.conv3x3(x, pk, b, ci, co, ocb, ops.ACT_RELU, ksplit=ks)
            e1.record(); torch.cuda.synchronize()
            line += f'  ks{ks} {1e3 * e0.elapsed_time(e1) / 200:6.1f}us'
        except Exception as e:
            line += f'  ks{ks} err'
    from tecogan_pytorch_amd import _lib as L
    line += f'   heuristic={L.lib().tg_conv3x3_pick_ksplit(n, ci, co, h, w)}'
    print(line)
