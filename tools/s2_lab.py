"""Stride-2 data gradient of ConvTranspose2d: the one-shot stride-2 kernel against the phased form on s2d(dZ)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tecogan_pytorch_amd import ops
from tecogan_pytorch_amd.models import train_graph as TG


def bench(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for n, h in ((2, 32), (2, 64), (2, 128)):
    w = torch.randn(64, 64, 3, 3, device='cuda') * 0.05
    dz = torch.randn(n, 64, 2 * h, 2 * h, device='cuda')
    x = torch.randn(n, 64, h, h, device='cuda')
    wk = ops.pack_conv3x3(w, ocb=64)[0]
    we = ops.pack_conv3x3(TG._convt_embed(w))
    s = ops.space_to_depth(dz, 2)
    t_ph = bench(lambda: ops.conv3x3_phased(s, we[0], 256, 64, we[3], 1, 64, ops.TAPS_1, ops.TAPS_01, relu_mask=x))
    if ops.conv3x3s2_supported(n, 64, 64, h, h):
        t_s2 = bench(lambda: ops.conv3x3s2(dz, wk, 64, 64, relu_mask=x))
    else:
        t_s2 = float('nan')
    print(f'n={n} out {h}x{h}: phased {t_ph:.1f} us   stride-2 one-shot {t_s2:.1f} us')
