"""conv_out (64 -> 3) forward on the training frames: small-cout kernel (4-row form) against the MFMA kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tecogan_pytorch_amd import ops


def bench(fn, it=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for n, h, cin, cout in ((2, 128, 64, 3), (2, 256, 64, 3), (36, 32, 32, 2), (36, 64, 32, 2)):
    x = torch.randn(n, cin, h, h, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    res = torch.randn(n, cout, h, h, device='cuda')
    pk = ops.pack_conv3x3(w)
    t_small = bench(lambda: ops.conv3x3_small(x, w, b, 0))
    t_res = bench(lambda: ops.conv3x3_small(x, w, b, 0, res=res))
    t_mfma = bench(lambda: ops.conv3x3(x, pk[0], b, cin, cout, pk[3], 0, res=res, ksplit=1))
    print(f'n={n} {h}x{h} {cin}->{cout}: small {t_small:.1f} us  small+res {t_res:.1f} us  mfma+res {t_mfma:.1f} us')

print('data gradient of the head (cout_head -> cin_head channels), with the ReLU mask:')
for n, h, cin, cout in ((2, 128, 64, 3), (2, 256, 64, 3), (36, 64, 32, 2)):
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    dz = torch.randn(n, cout, h, h, device='cuda')
    x = torch.relu(torch.randn(n, cin, h, h, device='cuda'))
    wd = w.transpose(0, 1).flip(2, 3).contiguous()
    pkd = ops.pack_conv3x3_dgrad(w)
    t_few = bench(lambda: ops.conv3x3_fewin(dz, wd, relu_mask=x))
    t_mfma = bench(lambda: ops.conv3x3(dz, pkd[0], None, cout, cin, pkd[3], ksplit=1, relu_mask=x))
    print(f'n={n} {h}x{h} {cout}->{cin}: fewin {t_few:.1f} us  mfma {t_mfma:.1f} us')
