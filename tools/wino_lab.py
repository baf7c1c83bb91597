"""Lab: Winograd F(2x2,3x3) conv kernel vs the direct MFMA kernel -- correctness against torch's
CPU fp32 conv2d (and an fp64 truth) and launch time (HIP events, 200 launches).
  python tools/wino_lab.py [h w cin cout]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tecogan_pytorch_amd import ops


def check(n, cin, cout, h, w, act=ops.ACT_RELU, dual=False, res=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g)
    r = torch.randn(n, cout, h, w, generator=g) if res else None
    ref64 = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    ref32 = F.conv2d(x, wt, b, padding=1)
    if act == ops.ACT_RELU:
        ref64, ref32 = ref64.relu(), ref32.relu()
    if res:
        ref64, ref32 = ref64 + r.double(), ref32 + r
    xd, wd, bd = x.cuda(), wt.cuda(), b.cuda()
    u = ops.pack_conv3x3_wino(wd)
    if dual:
        c1 = 3
        y = ops.conv3x3_wino(xd[:, :c1].contiguous(), u, bd, cin, cout, act, x2=xd[:, c1:].contiguous(),
                             res=r.cuda() if res else None)
    else:
        y = ops.conv3x3_wino(xd, u, bd, cin, cout, act, res=r.cuda() if res else None)
    pk = ops.pack_conv3x3(wd)
    yd = ops.conv3x3(xd, pk[0], bd, cin, cout, pk[3], act, res=r.cuda() if res else None, ksplit=1)
    torch.cuda.synchronize()
    e_w = (y.cpu().double() - ref64).abs().max().item()
    e_d = (yd.cpu().double() - ref64).abs().max().item()
    e_c = (ref32.double() - ref64).abs().max().item()
    print(f'n={n} cin={cin} cout={cout} {h}x{w} dual={dual} res={res}: max|err| vs fp64  wino {e_w:.3e}  '
          f'direct-hip {e_d:.3e}  torch-cpu-fp32 {e_c:.3e}', flush=True)
    return e_w


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bench(n, cin, cout, h, w):
    x = [torch.randn(n, cin, h, w, device='cuda') for _ in range(2)]
    y = [torch.empty(n, cout, h, w, device='cuda') for _ in range(2)]
    wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    u = ops.pack_conv3x3_wino(wt)
    pk = ops.pack_conv3x3(wt)
    st = {'i': 0}

    def f_w():
        i = st['i'] = st['i'] ^ 1
        ops.conv3x3_wino(x[i], u, b, cin, cout, ops.ACT_RELU, res=x[i] if cin == cout else None, out=y[i])

    def f_d():
        i = st['i'] = st['i'] ^ 1
        ops.conv3x3(x[i], pk[0], b, cin, cout, pk[3], ops.ACT_RELU, res=x[i] if cin == cout else None, out=y[i], ksplit=1)
    tw, td = timeit(f_w), timeit(f_d)
    gf = 2.0 * n * cin * cout * 9 * h * w / 1e9
    print(f'TIME n={n} cin={cin} cout={cout} {h}x{w}: wino {tw:.1f} us ({gf / tw * 1e3:.1f} TFLOP/s algorithmic)   '
          f'direct {td:.1f} us ({gf / td * 1e3:.1f})', flush=True)


if __name__ == '__main__':
    if len(sys.argv) == 5:
        h, w, ci, co = [int(v) for v in sys.argv[1:]]
        check(1, ci, co, h, w); bench(1, ci, co, h, w)
        sys.exit(0)
    check(1, 16, 16, 8, 8)
    check(1, 64, 64, 20, 36, res=True)
    check(2, 51, 64, 21, 37, dual=True)
    check(1, 32, 128, 33, 80, act=ops.ACT_NONE)
    check(1, 64, 64, 134, 320, res=True)
    bench(1, 64, 64, 134, 320)
    bench(4, 64, 64, 134, 320)
    bench(1, 128, 128, 33, 80)
    bench(2, 64, 64, 64, 64)
    bench(1, 64, 64, 268, 640)
