"""Lab: time of the critic's Conv2d(4, 2, 1) blocks (forward and data gradient through their space-to-depth embedding)
at the training shapes.  python tools/d_conv_lab.py [crop]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_pytorch_amd import ops
from tecogan_pytorch_amd.models import train_graph as TG

crop = int(sys.argv[1]) if len(sys.argv) > 1 else 128


class H:
    pass


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for clips in (24, 12):
    tot = 0.0
    for name, ci, co, hw in (('block1', 64, 64, crop), ('block2', 64, 64, crop // 2), ('block3', 64, 128, crop // 4),
                             ('block4', 128, 256, crop // 8)):
        h = H(); h.weight = torch.randn(co, ci, 4, 4, device='cuda') * 0.05
        h.weight.requires_grad_(False)
        x = torch.randn(clips, ci, hw, hw, device='cuda')
        y = TG.conv4x4s2(None, h, x)
        g = torch.randn_like(y)
        s = ops.space_to_depth(x, 2)
        pk = ops.pack_conv3x3(TG._conv4_embed(h.weight))
        pkd = ops.pack_conv3x3_dgrad(TG._conv4_embed(h.weight))
        t_s2d = timeit(lambda: ops.space_to_depth(x, 2))
        t_f = timeit(lambda: ops.conv3x3_phased(s, pk[0], 4 * ci, co, pk[3], 1, ci, ops.TAPS_12, ops.TAPS_01))
        if ci % 64 == 0:
            t_d = timeit(lambda: ops.conv3x3_phased(g, pkd[0], co, 4 * ci, pkd[3], 2, ci, ops.TAPS_01, ops.TAPS_12))
        else:
            t_d = timeit(lambda: ops.conv3x3(g, pkd[0], None, co, 4 * ci, pkd[3], ksplit=1))
        ds = ops.conv3x3_phased(g, pkd[0], co, 4 * ci, pkd[3], 2, ci, ops.TAPS_01, ops.TAPS_12)
        t_d2s = timeit(lambda: ops.depth_to_space(ds, 2))
        gf = 2.0 * clips * (hw // 2) ** 2 * 16 * ci * co / 1e9
        print(f'{clips:2d} clips {name} {ci:3d}->{co:3d} in {hw:3d}^2: s2d {t_s2d:6.1f}  fwd {t_f:6.1f} us ({gf / t_f * 1e3:5.1f} TF/s)  '
              f'dgrad {t_d:6.1f} us ({gf / t_d * 1e3:5.1f} TF/s)  d2s {t_d2s:6.1f}   [{gf:.2f} GFLOP]')
        tot += t_s2d + t_f + t_d + t_d2s
    print(f'{clips} clips: sum {tot:.0f} us')

print('direct kernels (tg_conv4x4s2_fwd / _dgrad):')
for clips in (24, 12):
    for name, ci, co, hw in (('block1', 64, 64, crop), ('block2', 64, 64, crop // 2), ('block3', 64, 128, crop // 4),
                             ('block4', 128, 256, crop // 8)):
        if not ops.conv4x4s2_supported(clips, ci, co, hw, hw):
            continue
        wt = torch.randn(co, ci, 4, 4, device='cuda') * 0.05
        pf, pd = ops.pack_conv4x4s2(wt)
        x = torch.randn(clips, ci, hw, hw, device='cuda')
        y = ops.conv4x4s2(x, pf, co)
        g = torch.randn_like(y)
        t_f = timeit(lambda: ops.conv4x4s2(x, pf, co, out=y))
        t_d = timeit(lambda: ops.conv4x4s2_dgrad(g, pd, ci))
        t_da = timeit(lambda: ops.conv4x4s2_dgrad(g, pd, ci, act_y=x, act=ops.ACT_LRELU02))
        gf = 2.0 * clips * (hw // 2) ** 2 * 16 * ci * co / 1e9
        print(f'{clips:2d} clips {name} {ci:3d}->{co:3d} in {hw:3d}^2: fwd {t_f:6.1f} us ({gf / t_f * 1e3:5.1f} TF/s)  '
              f'dgrad {t_d:6.1f} us ({gf / t_d * 1e3:5.1f} TF/s)  dgrad+act {t_da:6.1f} us')
