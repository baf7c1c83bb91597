"""Lab: SRNet's conv_in + 20 residual-block convs at 134x320 -- 21 per-layer Winograd launches against
the ONE LDS-resident launch (tg_conv3x3_wino_resident).  HIP events around `reps` back-to-back bodies;
prints us per body and per layer, and checks the two outputs bit for bit.
  python tools/wino_res_lab.py [h w [nb]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tecogan_pytorch_amd.ops as ops  # noqa: E402


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (134, 320)
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    g = torch.Generator().manual_seed(1)
    dev = 'cuda'
    lr = torch.rand(1, 3, h, w, generator=g).to(dev)
    s2d = torch.rand(1, 48, h, w, generator=g).to(dev)
    ws = [torch.randn(64, 51, 3, 3, generator=g).to(dev) * 0.04] + \
         [torch.randn(64, 64, 3, 3, generator=g).to(dev) * 0.03 for _ in range(2 * nb)]
    bs = [torch.randn(64, generator=g).to(dev) * 0.1 for _ in range(2 * nb + 1)]
    us = [ops.pack_conv3x3_wino(x) for x in ws]

    def make(A, B):
        layers = [dict(x=lr, x2=s2d, u=us[0], bias=bs[0], cin=51, act=1, y=A)]
        for b in range(nb):
            layers.append(dict(x=A, u=us[1 + 2 * b], bias=bs[1 + 2 * b], cin=64, act=1, y=B))
            layers.append(dict(x=B, u=us[2 + 2 * b], bias=bs[2 + 2 * b], cin=64, act=0, res=A, y=A))
        return layers
    A1, B1, A2, B2 = (torch.empty(1, 64, h, w, device=dev) for _ in range(4))
    seq = make(A1, B1)
    print('supported:', ops.WinoResident.supported(64, h, w))
    res = ops.WinoResident(make(A2, B2), 64, h, w)

    def run_seq():
        for d in seq:
            ops.conv3x3_wino(d['x'], d['u'], d['bias'], d['cin'], 64, d['act'], x2=d.get('x2'), res=d.get('res'),
                             out=d['y'])

    def timeit(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    run_seq()
    res.run()
    torch.cuda.synchronize()
    print('bit-identical:', torch.equal(A1, A2), 'max diff', (A1 - A2).abs().max().item(), 'bailouts', res.bailouts())
    nl = 2 * nb + 1
    for rep in range(3):
        t_seq = timeit(run_seq)
        t_res = timeit(res.run)
        print('per-layer launches: %.1f us (%.2f us/layer)   resident: %.1f us (%.2f us/layer)   ratio %.3f'
              % (t_seq, t_seq / nl, t_res, t_res / nl, t_res / t_seq))
    print('bailouts', res.bailouts())
    if '--stamps' in sys.argv or os.environ.get('WRES_STAMPS'):
        import ctypes
        import numpy as np
        from tecogan_pytorch_amd import _lib
        lib = ctypes.CDLL(_lib.LIB_PATH)
        buf = (ctypes.c_longlong * (12 * 24 * 8))()
        res.run(); torch.cuda.synchronize()
        assert lib.tg_lab_wres_stamps(buf) == 0
        st = np.array(buf, dtype=np.int64).reshape(12, 24, 8)
        names = ['K loop', 'epilogue+publish', 'vmcnt(0)', 'barrier 1', 'flag+poll+barrier 2', 'ring loads', 'barrier 3']
        for wv in (0, 1, 5, 8, 11):
            for L in (3, 4, 9, 10):
                d = np.diff(st[wv, L])
                nxt = st[wv, L + 1, 0] - st[wv, L, 0]
                print('wave %2d layer %2d: ' % (wv, L) + '  '.join('%s %d' % (n, v) for n, v in zip(names, d)) + '  | layer period %d ticks' % nxt)
        t0 = st[:, 4, 0].min()
        print('layer 4 start skew over waves (ticks):', (st[:, 4, 0] - t0).tolist())
        print('layer 4 K-loop end (ticks after first start):', (st[:, 4, 1] - t0).tolist())


if __name__ == '__main__':
    main()
