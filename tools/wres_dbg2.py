import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tecogan_pytorch_amd.ops as ops
h, w = 16, 48
g = torch.Generator().manual_seed(1)
x = torch.rand(1, 64, h, w, generator=g).cuda()
w0 = (torch.randn(64, 64, 3, 3, generator=g) * 0.03).cuda()
b0 = (torch.randn(64, generator=g) * 0.1).cuda()
for name, (ky, kx) in (('right ring', (1, 2)), ('left ring', (1, 0)), ('bottom ring', (2, 1)), ('top ring', (0, 1)), ('SE corner', (2, 2))):
    w1 = torch.zeros(64, 64, 3, 3)
    for c in range(64): w1[c, c, ky, kx] = 1.0
    w1 = w1.cuda(); b1 = torch.zeros(64).cuda()
    us = [ops.pack_conv3x3_wino(w0), ops.pack_conv3x3_wino(w1)]
    A1, B1, A2, B2 = (torch.empty(1, 64, h, w, device='cuda') for _ in range(4))
    def make(A, B):
        return [dict(x=x, u=us[0], bias=b0, cin=64, act=1, y=A), dict(x=A, u=us[1], bias=b1, cin=64, act=1, y=B)]
    for d in make(A1, B1):
        ops.conv3x3_wino(d['x'], d['u'], d['bias'], d['cin'], 64, d['act'], out=d['y'])
    res = ops.WinoResident(make(A2, B2), 64, h, w)
    res.run(); torch.cuda.synchronize()
    bad = ((B1 - B2).abs() > 1e-5)[0]          # (64, h, w)
    print(name, 'bad elements', int(bad.sum()), 'bad channels', sorted(set(bad.nonzero()[:, 0].tolist()))[:70])
    print('   bad positions (y,x):', sorted(set((int(a), int(b)) for a, b in bad.nonzero()[:, 1:].tolist()))[:40])
