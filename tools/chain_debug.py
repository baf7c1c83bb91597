import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.test_hip_train_ops import _srnet, dev, rs, _body_both_ways, relerr
from tecogan_pytorch_amd.models import train_graph as TG
n, h, w, nb = 2, 32, 32, int(os.environ.get('NB', '10'))
net = _srnet(nb)
lr, tran, g = dev(rs(1, (n, 3, h, w), 0, 1)), dev(rs(2, (n, 48, h, w), 0, 1)), dev(rs(3, (n, 64, h, w)))
(o1, t1, g1), (o2, t2, g2) = _body_both_ways(net, lr, tran, g)
d = (t1 - t2).abs()
print('d_tran max err', d.max().item(), 'scale', t2.abs().max().item(), 'bad elems', int((d > 1e-4 * t2.abs().max()).sum()), 'of', d.numel())
bad = (d > 1e-4 * t2.abs().max())
print('bad per image', bad.sum((1, 2, 3)).tolist())
print('bad per channel', bad.sum((0, 2, 3)).tolist())
print('bad per row', bad.sum((0, 1, 3)).tolist())
print('bad per col', bad.sum((0, 1, 2)).tolist())
for k in g1:
    print(k, relerr(g1[k], g2[k]))
