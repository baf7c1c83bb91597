"""Probe: SRNet's 21-layer chain (conv_in + 10 residual blocks, in-place residual sums, two ping-pong
tensors) as ONE launch (tg_conv3x3_wino_chain) against 21 launches of the same kernel: results must be
bit-identical; time per layer with HIP events."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tecogan_pytorch_amd import ops

torch.manual_seed(0)
dev = 'cuda'
h, w, nb = (int(sys.argv[1]), int(sys.argv[2])) + (10,) if len(sys.argv) > 2 else (134, 320, 10)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lr = torch.rand(n, 3, h, w, device=dev)
s2d = torch.rand(n, 48, h, w, device=dev)
ws = [torch.randn(64, 51, 3, 3, device=dev) * 0.04] + [torch.randn(64, 64, 3, 3, device=dev) * 0.03 for _ in range(2 * nb)]
bs = [torch.randn(64, device=dev) * 0.1 for _ in range(2 * nb + 1)]
us = [ops.pack_conv3x3_wino(x) for x in ws]


def make(A, B):
    layers = [dict(x=lr, x2=s2d, u=us[0], bias=bs[0], cin=51, act=ops.ACT_RELU, y=A)]
    for b in range(nb):
        layers.append(dict(x=A, u=us[1 + 2 * b], bias=bs[1 + 2 * b], cin=64, act=ops.ACT_RELU, y=B))
        layers.append(dict(x=B, u=us[2 + 2 * b], bias=bs[2 + 2 * b], cin=64, act=ops.ACT_NONE, res=A, y=A))
    return layers


def sequential(layers):
    for d in layers:
        ops.conv3x3_wino(d['x'], d['u'], d['bias'], d['cin'], 64, d['act'], x2=d.get('x2'), res=d.get('res'), out=d['y'])


A1, B1 = torch.empty(n, 64, h, w, device=dev), torch.empty(n, 64, h, w, device=dev)
A2, B2 = torch.empty(n, 64, h, w, device=dev), torch.empty(n, 64, h, w, device=dev)
seq_layers, chain = make(A1, B1), ops.WinoChain(make(A2, B2), n, 64, h, w)
bad = 0
for it in range(30):
    lr.uniform_(); s2d.uniform_()
    sequential(seq_layers)
    chain.run()
    torch.cuda.synchronize()
    if not (torch.equal(A1, A2) and torch.equal(B1, B2)):
        bad += 1
        print('iteration', it, 'max diff', float((A1 - A2).abs().max()), float((B1 - B2).abs().max()), flush=True)
print('iterations with a mismatch:', bad, ' poll bail-outs:', chain.bailouts(), flush=True)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_seq, t_chain = timed(lambda: sequential(seq_layers)), timed(chain.run)
nl = 2 * nb + 1
print(f'{n}x{h}x{w}: {nl} launches {t_seq:.0f} us ({t_seq / nl:.1f} per layer);  one chained launch {t_chain:.0f} us '
      f'({t_chain / nl:.1f} per layer)  bail-outs {chain.bailouts()}', flush=True)
