"""Probe: would running the top and bottom half of the frame as two independent kernel chains on two
streams (so that one chain's prologue / epilogue overlaps the other's MFMA phase) beat one chain of
full-frame launches?  20 dependent 64->64 layers, Winograd kernel, HIP events."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tecogan_pytorch_amd import ops

torch.manual_seed(0)
dev = 'cuda'
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.03
b = torch.zeros(64, device=dev)
u = ops.pack_conv3x3_wino(wt)


def chain(x, y, layers=20):
    a, c = x, y
    for _ in range(layers):
        ops.conv3x3_wino(a, u, b, 64, 64, ops.ACT_RELU, out=c)
        a, c = c, a


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


full = [torch.rand(1, 64, 134, 320, device=dev) for _ in range(2)]
halves = [[torch.rand(1, 64, 68, 320, device=dev) for _ in range(2)] for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


import ctypes
from tecogan_pytorch_amd import _lib as L
lab = ctypes.CDLL(os.environ['TECOGAN_HIP_LIB'])      # tools/_lab_libs/libtecogan_wino_lab.so
lab.tg_lab_wino_chains.restype = ctypes.c_int
lab.tg_lab_wino_chains.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int]


def one():
    st = torch.cuda.current_stream().cuda_stream
    assert lab.tg_lab_wino_chains(full[0].data_ptr(), full[1].data_ptr(), None, None, u.data_ptr(), b.data_ptr(),
                                  20, 134, 320, st, st, 0) == 0


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    assert lab.tg_lab_wino_chains(halves[0][0].data_ptr(), halves[0][1].data_ptr(), halves[1][0].data_ptr(),
                                  halves[1][1].data_ptr(), u.data_ptr(), b.data_ptr(), 20, 68, 320,
                                  s1.cuda_stream, s2.cuda_stream, 1) == 0
    cur.wait_stream(s1); cur.wait_stream(s2)


t1, t2 = timed(one), timed(two)
print(f'20 layers 134x320 one chain: {t1:.0f} us ({t1 / 20:.1f} per layer);  two 68x320 chains on two streams: '
      f'{t2:.0f} us ({t2 / 20:.1f} per layer pair)')
