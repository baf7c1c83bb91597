"""wgrad3x3 on the batched SRNet layer of the training step (19 frames x 2 x 64 x 64 x 64ch, or 32x32),
timed alone.  With the lab library (TECOGAN_HIP_LIB=tools/_lab_libs/libtecogan_lab.so): TG_WGRAD_ABL
bits 1 no global loads after the first tile, 2 no LDS stores, 4 no MFMAs, 8 no LDS operand reads;
TG_WGRAD_MAXWG caps the K split."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tecogan_pytorch_amd import ops
hw = int(os.environ.get('HW', '64'))
frames = 19
p = [0.1 * torch.randn(2, 64, hw, hw, device='cuda') for _ in range(frames)]
q = [0.1 * torch.randn(2, 64, hw, hw, device='cuda') for _ in range(frames)]
g = torch.zeros(64, 64, 3, 3, device='cuda')
for _ in range(3):
    ops.wgrad3x3_multi(p, q, g, accumulate=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
R = 20
for _ in range(R):
    ops.wgrad3x3_multi(p, q, g, accumulate=False)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / R * 1e3
fl = 2.0 * 64 * 64 * 9 * frames * 2 * hw * hw
print(f'hw={hw} ABL={os.environ.get("TG_WGRAD_ABL", "0")} MAXWG={os.environ.get("TG_WGRAD_MAXWG", "-")}: wgrad + reduce {us:.1f} us  ({fl / us / 1e6:.1f} TFLOP/s)')
