"""convout_tail forms (one / four HR pixels per thread), stand-alone timing at 536x1280."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tecogan_pytorch_amd  # noqa
from tecogan_pytorch_amd import ops
g = torch.Generator().manual_seed(0)
z = torch.rand(1, 32, 536, 1280, generator=g).cuda()
b = torch.rand(3, generator=g).cuda()
src = torch.rand(1, 3, 134, 320, generator=g).cuda()
for rep in range(2):
    for form in (0, 1):
        f = lambda: ops.convout_tail(z, 3, b, up_src=src, up_mode=ops.UP_BICUBIC, up_scale=4, want_u8=True, form=form)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f'form {form}: {us:6.1f} us  {84.0 / us:5.2f} TB/s (84 MB algorithmic)', flush=True)
