"""Z-mode transposed conv (inference's last up-sampling layer): tiled vs streaming form, stand-alone timing.
  python tools/convtz_lab.py [h w]      (default 268 640: the 4x frame's second up-sampling layer)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tecogan_pytorch_amd  # noqa
from tecogan_pytorch_amd import ops

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (268, 640)
g = torch.Generator().manual_seed(0)
x = torch.rand(1, 64, h, w, generator=g).cuda()
wt = (torch.rand(64, 64, 3, 3, generator=g) - 0.5).cuda() / 12
b = (torch.rand(64, generator=g) - 0.5).cuda()
wo = (torch.rand(3, 64, 3, 3, generator=g) - 0.5).cuda() / 24
pk = ops.pack_conv3x3(wt, transposed=True)[0]
wz = ops.convt_pack_wz(wo)
outs = {}
gflop = (2.0 * 64 * 9 * 64 * h * w + 2.0 * 64 * 27 * 4 * h * w) / 1e9
for rep in range(2):
    for form in (0, 1, 2, 3):
        out = torch.empty(1, 32, 2 * h, 2 * w, device='cuda')
        for _ in range(5):
            ops.convt3x3s2_z(x, pk, b, wz, 3, 64, act=1, form=form, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            ops.convt3x3s2_z(x, pk, b, wz, 3, 64, act=1, form=form, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        outs[form] = out[:, :27].clone()
        print(f'form {form} ({("tiled", "streaming", "streaming, static list", "tiled, split tail")[form]}): {us:7.1f} us  {gflop / us * 1e3:6.1f} TFLOP/s '
              f'({gflop / us * 1e3 / 157.3:.2f} of peak)', flush=True)
print('bit-identical:', torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]), torch.equal(outs[0], outs[3]), 'max diff', (outs[0] - outs[1]).abs().max().item())
