"""Soak: a long 2-clip (and 3-clip) inference with the chained SRNet launch against the same run with
one launch per layer -- every frame must be bit-identical (a stale halo or a lost flag would show).
  python tools/chain_soak.py [frames]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
script = (
    "import sys, torch; sys.path.insert(0, %r)\n"
    "import tecogan_pytorch_amd\n"
    "from tecogan_pytorch_amd.models.networks import FRNet\n"
    "torch.manual_seed(0)\n"
    "net = FRNet(3, 3, 64, 10, 'BD', 4).cuda().eval()\n"
    "out = []\n"
    "for k in (2, 3):\n"
    "    g = torch.Generator().manual_seed(k)\n"
    "    x = torch.rand(k, %d, 3, 134, 320, generator=g).cuda()\n"
    "    y = net.infer_sequence(x, torch.device('cuda'), return_device_tensor=True)\n"
    "    out.append(y.cpu())\n"
    "torch.save(out, sys.argv[1])\n" % (ROOT, frames))
res = []
with tempfile.TemporaryDirectory() as d:
    for chain in ('0', '1'):
        p = os.path.join(d, 'y%s.pt' % chain)
        subprocess.run([sys.executable, '-c', script, p], check=True, env=dict(os.environ, TG_WINO_CHAIN=chain))
        import torch
        res.append(torch.load(p))
ok = all(torch.equal(a, b) for a, b in zip(*res))
print('chained == separate launches over %d frames x (2 + 3) clips:' % frames, ok)
sys.exit(0 if ok else 1)
