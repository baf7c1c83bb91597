"""Per-call timing of selected ops inside one training step (events + sync around each call; the
step is slower, the per-call numbers are what matters).  python tools/time_ops.py --crop 256 [--ops wgrad]"""
import argparse, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser(); ap.add_argument('--crop', type=int, default=256); ap.add_argument('--ops', default='wgrad3x3,wgrad3x3_multi,wgrad3x3_body,bias_grad_multi,bias_grad,bias_grad_body')
a = ap.parse_args()
from tecogan_pytorch_amd import ops
from tecogan_pytorch_amd.models import define_model
rows = []
def wrap(name):
    fn = getattr(ops, name)
    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        r = fn(*args, **kw)
        e1.record(); torch.cuda.synchronize()
        def sh(x):
            if torch.is_tensor(x): return tuple(x.shape)
            if isinstance(x, (list, tuple)) and x and torch.is_tensor(x[0]): return (len(x),) + tuple(x[0].shape)
            return x
        rows.append((name, e0.elapsed_time(e1) * 1e3, [sh(x) for x in args[:3]], {k: sh(v) for k, v in kw.items()}))
        return r
    setattr(ops, name, w)
opt = {'scale': 4, 'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1, 'is_train': True,
       'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5}, 'train': {'crop_size': a.crop}},
       'model': {'name': 'TecoGAN', 'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10},
                 'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3}},
       'train': {'tempo_extent': 10, 'ckpt_dir': '/tmp', 'generator': {'lr': 5e-5, 'betas': [0.9, 0.999]},
                 'discriminator': {'update_policy': 'adaptive', 'update_threshold': 1e9, 'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                 'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'}, 'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                 'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'}, 'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
       'logger': {'decay': 0.99}}
torch.manual_seed(0)
m = define_model(opt)
data = {'gt': torch.rand(2, 10, 3, a.crop + 8, a.crop + 8).cuda()}
for _ in range(2):
    m.prepare_training_data(data); m.train()
for name in a.ops.split(','):
    wrap(name)
rows.clear()
m.prepare_training_data(data); m.train()
tot = collections.Counter()
for name, us, sh, kw in rows:
    tot[name] += us
    print(f'{name:18s} {us:9.1f} us  {sh} {kw if kw else ""}')
print({k: round(v / 1e3, 3) for k, v in tot.items()}, 'ms;  sum', round(sum(tot.values()) / 1e3, 3), 'ms')
