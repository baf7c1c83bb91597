"""Lab: who calls Tensor.copy_ / clone / contiguous-with-copy / zeros / torch.cat during one TecoGAN training step (crop 128)?"""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(__file__), 'host_profile.py')).read().split("torch.cuda.synchronize()\npr = cProfile")[0])
cnt = collections.Counter()
def wrap(name, fn):
    def f(*a, **k):
        fr = traceback.extract_stack(limit=4)[:-1]
        cnt[(name, ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in reversed(fr)))] += 1
        return fn(*a, **k)
    return f
torch.Tensor.copy_ = wrap('copy_', torch.Tensor.copy_)
torch.Tensor.clone = wrap('clone', torch.Tensor.clone)
_oc = torch.Tensor.contiguous
def contig(self, *a, **k):
    if not self.is_contiguous():
        fr = traceback.extract_stack(limit=3)[:-1]
        cnt[('contiguous(copy)', ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in reversed(fr)))] += 1
    return _oc(self, *a, **k)
torch.Tensor.contiguous = contig
for nm in ('zeros', 'zeros_like', 'cat', 'stack'):
    setattr(torch, nm, wrap(nm, getattr(torch, nm)))
m.prepare_training_data(data); m.train()
torch.cuda.synchronize()
for k, v in cnt.most_common(30):
    print(v, k)
