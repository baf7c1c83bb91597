"""Losses and optimiser of the training step as HIP launches
(codes/models/optim/losses.py, torch.optim.Adam at vsr_model.py:47-52)."""
import torch

from .. import ops


class Adam:
    """torch.optim.Adam semantics (no amsgrad) with one fused kernel per tensor."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.param_groups = [{'lr': lr, 'betas': tuple(betas), 'eps': eps,
                              'weight_decay': weight_decay}]
        self.state = {}
        self.steps = 0

    def zero_grad(self):
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            else:
                p.grad.zero_()              # memset

    def step(self):
        g = self.param_groups[0]
        self.steps += 1
        for p in self.params:
            if p.grad is None or not p.requires_grad:
                continue
            st = self.state.get(id(p))
            if st is None:
                st = self.state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
            ops.adam_step(p.data, p.grad, st[0], st[1], g['lr'], g['betas'], g['eps'],
                          g['weight_decay'], self.steps)
            # the kernel writes through the raw pointer, which torch's version counter does
            # not see; the packed-weight caches key on this explicit counter as well
            ops.bump_version(p)


def define_criterion(criterion_opt):
    """codes/models/optim/__init__.py:5-35 for the criteria the shipped configs use."""
    if criterion_opt is None:
        return None
    if criterion_opt['type'] == 'CB':
        return ('CB', criterion_opt.get('reduction', 'mean'))
    if criterion_opt['type'] == 'GAN':
        return ('GAN', criterion_opt.get('reduction', 'mean'))
    raise ValueError(f'Unrecognized criterion: {criterion_opt["type"]}')
