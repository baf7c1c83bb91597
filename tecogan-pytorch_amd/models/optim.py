"""Losses and optimiser of the training step as HIP launches
(codes/models/optim/losses.py, torch.optim.Adam at vsr_model.py:47-52)."""
import torch

from .. import ops


class Adam:
    """torch.optim.Adam semantics (no amsgrad).

    Parameters, gradients and both moments of a network live in FOUR flat device buffers
    (each tensor a 256-byte aligned view): zero_grad is one memset, the step is ONE fused
    kernel over the whole network instead of one per tensor (78 + 21 launches for G + D),
    and the data-parallel gradient exchange all-reduces the gradient buffer in place
    (`self.flat_grad`, DDP's gradient_as_bucket_view) -- no concatenation, no copy back.
    The padding between tensors stays zero (zero gradient => zero update)."""

    ALIGN = 64      # floats

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, flatten=True):
        self.params = [p for p in params]
        self.param_groups = [{'lr': lr, 'betas': tuple(betas), 'eps': eps,
                              'weight_decay': weight_decay}]
        self.state = {}
        self.steps = 0
        self.flat_param = self.flat_grad = self.flat_m = self.flat_v = None
        self.fault_slot = None
        if flatten and self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params):
            self._flatten()

    def _flatten(self):
        offs, tot = [], 0
        for p in self.params:
            offs.append(tot)
            tot += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        dev = self.params[0].device
        # one more aligned block behind the last tensor: its first float is the FAULT SLOT of the chained
        # launches' fail-safe -- it travels with the gradient all-reduce, and a non-zero value turns this
        # network's step into a no-op on every rank (models/train_graph.py: stamp_fault)
        self._ntot = tot
        self.flat_param = torch.zeros(tot + self.ALIGN, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(tot + self.ALIGN, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(tot + self.ALIGN, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(tot + self.ALIGN, dtype=torch.float32, device=dev)
        self.fault_slot = self.flat_grad[tot:tot + 1]
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                k = p.numel()
                view = self.flat_param[o:o + k].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + k].view_as(p)
                self.state[id(p)] = (self.flat_m[o:o + k].view_as(p), self.flat_v[o:o + k].view_as(p))
        self._offs = offs

    def _is_flat(self):
        """The views can be replaced behind our back (net.to(), p.grad = None): check cheaply."""
        if self.flat_param is None:
            return False
        p0, pl = self.params[0], self.params[-1]
        base, gbase = self.flat_param.data_ptr(), self.flat_grad.data_ptr()
        return (p0.data_ptr() == base and p0.grad is not None and p0.grad.data_ptr() == gbase and
                pl.data_ptr() == base + 4 * self._offs[-1] and pl.grad is not None and
                pl.grad.data_ptr() == gbase + 4 * self._offs[-1])

    def zero_grad(self):
        if self._is_flat():
            self.flat_grad.zero_()          # one memset for the whole network
            return
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            else:
                p.grad.zero_()              # memset

    def undo_step_count(self):
        """A step whose update the device-side guard DROPPED (chained-launch fault, models/train_graph.py) must not
        advance the bias-correction exponent: the caller takes the count back once it learns of the drop."""
        if self.steps > 0:
            self.steps -= 1

    def step(self):
        g = self.param_groups[0]
        self.steps += 1
        flat = self._is_flat()
        if flat and all(p.requires_grad for p in self.params):
            n = self._ntot
            ops.adam_step(self.flat_param[:n], self.flat_grad[:n], self.flat_m[:n], self.flat_v[:n], g['lr'],
                          g['betas'], g['eps'], g['weight_decay'], self.steps, skip=self.fault_slot)
            for p in self.params:
                ops.bump_version(p)
            return
        for p in self.params:
            if p.grad is None or not p.requires_grad:
                continue
            st = self.state.get(id(p))
            if st is None:
                st = self.state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
            # (frozen parameters: the per-tensor launches carry the same device-side guard as the fused one)
            ops.adam_step(p.data, p.grad, st[0], st[1], g['lr'], g['betas'], g['eps'],
                          g['weight_decay'], self.steps, skip=self.fault_slot if flat else None)
            # the kernel writes through the raw pointer, which torch's version counter does
            # not see; the packed-weight caches key on this explicit counter as well
            ops.bump_version(p)


def _adam_state_dict(self):
    """Moments in parameter order (the reference saves weights only -- its resume path is a
    TODO at base_model.py:220-222; this is the missing half of a restartable run)."""
    st = []
    for p in self.params:
        e = self.state.get(id(p))
        st.append(None if e is None else (e[0].detach().cpu(), e[1].detach().cpu()))
    return {'steps': self.steps, 'param_groups': [dict(g) for g in self.param_groups], 'state': st}


def _adam_load_state_dict(self, sd):
    if len(sd['state']) != len(self.params):
        raise ValueError(f"optimizer state has {len(sd['state'])} entries, the network has "
                         f"{len(self.params)} parameters")
    self.steps = int(sd['steps'])
    # hyper-parameters come from the CURRENT configuration (yml), as torch's schedulers expect
    # after a resume; only the learning rate in force at the checkpoint is restored (the
    # schedule object recomputes it from its own position on the next step anyway)
    for g, src in zip(self.param_groups, sd['param_groups']):
        if 'lr' in src:
            g['lr'] = src['lr']
    for p, e in zip(self.params, sd['state']):
        if e is not None:
            if e[0].shape != p.shape:
                raise ValueError(f'optimizer state shape {tuple(e[0].shape)} vs parameter {tuple(p.shape)}')
            cur = self.state.get(id(p))
            if cur is not None:                      # flat storage: fill the views in place
                cur[0].copy_(e[0])
                cur[1].copy_(e[1])
            else:
                self.state[id(p)] = (e[0].to(p.device).contiguous(), e[1].to(p.device).contiguous())


Adam.state_dict = _adam_state_dict
Adam.load_state_dict = _adam_load_state_dict


def define_criterion(criterion_opt):
    """codes/models/optim/__init__.py:5-35: (type, reduction) descriptors; the arithmetic is in
    the HIP loss kernels."""
    if criterion_opt is None:
        return None
    kind = criterion_opt['type']
    if kind in ('CB', 'L1', 'MSE', 'GAN', 'LSGAN'):
        return (kind, criterion_opt.get('reduction', 'mean'))
    if kind == 'CosineSimilarity':
        return (kind, 'mean')                      # losses.py:53-62 takes no reduction
    raise ValueError(f'Unrecognized criterion: {criterion_opt["type"]}')


def pointwise_loss(crit, x, y, weight, acc):
    """weight * crit(x, y) for the element-wise criteria: the value is accumulated into the
    device scalar `acc`, the gradient w.r.t. x is returned."""
    kind, reduction = crit
    scale = weight / x.numel() if reduction == 'mean' else weight
    if kind == 'CB':
        return ops.charbonnier(x, y, acc, scale, grad_scale=scale)
    if kind in ('L1', 'MSE'):
        return ops.pixel_loss(x, y, ops.LOSS_L1 if kind == 'L1' else ops.LOSS_MSE, acc, scale,
                              grad_scale=scale)
    raise ValueError(f'{kind} is not an element-wise criterion')


class _Schedule:
    """Closed-form learning-rate schedules (host arithmetic on the optimiser's `lr`)."""

    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.base_lr = optimizer.param_groups[0]['lr']
        self.last_epoch = 0

    def lr_at(self, it):
        raise NotImplementedError

    def step(self):
        self.last_epoch += 1
        self.optimizer.param_groups[0]['lr'] = self.lr_at(self.last_epoch)


class MultiStepLR(_Schedule):
    """lr = base * gamma ** (number of milestones <= iteration), as torch's MultiStepLR
    (selected by the shipped FRVSR ymls)."""

    def __init__(self, optimizer, milestones, gamma):
        super().__init__(optimizer)
        self.milestones, self.gamma = sorted(milestones), gamma

    def lr_at(self, it):
        return self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= it)


class CosineAnnealingRestartLR(_Schedule):
    """eta_min + w * 0.5 * (base - eta_min) * (1 + cos(pi * (it - restart) / period))
    (codes/models/optim/lr_schedules.py:30-78)."""

    def __init__(self, optimizer, periods, restart_weights=(1,), eta_min=0):
        super().__init__(optimizer)
        assert len(periods) == len(restart_weights)
        self.periods, self.weights, self.eta_min = list(periods), list(restart_weights), eta_min
        self.cum = [sum(self.periods[:i + 1]) for i in range(len(self.periods))]

    def lr_at(self, it):
        import math
        idx = next((i for i, p in enumerate(self.cum) if it <= p), len(self.cum) - 1)
        restart = 0 if idx == 0 else self.cum[idx - 1]
        return self.eta_min + self.weights[idx] * 0.5 * (self.base_lr - self.eta_min) * (
            1 + math.cos(math.pi * ((it - restart) / self.periods[idx])))


def define_lr_schedule(schedule_opt, optimizer):
    """codes/models/optim/__init__.py:38-62."""
    if schedule_opt is None or schedule_opt['type'] == 'FixedLR':
        return None
    if schedule_opt['type'] == 'MultiStepLR':
        return MultiStepLR(optimizer, schedule_opt['milestones'], schedule_opt['gamma'])
    if schedule_opt['type'] == 'CosineAnnealingRestartLR':
        return CosineAnnealingRestartLR(optimizer, schedule_opt['periods'],
                                        schedule_opt['restart_weights'], schedule_opt['eta_min'])
    raise ValueError(f'Unrecognized lr schedule: {schedule_opt["type"]}')
