"""VSRModel (FRVSR): generator only, Charbonnier pixel + warping loss
(codes/models/vsr_model.py)."""
from collections import OrderedDict

import torch

from .. import ops
from . import train_graph as TG
from .base_model import BaseModel
from .networks import define_generator
from .optim import Adam, define_criterion, define_lr_schedule, pointwise_loss


class VSRModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.set_networks()
        if self.is_train:
            self.set_criterions()
            self.set_optimizers()
            self.check_ranks_agree()

    def set_networks(self):
        self.net_G = self.model_to_device(define_generator(self.opt))
        load_path_G = self.opt['model']['generator'].get('load_path')
        if load_path_G:
            self.load_network(self.net_G, load_path_G)

    def set_criterions(self):
        self.pix_crit = define_criterion(self.opt['train'].get('pixel_crit'))
        self.warp_crit = define_criterion(self.opt['train'].get('warping_crit'))

    def set_optimizers(self):
        g = self.opt['train']['generator']
        self.optim_G = Adam(self.net_G.parameters(), lr=g['lr'],
                            weight_decay=g.get('weight_decay', 0), betas=g.get('betas', (0.9, 0.999)))
        self.sched_G = define_lr_schedule(g.get('lr_schedule'), self.optim_G)

    # -- loss helpers (value accumulates on the device; gradient returned) -----
    @staticmethod
    def _crit(crit, x, y, weight, acc):
        return pointwise_loss(crit, x, y, weight, acc)

    def train(self):
        """vsr_model.py:61-95."""
        self.net_G.train()
        self.optim_G.zero_grad()
        out = self.net_G(self.lr_data)
        tape = self.net_G.tape
        self.hr_data = out['hr_data']
        losses = torch.zeros(3, dtype=torch.float32, device=self.device)     # [pixel, warp, fault slot]
        pix_w = self.opt['train']['pixel_crit'].get('weight', 1.0)
        tape.add_grad(out['hr_data'], self._crit(self.pix_crit, out['hr_data'],
                                                 self.gt_data.contiguous(), pix_w, losses[0:1]))
        if self.warp_crit is not None:
            lr_warp = TG.backward_warp(tape, out['lr_prev'], out['lr_flow'], need_dimg=False)
            warp_w = self.opt['train']['warping_crit'].get('weight', 1.0)
            tape.add_grad(lr_warp, self._crit(self.warp_crit, lr_warp, out['lr_curr'], warp_w,
                                              losses[1:2]))
        tape.backward()
        TG.stamp_fault(self.optim_G)                 # a chained-launch fault (any rank) turns the step into a no-op
        self.allreduce_grads(self.net_G, 'G')
        self.optim_G.step()
        if getattr(self.optim_G, 'fault_slot', None) is not None:
            losses[2:3].copy_(self.optim_G.fault_slot)
        has_warp = self.warp_crit is not None
        ep, optim_G = TG.chain_epoch(), self.optim_G

        def build(vals):                             # runs when the log is looked at (base_model: asynchronous scalars)
            try:                                     # fail-safe of the chained launches: raises on EVERY rank, the update was dropped
                dropped = TG.chain_check(vals[2], counter=False, epoch=ep)
            except Exception:
                optim_G.undo_step_count()
                raise
            if dropped:                              # in flight behind an iteration that already raised
                optim_G.undo_step_count()
            d = OrderedDict(l_pix_G=vals[0])
            if has_warp:
                d['l_warp_G'] = vals[1]
            return d
        self._set_pending_log(losses, build)

    def infer(self, device_output=False):
        """vsr_model.py:97-113: temporal padding, inference, crop the padding back.
        device_output=True keeps the (t,H,W,3) uint8 result on the GPU (for the on-device
        PSNR) instead of returning the reference's numpy array."""
        lr_data, n_pad_front = self.pad_sequence(self.lr_data)
        self.net_G.eval()
        if device_output:
            hr_seq = self.net_G.infer_sequence(lr_data, self.device, return_device_tensor=True)
        else:
            hr_seq = self.net_G(lr_data, self.device)
        return hr_seq[n_pad_front:]

    def save(self, current_iter):
        self.sync_log()          # (a pending fault check must run before weights are written)
        self.save_network(self.net_G, 'G', current_iter)
