"""VSRGANModel (TecoGAN): generator + spatio-temporal discriminator with the
adaptive D update, ping-pong loss and vanilla GAN loss
(codes/models/vsrgan_model.py).  feature_crit (VGG) / feature_matching_crit
are not built."""
from collections import OrderedDict

import torch

from .. import ops
from . import train_graph as TG
from .networks import define_discriminator
from .optim import Adam, define_criterion, define_lr_schedule
from .vsr_model import VSRModel


class VSRGANModel(VSRModel):
    def __init__(self, opt):
        super().__init__(opt)
        if self.is_train:
            self.cnt_upd_D = 0

    def set_networks(self):
        super().set_networks()
        if self.is_train:
            self.net_D = self.model_to_device(define_discriminator(self.opt))
            load_path_D = self.opt['model']['discriminator'].get('load_path', '')
            if load_path_D:
                self.load_network(self.net_D, load_path_D)

    def set_criterions(self):
        tr = self.opt['train']
        if tr.get('feature_crit') is not None or tr.get('feature_matching_crit') is not None:
            raise NotImplementedError('feature_crit / feature_matching_crit need the VGG19 extractor '
                                      '(torchvision weights): not built on the HIP path yet')
        self.pix_crit = define_criterion(tr.get('pixel_crit'))
        self.warp_crit = define_criterion(tr.get('warping_crit'))
        self.pp_crit = define_criterion(tr.get('pingpong_crit'))
        self.gan_crit = define_criterion(tr.get('gan_crit'))

    def set_optimizers(self):
        super().set_optimizers()
        d = self.opt['train']['discriminator']
        self.optim_D = Adam(self.net_D.parameters(), lr=d['lr'],
                            weight_decay=d.get('weight_decay', 0), betas=d.get('betas', (0.9, 0.999)))
        self.sched_D = define_lr_schedule(d.get('lr_schedule'), self.optim_D)

    def _sync_scalars(self, real_stats, fake_stats):
        """mean log-sigmoid of both passes, agreed across ranks with ONE 2-float
        all-reduce (the reference issues two all-reduces and a barrier, :166-173)."""
        v = torch.stack([real_stats[2], fake_stats[2]])
        if self.dist:
            import torch.distributed as dist
            dist.all_reduce(v)
            v = v / self.opt['world_size']
        return v.tolist()

    def train(self):
        """vsrgan_model.py:98-286."""
        opt_tr = self.opt['train']
        lr_data, gt_data = self.lr_data, self.gt_data
        n, t, c, lr_h, lr_w = lr_data.size()
        gt_h, gt_w = gt_data.shape[3:]
        up_mode = self.net_G.srnet.up_mode()
        bi_data = ops.upsample(lr_data.reshape(n * t, c, lr_h, lr_w).contiguous(), self.scale,
                               up_mode).view(n, t, c, gt_h, gt_w)
        if self.pp_crit is not None:        # ping-pong augmentation (:112-119)
            lr_data = torch.cat([lr_data, lr_data.flip(1)[:, 1:]], dim=1).contiguous()
            gt_data = torch.cat([gt_data, gt_data.flip(1)[:, 1:]], dim=1).contiguous()
            bi_data = torch.cat([bi_data, bi_data.flip(1)[:, 1:]], dim=1).contiguous()

        self.net_G.train()
        self.net_D.train()
        self.optim_G.zero_grad()
        self.optim_D.zero_grad()
        out = self.net_G(lr_data)
        tape_G = self.net_G.tape
        hr_data = out['hr_data']

        for p in self.net_D.parameters():
            p.requires_grad = True
        d_in = {'net_G': self.net_G, 'lr_data': lr_data, 'bi_data': bi_data,
                'use_pp_crit': self.pp_crit is not None,
                'crop_border_ratio': opt_tr['discriminator'].get('crop_border_ratio', 1.0)}
        d_in.update(out)
        tape_D = TG.Tape()
        d_in['tape'] = tape_D
        (real_pred, _), d_out = self.net_D(gt_data, d_in)
        d_in.update(d_out)
        (fake_pred, _), _ = self.net_D(hr_data, d_in)        # no input grad: == hr_data.detach()

        n_clip = real_pred.numel()
        st_real = torch.zeros(3, dtype=torch.float32, device=self.device)
        st_fake = torch.zeros(3, dtype=torch.float32, device=self.device)
        red = self.gan_crit[1]
        gsc = (1.0 / n_clip) if red == 'mean' else 1.0
        g_real = ops.bce_logits(real_pred, 1.0, st_real, 1.0 / n_clip, grad_scale=gsc)
        g_fake = ops.bce_logits(fake_pred, 0.0, st_fake, 1.0 / n_clip, grad_scale=gsc)

        update_policy = opt_tr['discriminator']['update_policy']
        if update_policy == 'adaptive':
            lreal, lfake = self._sync_scalars(st_real, st_fake)
            distance = lreal - lfake
            upd_D = distance < opt_tr['discriminator']['update_threshold']
        else:
            upd_D = True
        if upd_D:
            self.cnt_upd_D += 1.0
            tape_D.add_grad(real_pred, g_real)
            tape_D.add_grad(fake_pred, g_fake)
            tape_D.backward()
            self.allreduce_grads(self.net_D)
            self.optim_D.step()
        tape_D.nodes, tape_D.grads = [], {}

        # === generator === (D frozen, already updated: :201-202 after :188)
        for p in self.net_D.parameters():
            p.requires_grad = False
        losses = torch.zeros(3, dtype=torch.float32, device=self.device)
        if self.pix_crit is not None:
            w_ = opt_tr['pixel_crit'].get('weight', 1)
            tape_G.add_grad(hr_data, self._cb(hr_data, gt_data, w_, self.pix_crit[1], losses[0:1]))
        if self.warp_crit is not None:
            lr_warp = TG.backward_warp(tape_G, out['lr_prev'], out['lr_flow'], need_dimg=False)
            w_ = opt_tr['warping_crit'].get('weight', 1)
            tape_G.add_grad(lr_warp, self._cb(lr_warp, out['lr_curr'], w_, self.warp_crit[1],
                                              losses[1:2]))
        if self.pp_crit is not None:
            te = opt_tr['tempo_extent']
            hr_fw = hr_data[:, :te - 1].contiguous()
            hr_bw = hr_data[:, te:].flip(1).contiguous()
            w_ = opt_tr['pingpong_crit'].get('weight', 1)
            scale = w_ / hr_fw.numel() if self.pp_crit[1] == 'mean' else w_
            g = ops.charbonnier(hr_fw, hr_bw, losses[2:3], scale, grad_scale=scale)
            full = torch.zeros_like(hr_data)
            full[:, :te - 1] = g
            tape_G.add_grad(hr_data, full)
            full2 = torch.zeros_like(hr_data)
            full2[:, te:] = g.flip(1)
            tape_G.add_grad(hr_data, self._neg(full2))
        d_in['tape'] = tape_G
        d_in['need_input_grad'] = True
        (fake_pred_G, _), _ = self.net_D(hr_data, d_in)
        st_g = torch.zeros(3, dtype=torch.float32, device=self.device)
        gan_w = opt_tr['gan_crit'].get('weight', 1)
        tape_G.add_grad(fake_pred_G, ops.bce_logits(fake_pred_G, 1.0, st_g, 1.0 / n_clip,
                                                    grad_scale=gan_w * gsc))
        tape_G.backward()
        self.allreduce_grads(self.net_G)
        self.optim_G.step()

        # === logging: one host read of all scalars ===
        sr, sf, sg, ls = st_real.tolist(), st_fake.tolist(), st_g.tolist(), losses.tolist()
        self.log_dict = OrderedDict()
        self.log_dict['l_gan_D'] = (sr[0] + sf[0]) if upd_D else 0.0
        self.log_dict['p_real_D'] = sr[1]
        self.log_dict['p_fake_D'] = sf[1]
        if update_policy == 'adaptive':
            self.log_dict['distance'] = distance
            self.log_dict['n_upd_D'] = self.cnt_upd_D
        if self.pix_crit is not None:
            self.log_dict['l_pix_G'] = ls[0]
        if self.warp_crit is not None:
            self.log_dict['l_warp_G'] = ls[1]
        if self.pp_crit is not None:
            self.log_dict['l_pp_G'] = ls[2]
        self.log_dict['l_gan_G'] = gan_w * sg[0]
        self.log_dict['p_fake_G'] = sg[1]

    @staticmethod
    def _neg(t):
        """-t through the axpy kernel (y = 0 + (-1) * t)."""
        out = torch.zeros_like(t)
        ops.axpy_(out, t, -1.0)
        return out

    def save(self, current_iter):
        self.save_network(self.net_G, 'G', current_iter)
        self.save_network(self.net_D, 'D', current_iter)
