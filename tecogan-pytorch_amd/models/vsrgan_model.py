"""VSRGANModel (TecoGAN): generator + spatio-temporal discriminator with the
adaptive D update, ping-pong loss and vanilla GAN loss
(codes/models/vsrgan_model.py), the VGG19 perceptual loss (feature_crit, :226-241) and the
discriminator feature-matching loss (feature_matching_crit, :255-271)."""
import os
from collections import OrderedDict

import torch

from .. import ops
from . import train_graph as TG
from .networks import define_discriminator
from .networks.vgg_nets import VGGFeatureExtractor
from .optim import Adam, define_criterion, define_lr_schedule, pointwise_loss
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
        """vsrgan_model.py:50-72."""
        tr = self.opt['train']
        # the critic's real and fake pass of the D update as one stacked pass (SpatioTemporalDiscriminator.forward_pair);
        # `train.discriminator.pair_pass: false` selects the reference's two separate passes
        self.pair_pass = bool(tr.get('discriminator', {}).get('pair_pass', True))
        self.pix_crit = define_criterion(tr.get('pixel_crit'))
        self.warp_crit = define_criterion(tr.get('warping_crit'))
        self.feat_crit = define_criterion(tr.get('feature_crit'))
        if self.feat_crit is not None:
            if self.feat_crit[0] not in ('CosineSimilarity', 'L1', 'MSE', 'CB'):
                raise ValueError(f'feature_crit type {self.feat_crit[0]} is not a feature criterion')
            fc = tr['feature_crit']
            self.net_F = VGGFeatureExtractor(fc.get('feature_layers', [8, 17, 26, 35])).to(self.device)
            self._load_vgg(fc)
        self.pp_crit = define_criterion(tr.get('pingpong_crit'))
        self.fm_crit = define_criterion(tr.get('feature_matching_crit'))
        self.gan_crit = define_criterion(tr.get('gan_crit'))

    def _load_vgg(self, fc):
        """The reference downloads torchvision's ImageNet weights (vgg_nets.py:10).  Here they
        are read from `train.feature_crit.load_path` (or $TECOGAN_VGG19_PTH): a torchvision
        vgg19 state dict.  `init: default` keeps the seeded default initialisation instead
        (benchmarks / parity tests only) -- never silently."""
        path = fc.get('load_path') or os.environ.get('TECOGAN_VGG19_PTH')
        if path:
            self.net_F.load_vgg19_state_dict(torch.load(path, map_location='cpu'))
        elif fc.get('init') != 'default':
            raise FileNotFoundError(
                'feature_crit needs the ImageNet VGG19 weights: set train.feature_crit.load_path '
                '(or TECOGAN_VGG19_PTH) to a torchvision vgg19 state dict, or train.feature_crit.init: '
                'default to run with untrained features')

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
            from ..utils import dist_utils
            dist_utils.all_reduce_sum_(v)
            return [x / self.opt['world_size'] for x in v.tolist()]
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
            lr_data, gt_data, bi_data = ops.pingpong(lr_data), ops.pingpong(gt_data), ops.pingpong(bi_data)

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
        pair = getattr(self.net_D, 'forward_pair', None) if self.pair_pass else None
        if pair is not None:
            # real and fake pass as ONE pass over the stacked batch (per-half BatchNorm statistics)
            (pair_pred, pair_feats), d_out = pair(gt_data, hr_data, d_in)
            d_in.update(d_out)
            n_clip = pair_pred.shape[0] // 2
            real_pred, fake_pred = pair_pred[:n_clip], pair_pred[n_clip:]
            real_feats = [f[:n_clip] for f in pair_feats]
        else:
            (real_pred, real_feats), d_out = self.net_D(gt_data, d_in)
            d_in.update(d_out)
            (fake_pred, _), _ = self.net_D(hr_data, d_in)        # no input grad: == hr_data.detach()
            n_clip = real_pred.numel()

        scal = torch.zeros(16, dtype=torch.float32, device=self.device)   # every scalar of the step (+ the fault slots of G [14] and D [15])
        st_real, st_fake, st_g, losses = scal[0:3], scal[3:6], scal[6:9], scal[9:14]
        red = self.gan_crit[1]
        lsgan = self.gan_crit[0] == 'LSGAN'        # LSGANLoss (losses.py:17-28) instead of VanillaGANLoss
        gsc = (1.0 / n_clip) if red == 'mean' else 1.0
        g_pair = torch.empty_like(pair_pred) if pair is not None else None
        g_real = ops.bce_logits(real_pred, 1.0, st_real, 1.0 / n_clip, grad_scale=gsc,
                                dx_out=None if g_pair is None else g_pair[:n_clip], lsgan=lsgan)
        g_fake = ops.bce_logits(fake_pred, 0.0, st_fake, 1.0 / n_clip, grad_scale=gsc,
                                dx_out=None if g_pair is None else g_pair[n_clip:], lsgan=lsgan)

        update_policy = opt_tr['discriminator']['update_policy']
        if update_policy == 'adaptive':
            lreal, lfake = self._sync_scalars(st_real, st_fake)
            distance = lreal - lfake
            upd_D = distance < opt_tr['discriminator']['update_threshold']
        else:
            upd_D = True
        bucket_D = None
        if upd_D:
            self.cnt_upd_D += 1.0
            if g_pair is not None:
                tape_D.add_grad(pair_pred, g_pair)
            else:
                tape_D.add_grad(real_pred, g_real)
                tape_D.add_grad(fake_pred, g_fake)
            tape_D.backward()
            # D's gradient all-reduce runs on RCCL's stream while the D-independent generator
            # losses below are evaluated on the compute stream
            TG.stamp_fault(self.optim_D)
            bucket_D = self.start_grad_exchange(self.net_D)
        tape_D.nodes, tape_D.grads = [], {}

        # === generator ===
        # losses: pix warp pp feat fm
        if self.pix_crit is not None:
            w_ = opt_tr['pixel_crit'].get('weight', 1)
            tape_G.add_grad(hr_data, pointwise_loss(self.pix_crit, hr_data, gt_data, w_, losses[0:1]))
        if self.warp_crit is not None:
            lr_warp = TG.backward_warp(tape_G, out['lr_prev'], out['lr_flow'], need_dimg=False)
            w_ = opt_tr['warping_crit'].get('weight', 1)
            tape_G.add_grad(lr_warp, pointwise_loss(self.warp_crit, lr_warp, out['lr_curr'], w_,
                                                    losses[1:2]))
        if self.feat_crit is not None:     # perceptual loss (:226-241)
            hr_merge = TG.view(tape_G, hr_data, (-1, c, gt_h, gt_w))
            hr_feats = self.net_F(hr_merge, tape_G)
            gt_feats = self.net_F(gt_data.reshape(-1, c, gt_h, gt_w).contiguous())   # detached
            w_ = opt_tr['feature_crit'].get('weight', 1)
            for hf, gf in zip(hr_feats, gt_feats):
                if self.feat_crit[0] == 'CosineSimilarity':
                    sc = w_ / (hf.shape[0] * hf.shape[2] * hf.shape[3])     # 1 - mean(cos)
                    tape_G.add_grad(hf, ops.cosine_loss(hf, gf, losses[3:4], sc, grad_scale=sc))
                else:      # any element-wise criterion define_criterion accepts (optim/__init__.py:5-35)
                    tape_G.add_grad(hf, pointwise_loss(self.feat_crit, hf, gf, w_, losses[3:4]))
            del gt_feats
        if self.pp_crit is not None:
            te = opt_tr['tempo_extent']
            hr_fw = ops.time_gather(hr_data, list(range(te - 1)))
            hr_bw = ops.time_gather(hr_data, [2 * te - 2 - k for k in range(te - 1)])
            w_ = opt_tr['pingpong_crit'].get('weight', 1)
            g = pointwise_loss(self.pp_crit, hr_fw, hr_bw, w_, losses[2:3])
            tape_G.add_grad(hr_data, ops.pingpong_grad(g, te))      # +g | 0 | -flip(g)
        # D's update lands here: the third D pass sees the UPDATED, frozen critic
        # (:201-202 after :188)
        if upd_D:
            self.finish_grad_exchange(bucket_D, 'D')
            self.optim_D.step()
            if getattr(self.optim_D, 'fault_slot', None) is not None:
                scal[15:16].copy_(self.optim_D.fault_slot)      # read by took_back(): D's dropped update takes its count back
        for p in self.net_D.parameters():
            p.requires_grad = False
        d_in['tape'] = tape_G
        d_in['need_input_grad'] = True
        (fake_pred_G, fake_feats), _ = self.net_D(hr_data, d_in)
        if self.fm_crit is not None:       # feature matching (:255-271); real features are
            fo = opt_tr['feature_matching_crit']                     # those of the D pass above
            layer_norm = fo.get('layer_norm', [12.0, 14.0, 24.0, 100.0])
            w_ = fo.get('weight', 1)
            for i, (ff, rf) in enumerate(zip(fake_feats, real_feats)):
                tape_G.add_grad(ff, pointwise_loss(self.fm_crit, ff, rf, w_ / layer_norm[i],
                                                   losses[4:5]))
        gan_w = opt_tr['gan_crit'].get('weight', 1)
        tape_G.add_grad(fake_pred_G, ops.bce_logits(fake_pred_G, 1.0, st_g, 1.0 / n_clip,
                                                    grad_scale=gan_w * gsc, lsgan=lsgan))
        tape_G.backward()
        TG.stamp_fault(self.optim_G)                 # a chained-launch fault (any rank) turns the step into a no-op
        self.allreduce_grads(self.net_G, 'G')
        self.optim_G.step()
        if getattr(self.optim_G, 'fault_slot', None) is not None:
            scal[14:15].copy_(self.optim_G.fault_slot)
        ep, optim_G, optim_D = TG.chain_epoch(), self.optim_G, self.optim_D

        # === logging: ONE asynchronous read of all scalars (base_model: resolved when the log is looked at) ===
        cnt_upd, adaptive = self.cnt_upd_D, update_policy == 'adaptive'
        dist_val = distance if adaptive else None
        has = dict(pix=self.pix_crit is not None, warp=self.warp_crit is not None, feat=self.feat_crit is not None,
                   pp=self.pp_crit is not None, fm=self.fm_crit is not None)

        def build(sc_):
            def took_back():                        # the guard dropped these updates: their step counts do not advance
                optim_G.undo_step_count()
                if upd_D and sc_[15] != 0.0:
                    optim_D.undo_step_count()
            try:                                    # fail-safe of the chained launches: raises on EVERY rank, G's update was dropped
                dropped = TG.chain_check(sc_[14], counter=False, epoch=ep)
            except Exception:
                took_back()
                raise
            if dropped:                             # in flight behind an iteration that already raised
                took_back()
            sr, sf, sg, ls = sc_[0:3], sc_[3:6], sc_[6:9], sc_[9:14]
            d = OrderedDict()
            d['l_gan_D'] = (sr[0] + sf[0]) if upd_D else 0.0
            d['p_real_D'] = sr[1]
            d['p_fake_D'] = sf[1]
            if adaptive:
                d['distance'] = dist_val
                d['n_upd_D'] = cnt_upd
            if has['pix']:
                d['l_pix_G'] = ls[0]
            if has['warp']:
                d['l_warp_G'] = ls[1]
            if has['feat']:
                d['l_feat_G'] = ls[3]
            if has['pp']:
                d['l_pp_G'] = ls[2]
            if has['fm']:
                d['l_fm_G'] = ls[4]
            d['l_gan_G'] = gan_w * sg[0]
            d['p_fake_G'] = sg[1]
            return d
        self._set_pending_log(scal, build)

    def save(self, current_iter):
        self.sync_log()          # (a pending fault check must run before weights are written)
        self.save_network(self.net_G, 'G', current_iter)
        self.save_network(self.net_D, 'D', current_iter)
