"""BaseModel with the reference's wrapper contract (codes/models/base_model.py):
data preparation (on-device BD degradation), running log, checkpoint save/load,
temporal padding, cross-rank log reduce."""
from collections import OrderedDict
import os.path as osp

import torch

from .. import ops
from ..utils import dist_utils
from ..utils.data_utils import gaussian_kernel2d


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.scale = opt['scale']
        self.device = torch.device(opt['device'])
        self.blur_kernel = None
        self.dist = opt['dist']
        self.is_train = opt['is_train']
        if self.is_train:
            self.lr_data, self.gt_data = None, None
            self.ckpt_dir = opt['train'].get('ckpt_dir')
            self.log_decay = opt['logger'].get('decay', 0.99)
            self.log_dict = OrderedDict()
            self.running_log_dict = OrderedDict()

    # -- data ---------------------------------------------------------------
    def _blur_weight(self, sigma, c):
        """create_kernel (data_utils.py:11-27) as the (c, c, 9, 9) block-diagonal tensor."""
        if self.blur_kernel is None:
            k = torch.from_numpy(gaussian_kernel2d(sigma))
            full = torch.zeros(c, c, k.shape[0], k.shape[1])
            for i in range(c):
                full[i, i] = k
            self.blur_kernel = full.to(self.device)
        return self.blur_kernel

    def prepare_training_data(self, data):
        """base_model.py:42-85.  BD: LR = Gaussian blur + stride-s decimation of the
        bordered GT on the device (HIP kernel), GT border cropped."""
        deg = self.opt['dataset']['degradation']['type']
        if deg == 'BI':
            self.gt_data = data['gt'].to(self.device)
            self.lr_data = data['lr'].to(self.device)
            return
        scale = self.opt['scale']
        sigma = self.opt['dataset']['degradation'].get('sigma', 1.5)
        border = int(sigma * 3.0)
        gt = data['gt'].to(self.device, dtype=torch.float32)
        n, t, c, gh, gw = gt.shape
        lr_h, lr_w = (gh - 2 * border) // scale, (gw - 2 * border) // scale
        flat = gt.reshape(n * t, c, gh, gw).contiguous()
        lr = ops.downsample_bd(flat, gaussian_kernel2d(sigma), scale, pad=False)
        self.lr_data = lr.view(n, t, c, lr_h, lr_w)
        self.gt_data = flat[..., border:border + scale * lr_h, border:border + scale * lr_w] \
            .contiguous().view(n, t, c, scale * lr_h, scale * lr_w)

    def prepare_inference_data(self, data):
        """base_model.py:87-122: thwc -> tchw; BD without 'lr': blur+decimate with reflect pad."""
        deg = self.opt['dataset']['degradation']['type']
        if deg == 'BI' or 'lr' in data:
            lr = data['lr']
        else:
            scale = self.opt['scale']
            sigma = self.opt['dataset']['degradation'].get('sigma', 1.5)
            gt = data['gt']
            if gt.dtype == torch.uint8:         # raw bytes over PCIe, converted by the HIP kernel
                gt = ops.dequantize_u8_hwc(gt.to(self.device).contiguous())
            else:
                gt = gt.permute(0, 3, 1, 2).float().div(255.0).to(self.device).contiguous()
            lr = ops.downsample_bd(gt, gaussian_kernel2d(sigma), scale, pad=True).permute(0, 2, 3, 1)
        self.lr_data = lr.permute(0, 3, 1, 2)

    # -- bookkeeping --------------------------------------------------------
    def model_to_device(self, net):
        return net.to(self.device)

    def update_learning_rate(self):
        """base_model.py:138-143."""
        for name in ('sched_G', 'sched_D'):
            sched = getattr(self, name, None)
            if sched is not None:
                sched.step()

    def get_learning_rate(self):
        d = OrderedDict()
        if hasattr(self, 'optim_G'):
            d['lr_G'] = self.optim_G.param_groups[0]['lr']
        if hasattr(self, 'optim_D'):
            d['lr_D'] = self.optim_D.param_groups[0]['lr']
        return d

    def reduce_log(self):
        """base_model.py:156-168: mean over ranks, result on rank 0."""
        if self.dist:
            keys = list(self.log_dict.keys())
            vals = dist_utils.reduce_sum_to_master([self.log_dict[k] for k in keys],
                                                   device=self.device)
            if self.opt['rank'] == 0:
                vals = vals / self.opt['world_size']
            self.log_dict = OrderedDict((k, v.item()) for k, v in zip(keys, vals))

    def update_running_log(self):
        self.reduce_log()
        d = self.log_decay
        for k, cur in self.log_dict.items():
            run = self.running_log_dict.get(k)
            self.running_log_dict[k] = cur if run is None else d * run + (1.0 - d) * cur

    def get_current_log(self):
        return self.log_dict

    def get_running_log(self):
        return self.running_log_dict

    def get_format_msg(self, epoch, iter):
        msg = f'[epoch: {epoch} | iter: {iter}'
        for k, lr in self.get_learning_rate().items():
            msg += f' | {k}: {lr:.2e}'
        msg += '] '
        msg += ', '.join(f'{k}: {v:.3e}' for k, v in self.get_running_log().items())
        return msg

    @dist_utils.master_only
    def save_network(self, net, net_label, current_iter):
        torch.save(net.state_dict(), osp.join(self.ckpt_dir, f'{net_label}_iter{current_iter}.pth'))

    @dist_utils.master_only
    def save_training_state(self, current_iter):
        """Optimiser moments, schedule position and the adaptive-D counter next to the weight
        files of `save()`: everything `resume_training_state` needs to continue the run."""
        st = {'iter': int(current_iter)}
        for name in ('optim_G', 'optim_D'):
            if hasattr(self, name):
                st[name] = getattr(self, name).state_dict()
        for name in ('sched_G', 'sched_D'):
            if getattr(self, name, None) is not None:
                st[name] = getattr(self, name).last_epoch
        if hasattr(self, 'cnt_upd_D'):
            st['cnt_upd_D'] = self.cnt_upd_D
        torch.save(st, osp.join(self.ckpt_dir, f'state_iter{current_iter}.pth'))

    def resume_training_state(self, path):
        st = torch.load(path, map_location='cpu')
        for name in ('optim_G', 'optim_D'):
            if name in st and hasattr(self, name):
                getattr(self, name).load_state_dict(st[name])
        for name in ('sched_G', 'sched_D'):
            if name in st and getattr(self, name, None) is not None:
                getattr(self, name).last_epoch = st[name]
        if 'cnt_upd_D' in st:
            self.cnt_upd_D = st['cnt_upd_D']
        return st['iter']

    def load_network(self, net, load_path):
        net.load_state_dict(torch.load(load_path, map_location='cpu'))

    def pad_sequence(self, lr_data):
        """base_model.py:230-251: reflect / replicate temporal padding at the front."""
        mode = self.opt['test'].get('padding_mode', 'reflect')
        n_pad = self.opt['test'].get('num_pad_front', 0)
        assert n_pad < lr_data.size(0)
        if mode == 'reflect':
            lr_data = torch.cat([lr_data[1:1 + n_pad].flip(0), lr_data], dim=0)
        elif mode == 'replicate':
            lr_data = torch.cat([lr_data[:1].expand(n_pad, -1, -1, -1), lr_data], dim=0)
        else:
            raise ValueError(f'Unrecognized padding mode: {mode}')
        return lr_data, n_pad

    # -- gradient exchange (clip-level data parallel) ------------------------
    def allreduce_grads(self, net):
        """Mean of the gradients over ranks through ONE flat fp32 bucket (RCCL all-reduce
        over xGMI when the backend is nccl); payload 10.4 MB (G) / 3.3 MB (D)."""
        if not self.dist:
            return
        grads = [p.grad for p in net.parameters() if p.requires_grad and p.grad is not None]

        def scale(dst, src, a):
            dst.zero_()
            ops.axpy_(dst.view(-1), src.contiguous(), a)
        dist_utils.allreduce_mean_(grads, scale_fn=scale)
