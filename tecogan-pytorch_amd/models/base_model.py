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
        self._log_dict, self._pending_log, self._log_queue, self._log_pinned = OrderedDict(), None, [], []
        self.log_decay = 0.99
        if self.is_train:
            self.lr_data, self.gt_data = None, None
            self.ckpt_dir = opt['train'].get('ckpt_dir')
            self.log_decay = opt['logger'].get('decay', 0.99)
            self.running_log_dict = OrderedDict()

    def agreement_vector(self):
        """What every rank of a data-parallel run must have in common for the step's exchange pattern to match: the
        library's ABI / build number, whether the chained body launches are available on this device for the training
        frames (they decide whether a fault slot is stamped), the critic's pair pass (19 vs 27 collectives per step),
        the flat gradient buckets' sizes, the gradient transport."""
        import os
        from .. import _lib as L
        from . import train_graph as TG
        crop = int(self.opt.get('dataset', {}).get('train', {}).get('crop_size', 0) or 0)
        lr = crop // max(1, self.scale)
        chain = -1
        if self.device.type == 'cuda' and lr > 0:
            chain = 0 if TG._ChainState.disabled else int(TG._ChainState.parts(2, lr, lr))
        flat = [int(o.flat_grad.numel()) if getattr(o, 'flat_grad', None) is not None else -1
                for o in (getattr(self, 'optim_G', None), getattr(self, 'optim_D', None)) if o is not None]
        return [int(L.lib().tg_version()), chain, int(bool(getattr(self, 'pair_pass', False))),
                int(os.environ.get('TECOGAN_COMM', '') == 'c_abi')] + flat

    def check_ranks_agree(self):
        """One small all-gather at construction (dist_utils.assert_ranks_agree): a mixed build cannot deadlock the first
        gradient bucket."""
        if self.dist:
            dist_utils.assert_ranks_agree(self.agreement_vector(), 'library version / chained-launch capability / '
                                          'pair_pass / transport / gradient-bucket sizes', device=self.device)

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
        """base_model.py:124-136.  Under DDP the reference's wrapper broadcasts rank 0's
        parameters and buffers at construction; main.setup seeds every rank differently
        (base_utils.py:46), so without this the replicas would start from different weights
        and never converge to one model."""
        net = net.to(self.device)
        if self.dist:
            self.sync_module_states(net)
        return net

    def sync_module_states(self, net):
        for prm in net.parameters():
            dist_utils.broadcast_(prm.data, 0)
            ops.bump_version(prm)
        for buf in net.buffers():
            dist_utils.broadcast_(buf, 0)

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

    # -- the iteration's scalars, read asynchronously ----------------------------------------------------------------
    # train() ends with ONE asynchronous device-to-host copy of all its scalars into pinned memory and an event;
    # `log_dict` (and everything that reads it) waits for that event only when somebody looks.  The reference reads its
    # losses with .item() inside the iteration (vsrgan_model.py:166-173, 206-286): the host then cannot enqueue the next
    # iteration before the GPU has finished this one, and the first ~40 launches of every iteration run behind an empty
    # queue (0.4 ms of a 10.4 ms step at the REDS crop, tools/prof_train_gaps.sh).  The fail-safe check of the chained
    # launches rides on the same scalars: a fault of iteration k raises on the first look at its log, at the latest at
    # the end of iteration k + 1 (whose update the device-side guard drops as well) and before every save().
    @property
    def log_dict(self):
        self._drain_log_queue()
        self._materialize_log()
        return self._log_dict

    @log_dict.setter
    def log_dict(self, value):
        self._pending_log = None
        self._log_dict = value

    def _pinned_like(self, scal):
        for i, t in enumerate(self._log_pinned):
            if t.numel() == scal.numel():
                return self._log_pinned.pop(i)
        return torch.empty(scal.numel(), dtype=torch.float32, pin_memory=True)

    def _set_pending_log(self, scal, build):
        """scal: device tensor of the iteration's scalars; build(list of floats) -> OrderedDict (may raise)."""
        prev, self._pending_log = self._pending_log, None
        if prev is not None:          # nobody looked at the previous iteration: its checks still run (its event is long
            self._resolve(prev)       # complete).  A fault raises HERE, before this iteration's scalars are recorded.
        host = self._pinned_like(scal)
        host.copy_(scal, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending_log = (ev, host, build)

    def _resolve(self, pending):
        ev, host, build = pending
        ev.synchronize()
        vals = host.tolist()
        self._log_pinned.append(host)
        return build(vals)

    def _materialize_log(self):
        p = self._pending_log
        if p is not None:
            self._pending_log = None
            self._log_dict = self._resolve(p)

    def sync_log(self):
        """Wait for the last iteration's scalars (and its fault check) now."""
        return self.log_dict

    def _drain_log_queue(self, ready_only=False):
        """ready_only: fold in the queued iterations whose scalars have ARRIVED (no waiting) -- called every iteration, so
        a fault is reported within an iteration or two although nobody waits for the log."""
        q = self._log_queue
        d = self.log_decay
        # one entry at a time, each removed only once it has been looked at: a fault raised by an entry's checks
        # (chained-launch fail-safe) then leaves the LATER entries -- and their pinned buffers -- queued for the next
        # call instead of dropping them; the faulty entry itself is consumed (it has been reported)
        while q and (not ready_only or q[0][0].query()):
            p = q.pop(0)
            self._log_dict = cur_log = self._resolve(p)
            for k, cur in cur_log.items():
                run = self.running_log_dict.get(k)
                self.running_log_dict[k] = cur if run is None else d * run + (1.0 - d) * cur

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
        if not self.dist and self._pending_log is not None:
            # single process: the running mean is folded in when somebody asks for it (get_running_log / the log line)
            self._log_queue.append(self._pending_log)
            self._pending_log = None
            self._drain_log_queue(ready_only=len(self._log_queue) < 64)
            return
        self.reduce_log()
        d = self.log_decay
        for k, cur in self.log_dict.items():
            run = self.running_log_dict.get(k)
            self.running_log_dict[k] = cur if run is None else d * run + (1.0 - d) * cur

    def get_current_log(self):
        return self.log_dict

    def get_running_log(self):
        self._drain_log_queue()
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
        # parameters are views of the optimiser's flat buffer: clone so that each entry is
        # serialised as its own storage, exactly like a reference checkpoint (base_model.py:213-218)
        sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
        torch.save(sd, osp.join(self.ckpt_dir, f'{net_label}_iter{current_iter}.pth'))

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
        """Optimiser moments + step counts, schedule positions (and the learning rate they imply),
        the adaptive-D counter.  Weights are loaded separately (load_network); betas / eps /
        weight decay follow the current configuration.  main.train(start_iter=...) wires both."""
        st = torch.load(path, map_location='cpu')
        for name in ('optim_G', 'optim_D'):
            if name in st and hasattr(self, name):
                getattr(self, name).load_state_dict(st[name])
        for name in ('sched_G', 'sched_D'):
            if name in st and getattr(self, name, None) is not None:
                sched = getattr(self, name)
                sched.last_epoch = st[name]
                sched.optimizer.param_groups[0]['lr'] = sched.lr_at(sched.last_epoch) if st[name] else sched.base_lr
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
    def _optim_of(self, net):
        for name_n, name_o in (('net_G', 'optim_G'), ('net_D', 'optim_D')):
            if getattr(self, name_n, None) is net:
                return getattr(self, name_o, None)
        return None

    def start_grad_exchange(self, net):
        """Launch the mean-over-ranks of one network's gradients: ONE flat fp32 all-reduce
        (RCCL over xGMI when the backend is nccl; 10.4 MB G / 3.3 MB D -- latency bound, so a
        single collective replaces DDP's many buckets), asynchronous on RCCL's own stream.
        Returns a handle for finish_grad_exchange, or None outside data-parallel runs.  The
        optimiser's flat gradient buffer is reduced in place (no concatenation / copy back)."""
        if not self.dist:
            return None
        optim = self._optim_of(net)
        if optim is not None and optim._is_flat():
            bucket = dist_utils.GradBucket([optim.flat_grad], flat=optim.flat_grad)
        else:
            bucket = dist_utils.GradBucket(
                [p.grad for p in net.parameters() if p.requires_grad and p.grad is not None])
        return bucket.start()

    def finish_grad_exchange(self, bucket, tag=None):
        """Order the compute stream after the collective and turn the sum into the mean.  When
        `self.exchange_timing` is a dict (bench.py --gpus N sets it), two events bracket the wait on
        the COMPUTE stream: their distance is how long the step actually stalled for the exchange
        (0 when the collective finished under the work that was enqueued in between)."""
        if bucket is None:
            return
        timing = getattr(self, 'exchange_timing', None)
        ev = None
        if timing is not None and tag is not None and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()

        def mean(dst, src, world):                # dst = (src or dst) / world, IEEE division as DDP's
            ops.div_scalar_(dst.view(-1), world, None if src is None else src.contiguous())
        bucket.finish(scale_fn=mean)
        if ev is not None:
            ev[1].record()
            timing.setdefault(tag, []).append(ev)

    def allreduce_grads(self, net, tag=None):
        """Blocking form (reference: DDP's backward hook, base_model.py:130-136)."""
        self.finish_grad_exchange(self.start_grad_exchange(net), tag)
