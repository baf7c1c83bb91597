"""PairedLMDBDataset (codes/data/paired_lmdb_dataset.py:12-166) for BI training: GT and LR frames come
from two LMDBs (`gt_seq_dir`, `lr_seq_dir`) whose keys match pair by pair.  Same split as the unpaired
mirror:

  * `draw_plan(item)`   consumes the Python / numpy random streams EXACTLY as the reference's __getitem__
                        does (:55-116 frame selection incl. the "moving first frame" motion drawn on the LR
                        grid, :131-150 crop on the LR grid, :153-166 flip / rotation -- no temporal flip in
                        the paired set) and returns the geometry of the sample as a (gt, lr) pair of
                        ClipPlans: the GT window is the LR window times the scale;
  * `apply_plan_numpy`  executes the pair on host frames with the reference's numpy operations;
  * device execution    two DeviceClipStores (GT, LR) and ONE gather kernel launch each per batch.

`__getitem__` keeps the reference's contract: {'gt': (t, c, S, S), 'lr': (t, c, S/s, S/s)} float32 in [0, 1]."""
import os.path as osp
import pickle
import random

import numpy as np
import torch

from .lmdb_io import LMDBReader, is_frame_key, make_key, parse_lmdb_key
from .unpaired_lmdb_dataset import ClipPlan


def _keys_of(seq_dir):
    meta_path = osp.join(seq_dir, 'meta_info.pkl')
    if osp.isfile(meta_path):
        with open(meta_path, 'rb') as f:
            return sorted(pickle.load(f)['keys'])
    # no meta_info.pkl (the reference requires it, paired_lmdb_dataset.py:24-27): the frame keys of the LMDB itself --
    # `{seq}_{n}x{h}x{w}_{i:04d}`, filtered as data.TrainSource._load filters them, so that bookkeeping entries are
    # not mistaken for frames -- and the reader is closed again (the dataset opens its own per worker)
    reader = LMDBReader(seq_dir)
    try:
        keys = [k.decode('ascii') for k in reader.keys()]
    finally:
        reader.close()
    return sorted(k for k in keys if is_frame_key(k))


class PairedLMDBDataset:
    def __init__(self, data_opt, **kwargs):
        for k, v in data_opt.items():
            setattr(self, k, v)
        for k, v in kwargs.items():               # scale, tempo_extent, moving_*
            setattr(self, k, v)
        gt_keys, lr_keys = _keys_of(self.gt_seq_dir), _keys_of(self.lr_seq_dir)
        self.check_info(gt_keys, lr_keys)
        self.gt_lr_keys = list(zip(gt_keys, lr_keys))
        if getattr(self, 'filter_file', None):
            with open(self.filter_file) as f:
                sel = {line.strip() for line in f}
            self.gt_lr_keys = [p for p in self.gt_lr_keys if parse_lmdb_key(p[0])[0] in sel]
        self.moving_first_frame = getattr(self, 'moving_first_frame', False)
        self.moving_factor = getattr(self, 'moving_factor', 1.0)
        self.data_type = getattr(self, 'data_type', 'rgb')
        self.gt_env = self.lr_env = None

    def check_info(self, gt_keys, lr_keys):
        """base_dataset.py:21-44."""
        if len(gt_keys) != len(lr_keys):
            raise ValueError(f'GT & LR contain different numbers of images ({len(gt_keys)}  vs. {len(lr_keys)})')
        s = self.scale
        for i, (gk, lk) in enumerate(zip(gt_keys, lr_keys)):
            gi, li = parse_lmdb_key(gk), parse_lmdb_key(lk)
            if gi[0] != li[0]:
                raise ValueError(f'video index mismatch ({gi[0]} vs. {li[0]} for the {i} key)')
            (gn, gh, gw), (ln, lh, lw) = gi[1], li[1]
            if gn != ln or gh != lh * s or gw != lw * s:
                raise ValueError(f'video size mismatch ({gi[1]} vs. {li[1]} for the {i} key)')
            if gi[2] != li[2]:
                raise ValueError(f'frame mismatch ({gi[2]} vs. {li[2]} for the {i} key)')

    def __len__(self):
        return len(self.gt_lr_keys)

    # -- random geometry: the reference's draws, in the reference's order ---------------
    def draw_plan(self, item):
        gt_key, lr_key = self.gt_lr_keys[item]
        idx, (tot_frm, gt_h, gt_w), cur_frm = parse_lmdb_key(gt_key)
        _, (_, lr_h, lr_w), _ = parse_lmdb_key(lr_key)
        s, t = self.scale, self.tempo_extent
        assert gt_h == lr_h * s and gt_w == lr_w * s
        if self.moving_first_frame and (random.uniform(0, 1) > self.moving_factor):
            offsets = np.floor(np.random.uniform(-1.5, 1.5, size=(t, 2))).astype(np.int32)     # :69-71
            pos = np.cumsum(offsets, axis=0)
            min_pos = np.min(pos, axis=0)
            topleft = pos - min_pos
            rng = np.max(pos, axis=0) - min_pos
            c_h, c_w = lr_h - int(rng[0]), lr_w - int(rng[1])
            gkeys, lkeys = [gt_key] * t, [lr_key] * t
            base_r, base_c = [int(v) for v in topleft[:, 0]], [int(v) for v in topleft[:, 1]]
        else:
            gkeys, lkeys = [], []
            for i in range(cur_frm, cur_frm + t):
                j = 2 * tot_frm - i - 2 if i >= tot_frm else i           # reflect temporal padding
                gkeys.append(make_key(idx, tot_frm, gt_h, gt_w, j))
                lkeys.append(make_key(idx, tot_frm, lr_h, lr_w, j))
            c_h, c_w = lr_h, lr_w
            base_r, base_c = [0] * t, [0] * t
        gt_csz = self.gt_crop_size
        lr_csz = gt_csz // s
        assert lr_csz <= c_h and lr_csz <= c_w, 'the crop size is larger than the image size'
        top = random.randint(0, c_h - lr_csz)                            # :140-141
        left = random.randint(0, c_w - lr_csz)
        axis = random.randint(1, 3)                                      # :156
        rot_k = random.randint(0, 3)                                     # :162
        fa = axis if axis > 1 else 0
        lr_plan = ClipPlan(lkeys, [r + top for r in base_r], [c + left for c in base_c], lr_csz, fa, False, rot_k)
        gt_plan = ClipPlan(gkeys, [(r + top) * s for r in base_r], [(c + left) * s for c in base_c], gt_csz, fa,
                           False, rot_k)
        return gt_plan, lr_plan

    # -- host execution (reference semantics) ----------------------------------------------
    def _read(self, which, key):
        env = getattr(self, which + '_env')
        if env is None:
            env = LMDBReader(getattr(self, which + '_seq_dir'))
            setattr(self, which + '_env', env)
        _, (_, h, w), _ = parse_lmdb_key(key)
        c = 3 if self.data_type.lower() == 'rgb' else 1
        return np.frombuffer(env.get(key), dtype=np.uint8).reshape(h, w, c)

    @staticmethod
    def _apply(plan, read):
        s = plan.size
        frms = np.stack([read(k).transpose(2, 0, 1)[:, r:r + s, c:c + s]
                         for k, r, c in zip(plan.keys, plan.row0, plan.col0)])     # tchw uint8
        if plan.flip_axis:
            frms = np.flip(frms, plan.flip_axis)
        return np.rot90(frms, plan.rot_k, (2, 3))

    def apply_plan_numpy(self, plans):
        gt_plan, lr_plan = plans
        return (self._apply(gt_plan, lambda k: self._read('gt', k)),
                self._apply(lr_plan, lambda k: self._read('lr', k)))

    def __getitem__(self, item):
        gt, lr = self.apply_plan_numpy(self.draw_plan(item))
        return {'gt': torch.FloatTensor(np.ascontiguousarray(gt)) / 255.0,
                'lr': torch.FloatTensor(np.ascontiguousarray(lr)) / 255.0}
