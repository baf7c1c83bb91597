"""UnpairedLMDBDataset (codes/data/unpaired_lmdb_dataset.py) for BD training, split in two:

  * `draw_plan(item)`   consumes the Python / numpy random streams EXACTLY as the reference's
                        __getitem__ does (:42-83 frame selection incl. the "moving first frame"
                        synthetic motion, :95-109 crop, :112-129 flips / rotation) and returns
                        the geometry of the sample: which stored frames, which offsets, which
                        flips -- no pixel is touched;
  * `apply_plan_numpy`  executes a plan on host frames with the reference's numpy operations
                        (the CPU statement of the augmentation, pinned by tests/golden);
  * device execution    DeviceClipStore.gather (device_clip_store.py): the whole uint8 dataset
                        lives in HBM and ONE HIP kernel per batch crops / flips / rotates /
                        converts it into the (n, t, c, S, S) fp32 tensor prepare_training_data
                        wants.  Same plan, bit-identical result.

`__getitem__` keeps the reference's contract: {'gt': (t, c, S, S) float32 in [0, 1]}."""
import os.path as osp
import pickle
import random

import numpy as np
import torch

from .lmdb_io import LMDBReader, make_key, parse_lmdb_key


class ClipPlan:
    """Geometry of one training sample.  Output frame t takes stored frame keys[t], window
    rows [row0[t], row0[t] + size), columns [col0[t], col0[t] + size); then flip_axis
    (0 none | 2 rows | 3 columns, numpy axis numbering of the tchw stack), temporal reversal,
    and np.rot90(k) in the image plane."""
    __slots__ = ('keys', 'row0', 'col0', 'size', 'flip_axis', 'flip_t', 'rot_k')

    def __init__(self, keys, row0, col0, size, flip_axis, flip_t, rot_k):
        self.keys, self.row0, self.col0, self.size = keys, row0, col0, size
        self.flip_axis, self.flip_t, self.rot_k = flip_axis, flip_t, rot_k


class UnpairedLMDBDataset:
    def __init__(self, data_opt, **kwargs):
        for k, v in data_opt.items():
            setattr(self, k, v)
        for k, v in kwargs.items():               # crop_size (enlarged), tempo_extent, moving_*
            setattr(self, k, v)
        meta_path = osp.join(self.seq_dir, 'meta_info.pkl')
        if osp.isfile(meta_path):
            with open(meta_path, 'rb') as f:
                keys = pickle.load(f)['keys']
        else:                                     # the key list is in the database itself
            keys = [k.decode('ascii') for k in LMDBReader(self.seq_dir).keys()]
        self.keys = sorted(keys)
        if getattr(self, 'filter_file', None):
            with open(self.filter_file) as f:
                sel = {line.strip() for line in f}
            self.keys = [k for k in self.keys if parse_lmdb_key(k)[0] in sel]
        self.moving_first_frame = getattr(self, 'moving_first_frame', False)
        self.moving_factor = getattr(self, 'moving_factor', 1.0)
        self.data_type = getattr(self, 'data_type', 'rgb')
        self.env = None

    def __len__(self):
        return len(self.keys)

    # -- random geometry: the reference's draws, in the reference's order ---------------
    def draw_plan(self, item):
        key = self.keys[item]
        idx, (tot_frm, h, w), cur_frm = parse_lmdb_key(key)
        t = self.tempo_extent
        if self.moving_first_frame and (random.uniform(0, 1) > self.moving_factor):
            offsets = np.floor(np.random.uniform(-3.5, 4.5, size=(t, 2))).astype(np.int32)   # :50-52
            pos = np.cumsum(offsets, axis=0)
            min_pos = np.min(pos, axis=0)
            topleft = pos - min_pos
            rng = np.max(pos, axis=0) - min_pos
            c_h, c_w = h - int(rng[0]), w - int(rng[1])
            keys = [key] * t
            base_r, base_c = [int(v) for v in topleft[:, 0]], [int(v) for v in topleft[:, 1]]
        else:
            keys = []
            for i in range(cur_frm, cur_frm + t):
                j = 2 * tot_frm - i - 2 if i >= tot_frm else i           # reflect temporal padding
                keys.append(make_key(idx, tot_frm, h, w, j))
            c_h, c_w = h, w
            base_r, base_c = [0] * t, [0] * t
        csz = self.crop_size
        assert csz <= c_h and csz <= c_w, \
            f'The crop size is larger than the image size ({csz} vs. h{c_h}/w{c_w})'
        top = random.randint(0, c_h - csz)                                # :104-105
        left = random.randint(0, c_w - csz)
        axis = random.randint(1, 3)                                       # :115
        flip_t = random.randint(0, 1) < 1                                 # :120-122
        rot_k = random.randint(0, 3)                                      # :125
        return ClipPlan(keys, [r + top for r in base_r], [c + left for c in base_c], csz,
                        axis if axis > 1 else 0, flip_t, rot_k)

    # -- host execution (reference semantics) ----------------------------------------------
    def read_frame(self, key):
        if self.env is None:
            self.env = LMDBReader(self.seq_dir)
        _, (_, h, w), _ = parse_lmdb_key(key)
        c = 3 if self.data_type.lower() == 'rgb' else 1
        return np.frombuffer(self.env.get(key), dtype=np.uint8).reshape(h, w, c)

    def apply_plan_numpy(self, plan, read=None):
        read = read or self.read_frame
        s = plan.size
        frms = np.stack([read(k).transpose(2, 0, 1)[:, r:r + s, c:c + s]
                         for k, r, c in zip(plan.keys, plan.row0, plan.col0)])     # tchw uint8
        if plan.flip_axis:
            frms = np.flip(frms, plan.flip_axis)
        if plan.flip_t:
            frms = np.flip(frms, 0)
        return np.rot90(frms, plan.rot_k, (2, 3))

    def __getitem__(self, item):
        pats = self.apply_plan_numpy(self.draw_plan(item))
        return {'gt': torch.FloatTensor(np.ascontiguousarray(pats)) / 255.0}
