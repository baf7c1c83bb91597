"""The training set as ONE uint8 buffer in HBM, and batches cut out of it by a HIP kernel.

The reference decodes, crops, flips and converts every sample on CPU workers and ships fp32
over PCIe (unpaired_lmdb_dataset.py:36-93, DataLoader with pin_memory).  An MI355X has 288 GB:
REDS GT (240 x 100 frames of 720x1280) is 66 GB, VimeoTecoGAN less -- the decoded dataset fits
next to the model, so it is uploaded ONCE as raw bytes and a batch costs one kernel launch
(tg_gather_clips_u8: window, flips, rotation, u8 -> fp32 / 255 fused; 5 bytes moved per output
float) plus a few hundred bytes of geometry.  The random geometry is drawn on the host by
UnpairedLMDBDataset.draw_plan with the reference's own random streams."""
import numpy as np
import torch

from .. import _lib as L
from .lmdb_io import parse_lmdb_key


class DeviceClipStore:
    def __init__(self, device='cuda'):
        self.device = torch.device(device)
        self.index = {}            # key -> (byte offset, h, w)
        self.buf = None

    STAGE_BYTES = 64 << 20         # pinned staging buffer of the upload

    @classmethod
    def from_frames(cls, frames, device='cuda', channels=3):
        """frames: iterable of (key, uint8 array-like / bytes of h*w*channels) with LMDB key names,
        or a (keys, getter) pair for sources that can be walked twice (from_lmdb: no second host copy of
        the data set).  A plain iterable is materialised first (its keys size the device buffer before
        the first byte can be staged).  The device buffer is filled through one pinned staging buffer
        whose alignment gaps are zeroed."""
        self = cls(device)
        if isinstance(frames, tuple) and len(frames) == 2 and callable(frames[1]):
            keys, get = frames
        else:
            items = list(frames)
            keys, table = [k for k, _ in items], dict(items)
            get = table.__getitem__
        off = 0
        for key in keys:
            _, (_, h, w), _ = parse_lmdb_key(key)
            self.index[key] = (off, h, w)
            off += (h * w * channels + 15) // 16 * 16          # 16-byte aligned frames
        self.buf = torch.zeros(off, dtype=torch.uint8, device=self.device)
        pin = self.device.type == 'cuda'
        stage = torch.empty(max(cls.STAGE_BYTES, max((h * w * channels for _, h, w in self.index.values()),
                                                     default=0)), dtype=torch.uint8, pin_memory=pin)
        stage_np = stage.numpy()
        base, fill = 0, 0                                      # device offset of stage[0], bytes staged

        def flush():
            nonlocal base, fill
            if fill:
                self.buf[base:base + fill].copy_(stage[:fill])     # (synchronous: the stage is reused)
            base, fill = base + fill, 0
        for key in keys:
            o, h, w = self.index[key]
            data = get(key)
            arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else \
                np.ascontiguousarray(data).reshape(-1)
            if arr.size != h * w * channels:
                raise ValueError(f'{key}: {arr.size} bytes, expected {h}x{w}x{channels}')
            if o - base + arr.size > stage.numel():
                flush()
                base = o
            stage_np[o - base:o - base + arr.size] = arr
            fill = min(o - base + (arr.size + 15) // 16 * 16, stage.numel())
            stage_np[o - base + arr.size:fill] = 0             # the alignment gap: no stale bytes of an earlier flush
        flush()
        return self

    @classmethod
    def from_lmdb(cls, reader, keys, device='cuda'):
        return cls.from_frames((list(keys), reader.get), device)

    def nbytes(self):
        return 0 if self.buf is None else self.buf.numel()

    def gather(self, plans):
        """plans: list of n ClipPlan (same tempo extent and size) -> (n, t, 3, S, S) fp32 on the
        device, identical to stacking the reference's __getitem__ outputs."""
        n, t, s = len(plans), len(plans[0].keys), plans[0].size
        geo = np.empty((n, t, 4), dtype=np.int64)          # byte offset, width, row0, col0
        aug = np.empty((n, 3), dtype=np.int32)             # flip axis, temporal flip, rot k
        for i, p in enumerate(plans):
            if len(p.keys) != t or p.size != s:
                raise ValueError('gather: plans of one batch must share tempo extent and size')
            for j in range(t):
                off, h, w = self.index[p.keys[j]]
                if p.row0[j] < 0 or p.col0[j] < 0 or p.row0[j] + s > h or p.col0[j] + s > w:
                    raise ValueError(f'gather: window outside frame {p.keys[j]}')
                geo[i, j] = (off, w, p.row0[j], p.col0[j])
            aug[i] = (p.flip_axis, 1 if p.flip_t else 0, p.rot_k)
        geo_d = torch.from_numpy(geo).to(self.device)
        aug_d = torch.from_numpy(aug).to(self.device)
        out = torch.empty(n, t, 3, s, s, dtype=torch.float32, device=self.device)
        L.check(L.lib().tg_gather_clips_u8(self.buf.data_ptr(), geo_d.data_ptr(), aug_d.data_ptr(),
                                           out.data_ptr(), n, t, 3, s,
                                           torch.cuda.current_stream().cuda_stream),
                'tg_gather_clips_u8')
        return out
