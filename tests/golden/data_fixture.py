"""Procedural stand-in for a decoded training set (shared by the fixture generator and the
tests): a few short sequences of smooth-ish random RGB frames under the reference's LMDB key
names.  Nothing is stored: frames are regenerated from the key's crc."""
import zlib

import numpy as np

SEQS = [('walk_000', 8, 40, 48), ('000_ride', 6, 36, 52)]          # (name, frames, h, w)
TEMPO, CROP = 5, 24


def frame(key, h, w):
    rs = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    return rs.randint(0, 256, (h, w, 3), dtype=np.uint8)


def all_frames():
    out = {}
    for name, n, h, w in SEQS:
        for i in range(n):
            key = f'{name}_{n}x{h}x{w}_{i:04d}'
            out[key] = frame(key, h, w)
    return out


CONFIGS = {            # tag -> (moving_first_frame, moving_factor, python seed, numpy seed)
    'plain': (False, 1.0, 11, 12),
    'moving': (True, 0.2, 21, 22),
    'mixed': (True, 0.6, 31, 32),
}
N_ITEMS = 12

# paired (BI) sets: LR frames of half the size under the matching keys (make_golden_data_paired.py)
PAIRED_SCALE, PAIRED_GT_CROP = 2, 16


def all_lr_frames():
    out = {}
    for name, n, h, w in SEQS:
        for i in range(n):
            key = f'{name}_{n}x{h // PAIRED_SCALE}x{w // PAIRED_SCALE}_{i:04d}'
            out[key] = frame(key, h // PAIRED_SCALE, w // PAIRED_SCALE)
    return out
