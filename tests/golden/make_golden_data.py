"""Golden vectors of the training-data augmentation (authoring container only).

Runs the REFERENCE UnpairedLMDBDataset (codes/data/unpaired_lmdb_dataset.py:36-129) on the
procedural frame set of data_fixture.py -- its lmdb environment replaced by an in-memory
object serving the same bytes -- under fixed Python / numpy seeds, and stores what its
__getitem__ returns (as the uint8 values the float tensor was made from: x * 255 is exact).
    python tests/golden/make_golden_data.py
"""
import os
import pickle
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import data_fixture as F  # noqa: E402


class _Txn:
    def __init__(self, frames):
        self.frames = frames

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def get(self, key):
        return self.frames[key.decode('ascii')].tobytes()


class _Env:
    def __init__(self, frames):
        self.frames = frames

    def begin(self, write=False):
        return _Txn(self.frames)


def main():
    _ref_import.import_reference()
    from data.unpaired_lmdb_dataset import UnpairedLMDBDataset
    frames = F.all_frames()
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, 'meta_info.pkl'), 'wb') as f:
        pickle.dump({'name': 'fixture', 'color': 'RGB', 'keys': list(frames.keys())}, f)
    out = {}
    for tag, (moving, factor, pseed, nseed) in F.CONFIGS.items():
        ds = UnpairedLMDBDataset({'seq_dir': tmp, 'filter_file': None, 'data_type': 'rgb'},
                                 crop_size=F.CROP, tempo_extent=F.TEMPO, moving_first_frame=moving,
                                 moving_factor=factor)
        ds.env = _Env(frames)
        random.seed(pseed)
        np.random.seed(nseed)
        items = [(7 * i + 3) % len(ds) for i in range(F.N_ITEMS)]
        got = []
        for it in items:
            x = ds[it]['gt'].numpy()
            u8 = np.round(x * 255.0).astype(np.uint8)
            assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), x)
            got.append(u8)
        out[tag + '_items'] = np.array(items, dtype=np.int64)
        out[tag + '_u8'] = np.stack(got)
    np.savez_compressed(os.path.join(HERE, 'data_aug.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
