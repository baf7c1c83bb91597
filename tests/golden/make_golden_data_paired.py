"""Golden vectors of the PAIRED (BI) training-data augmentation (authoring container only).

Runs the REFERENCE PairedLMDBDataset (codes/data/paired_lmdb_dataset.py:12-166) at scale 2 on the procedural
frame set of data_fixture.py -- GT frames as they are, LR frames = procedural frames of half the size under
the matching keys; both lmdb environments replaced by in-memory objects serving the same bytes -- under fixed
Python / numpy seeds, and stores what its __getitem__ returns (as uint8: x * 255 is exact).
    python tests/golden/make_golden_data_paired.py"""
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
from make_golden_data import _Env  # noqa: E402


def main():
    _ref_import.import_reference()
    from data.paired_lmdb_dataset import PairedLMDBDataset
    gt_frames, lr_frames = F.all_frames(), F.all_lr_frames()
    dirs = []
    for frames in (gt_frames, lr_frames):
        tmp = tempfile.mkdtemp()
        with open(os.path.join(tmp, 'meta_info.pkl'), 'wb') as f:
            pickle.dump({'name': 'fixture', 'color': 'RGB', 'keys': list(frames.keys())}, f)
        dirs.append(tmp)
    out = {}
    for tag, (moving, factor, pseed, nseed) in F.CONFIGS.items():
        ds = PairedLMDBDataset({'gt_seq_dir': dirs[0], 'lr_seq_dir': dirs[1], 'filter_file': None, 'data_type': 'rgb',
                                'gt_crop_size': F.PAIRED_GT_CROP},
                               scale=F.PAIRED_SCALE, tempo_extent=F.TEMPO, moving_first_frame=moving, moving_factor=factor)
        ds.gt_env, ds.lr_env = _Env(gt_frames), _Env(lr_frames)
        random.seed(pseed)
        np.random.seed(nseed)
        items = [(7 * i + 3) % len(ds) for i in range(F.N_ITEMS)]
        got = {'gt': [], 'lr': []}
        for it in items:
            s = ds[it]
            for k in ('gt', 'lr'):
                x = s[k].numpy()
                u8 = np.round(x * 255.0).astype(np.uint8)
                assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), x)
                got[k].append(u8)
        out[tag + '_items'] = np.array(items, dtype=np.int64)
        out[tag + '_gt_u8'] = np.stack(got['gt'])
        out[tag + '_lr_u8'] = np.stack(got['lr'])
    np.savez_compressed(os.path.join(HERE, 'data_aug_paired.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
