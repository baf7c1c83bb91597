"""CPU: the training-data front end -- LMDB file format reader /
writer, the reference's key format, and the augmentation geometry + host execution against
golden vectors produced by the REFERENCE UnpairedLMDBDataset (tests/golden/make_golden_data.py;
same Python / numpy seeds => identical samples, bit for bit)."""
import os
import pickle
import random
import struct

import numpy as np
import pytest

import data_fixture as F
from tecogan_pytorch_amd.data import (LMDBReader, LMDBWriter, UnpairedLMDBDataset, make_key,
                                      parse_lmdb_key)


def test_key_format_round_trip():
    k = make_key('my_seq_01', 100, 720, 1280, 7)
    assert k == 'my_seq_01_100x720x1280_0007'                 # scripts/create_lmdb.py:57
    assert parse_lmdb_key(k) == ('my_seq_01', (100, 720, 1280), 7)


def test_lmdb_round_trip_inline_overflow_and_multi_level(tmp_path):
    rs = np.random.RandomState(5)
    items = {}
    for s in range(50):                                        # 1500 small (inline) values
        for i in range(30):
            items[make_key(f'small_{s:02d}', 30, 10, 12, i)] = rs.randint(0, 256, 360, dtype=np.uint8).tobytes()
    for i in range(20):                                        # overflow runs of 4 pages each
        items[make_key('big', 20, 64, 64, i)] = rs.randint(0, 256, 64 * 64 * 3, dtype=np.uint8).tobytes()
    items['edge_exact'] = bytes(2038 - 8 - len('edge_exact'))  # largest inline node
    items['edge_over'] = bytes(2038 - 8 - len('edge_over') + 1)  # first size that must overflow
    items['empty'] = b''
    n = LMDBWriter(str(tmp_path)).write(items)
    r = LMDBReader(str(tmp_path))
    assert n == len(items) == len(r)
    assert r.meta['depth'] >= 3 and r.psize == 4096
    assert r.keys() == sorted(k.encode() for k in items)
    for k, v in items.items():
        assert bytes(r.get(k)) == v, k
    assert r.get('absent') is None and r.get('big_20x64x64_9999') is None and r.get('') is None
    # file-level invariants of the format: meta magic, page numbers stored in every header
    raw = open(os.path.join(str(tmp_path), 'data.mdb'), 'rb').read()
    assert struct.unpack_from('<I', raw, 16)[0] == 0xBEEFC0DE
    assert struct.unpack_from('<I', raw, 4096 + 16)[0] == 0xBEEFC0DE
    assert struct.unpack_from('<Q', raw, 2 * 4096)[0] == 2
    assert len(raw) % 4096 == 0 and len(raw) // 4096 == r.meta['last_pg'] + 1
    r.close()


def test_reader_rejects_garbage(tmp_path):
    p = tmp_path / 'data.mdb'
    p.write_bytes(b'\0' * 8192)
    with pytest.raises(ValueError):
        LMDBReader(str(tmp_path))


def _make_env(tmp_path, with_meta=True):
    frames = F.all_frames()
    LMDBWriter(str(tmp_path)).write({k: v.tobytes() for k, v in frames.items()})
    if with_meta:
        with open(os.path.join(str(tmp_path), 'meta_info.pkl'), 'wb') as f:
            pickle.dump({'name': 'fixture', 'color': 'RGB', 'keys': list(frames.keys())}, f)
    return frames


@pytest.mark.parametrize('tag', list(F.CONFIGS))
@pytest.mark.parametrize('with_meta', [True, False])
def test_dataset_samples_equal_the_reference(tmp_path, golden, tag, with_meta):
    """Frames served from an LMDB file written here; random geometry drawn with the same
    seeds as the reference run => the same samples (incl. reflect temporal padding at the end
    of a sequence and the "moving first frame" windows)."""
    _make_env(tmp_path, with_meta)
    moving, factor, pseed, nseed = F.CONFIGS[tag]
    ds = UnpairedLMDBDataset({'seq_dir': str(tmp_path), 'filter_file': None, 'data_type': 'rgb'},
                             crop_size=F.CROP, tempo_extent=F.TEMPO, moving_first_frame=moving,
                             moving_factor=factor)
    g = golden('data_aug')
    assert len(ds) == sum(n for _, n, _, _ in F.SEQS)
    random.seed(pseed)
    np.random.seed(nseed)
    for it, ref in zip(g[tag + '_items'], g[tag + '_u8']):
        x = ds[int(it)]['gt']
        assert x.dtype.is_floating_point and tuple(x.shape) == (F.TEMPO, 3, F.CROP, F.CROP)
        assert np.array_equal(x.numpy(), ref.astype(np.float32) / np.float32(255.0)), (tag, int(it))


def test_filter_file_and_plan_geometry(tmp_path):
    _make_env(tmp_path)
    flt = tmp_path / 'sel.txt'
    flt.write_text('000_ride\n')
    ds = UnpairedLMDBDataset({'seq_dir': str(tmp_path), 'filter_file': str(flt), 'data_type': 'rgb'},
                             crop_size=F.CROP, tempo_extent=F.TEMPO, moving_first_frame=True,
                             moving_factor=0.0)
    assert len(ds) == 6 and all(k.startswith('000_ride') for k in ds.keys)
    random.seed(1)
    np.random.seed(2)
    for it in range(len(ds)):
        p = ds.draw_plan(it)
        assert len(set(p.keys)) == 1                        # moving first frame: one stored frame
        _, (_, h, w), _ = parse_lmdb_key(p.keys[0])
        assert all(0 <= r and r + F.CROP <= h for r in p.row0)
        assert all(0 <= c and c + F.CROP <= w for c in p.col0)
        assert p.flip_axis in (0, 2, 3) and p.rot_k in (0, 1, 2, 3)
    with pytest.raises(AssertionError):                    # crop larger than the frame
        UnpairedLMDBDataset({'seq_dir': str(tmp_path), 'filter_file': None, 'data_type': 'rgb'},
                            crop_size=64, tempo_extent=F.TEMPO).draw_plan(0)


def test_manual_seed_reproduces_augmentation(tmp_path):
    """main.seed_everything (setup_random_seed, base_utils.py:78-83) seeds Python's `random` too: the
    crop / flip / rotation geometry of the LMDB data set is drawn from it, so two runs under one
    manual_seed must draw identical plans (and a different seed different ones)."""
    from tecogan_pytorch_amd.main import seed_everything
    frames = F.all_frames()
    LMDBWriter(str(tmp_path)).write({k: v.tobytes() for k, v in frames.items()})
    with open(os.path.join(str(tmp_path), 'meta_info.pkl'), 'wb') as f:
        pickle.dump({'name': 'fixture', 'color': 'RGB', 'keys': list(frames.keys())}, f)
    ds = UnpairedLMDBDataset({'seq_dir': str(tmp_path), 'filter_file': None, 'data_type': 'rgb'},
                             crop_size=F.CROP, tempo_extent=F.TEMPO, moving_first_frame=True, moving_factor=0.7)

    def draw(seed):
        seed_everything(seed)
        return [(p.keys, p.row0, p.col0, p.flip_axis, p.flip_t, p.rot_k)
                for p in (ds.draw_plan(i % len(ds)) for i in range(16))]
    a, b, c = draw(2021), draw(2021), draw(2022)
    assert a == b and a != c


# ------------------------------------------------------------------ paired (BI) sets, round 4
def _make_paired_envs(tmp_path):
    from tecogan_pytorch_amd.data import LMDBWriter as W
    dirs = []
    for name, frames in (('gt', F.all_frames()), ('lr', F.all_lr_frames())):
        d = os.path.join(str(tmp_path), name)
        os.makedirs(d)
        W(d).write({k: v.tobytes() for k, v in frames.items()})
        with open(os.path.join(d, 'meta_info.pkl'), 'wb') as f:
            pickle.dump({'name': 'fixture', 'color': 'RGB', 'keys': list(frames.keys())}, f)
        dirs.append(d)
    return dirs


@pytest.mark.parametrize('tag', list(F.CONFIGS))
def test_paired_dataset_samples_equal_the_reference(tmp_path, golden, tag):
    """PairedLMDBDataset (codes/data/paired_lmdb_dataset.py:12-166, BI training): GT and LR windows of a
    sample cut from two LMDBs with the reference's random draws -- identical to the reference's __getitem__
    under the same seeds (tests/golden/make_golden_data_paired.py), incl. the moving-first-frame motion on
    the LR grid and the reflect temporal padding."""
    from tecogan_pytorch_amd.data import PairedLMDBDataset
    gt_dir, lr_dir = _make_paired_envs(tmp_path)
    moving, factor, pseed, nseed = F.CONFIGS[tag]
    ds = PairedLMDBDataset({'gt_seq_dir': gt_dir, 'lr_seq_dir': lr_dir, 'filter_file': None, 'data_type': 'rgb',
                            'gt_crop_size': F.PAIRED_GT_CROP},
                           scale=F.PAIRED_SCALE, tempo_extent=F.TEMPO, moving_first_frame=moving, moving_factor=factor)
    g = golden('data_aug_paired')
    assert len(ds) == sum(n for _, n, _, _ in F.SEQS)
    random.seed(pseed)
    np.random.seed(nseed)
    lc = F.PAIRED_GT_CROP // F.PAIRED_SCALE
    for it, rg, rl in zip(g[tag + '_items'], g[tag + '_gt_u8'], g[tag + '_lr_u8']):
        s = ds[int(it)]
        assert tuple(s['gt'].shape) == (F.TEMPO, 3, F.PAIRED_GT_CROP, F.PAIRED_GT_CROP) and tuple(s['lr'].shape) == (F.TEMPO, 3, lc, lc)
        assert np.array_equal(s['gt'].numpy(), rg.astype(np.float32) / np.float32(255.0)), (tag, int(it))
        assert np.array_equal(s['lr'].numpy(), rl.astype(np.float32) / np.float32(255.0)), (tag, int(it))


def test_paired_dataset_refuses_mismatched_sets(tmp_path):
    from tecogan_pytorch_amd.data import PairedLMDBDataset
    gt_dir, lr_dir = _make_paired_envs(tmp_path)
    opt = {'gt_seq_dir': gt_dir, 'lr_seq_dir': lr_dir, 'filter_file': None, 'data_type': 'rgb', 'gt_crop_size': 16}
    with pytest.raises(ValueError):            # the sets are 2x apart
        PairedLMDBDataset(opt, scale=4, tempo_extent=3)


def test_paired_dataset_without_meta_info_lists_frame_keys_only(tmp_path):
    """without meta_info.pkl the key list comes from the LMDB itself -- frame keys only (a bookkeeping
    entry such as `__len__` is not a frame), the listing reader is closed, and the pairs match the meta_info form."""
    from tecogan_pytorch_amd.data import LMDBWriter as W, PairedLMDBDataset
    from tecogan_pytorch_amd.data import paired_lmdb_dataset as P
    dirs = []
    for name, frames in (('gt', F.all_frames()), ('lr', F.all_lr_frames())):
        d = os.path.join(str(tmp_path), name)
        os.makedirs(d)
        items = {k: v.tobytes() for k, v in frames.items()}
        items['__len__'] = b'%d' % len(frames)
        W(d).write(items)
        dirs.append(d)
    closed = []
    real_close = P.LMDBReader.close

    def close(self):
        closed.append(self)
        return real_close(self)
    P.LMDBReader.close = close
    try:
        keys = P._keys_of(dirs[0])
    finally:
        P.LMDBReader.close = real_close
    assert len(closed) == 1 and keys == sorted(F.all_frames().keys())
    ds = PairedLMDBDataset({'gt_seq_dir': dirs[0], 'lr_seq_dir': dirs[1], 'filter_file': None, 'data_type': 'rgb',
                            'gt_crop_size': F.PAIRED_GT_CROP}, scale=F.PAIRED_SCALE, tempo_extent=F.TEMPO)
    assert len(ds) == sum(n for _, n, _, _ in F.SEQS)
