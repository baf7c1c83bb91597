"""GPU: batches cut out of the HBM-resident uint8 training set by tg_gather_clips_u8 are
bit-identical to the reference's CPU samples (golden) and feed the training step."""
import os
import pickle
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import data_fixture as F
from tecogan_pytorch_amd.data import (DeviceClipStore, LMDBWriter, TrainSource, UnpairedLMDBDataset)


def _env(tmp_path):
    frames = F.all_frames()
    LMDBWriter(str(tmp_path)).write({k: v.tobytes() for k, v in frames.items()})
    with open(os.path.join(str(tmp_path), 'meta_info.pkl'), 'wb') as f:
        pickle.dump({'name': 'fixture', 'color': 'RGB', 'keys': list(frames.keys())}, f)
    return frames


@pytest.mark.parametrize('tag', list(F.CONFIGS))
def test_device_gather_equals_reference_samples(tmp_path, golden, tag):
    frames = _env(tmp_path)
    moving, factor, pseed, nseed = F.CONFIGS[tag]
    ds = UnpairedLMDBDataset({'seq_dir': str(tmp_path), 'filter_file': None, 'data_type': 'rgb'},
                             crop_size=F.CROP, tempo_extent=F.TEMPO, moving_first_frame=moving,
                             moving_factor=factor)
    store = DeviceClipStore.from_frames(frames.items())
    g = golden('data_aug')
    random.seed(pseed)
    np.random.seed(nseed)
    plans = [ds.draw_plan(int(it)) for it in g[tag + '_items']]
    out = store.gather(plans).cpu().numpy()                  # one launch for the 12 samples
    assert out.shape == (F.N_ITEMS, F.TEMPO, 3, F.CROP, F.CROP)
    assert np.array_equal(out, g[tag + '_u8'].astype(np.float32) / np.float32(255.0))
    # and against the host execution of the same plans (all 24 flip/rotation combinations occur)
    for p, o in zip(plans, out):
        host = np.ascontiguousarray(ds.apply_plan_numpy(p)).astype(np.float32) / np.float32(255.0)
        assert np.array_equal(o, host)


def test_chunked_upload_equals_frames(tmp_path, monkeypatch):
    """The store is filled through a pinned staging buffer (no second host copy of the data set):
    with a buffer just large enough for one frame every flush path runs; the device bytes must equal the frames."""
    frames = _env(tmp_path)
    monkeypatch.setattr(DeviceClipStore, 'STAGE_BYTES', 1)      # -> sized to the largest frame
    small = DeviceClipStore.from_frames(frames.items())
    monkeypatch.undo()
    big = DeviceClipStore.from_frames(frames.items())
    assert small.index == big.index and small.nbytes() == big.nbytes()
    host = small.buf.cpu().numpy()
    for k, v in frames.items():
        o, h, w = small.index[k]
        assert np.array_equal(host[o:o + h * w * 3], np.ascontiguousarray(v).reshape(-1)), k


def test_every_flip_rotation_combination(tmp_path):
    from tecogan_pytorch_amd.data import ClipPlan
    frames = _env(tmp_path)
    ds = UnpairedLMDBDataset({'seq_dir': str(tmp_path), 'filter_file': None, 'data_type': 'rgb'},
                             crop_size=F.CROP, tempo_extent=F.TEMPO)
    store = DeviceClipStore.from_frames(frames.items())
    keys = sorted(k for k in frames if k.startswith('walk_000'))[:F.TEMPO]
    plans = [ClipPlan(keys, [3, 4, 5, 6, 7], [9, 8, 7, 6, 5], F.CROP, ax, ft, k)
             for ax in (0, 2, 3) for ft in (False, True) for k in (0, 1, 2, 3)]
    out = store.gather(plans).cpu().numpy()
    for p, o in zip(plans, out):
        host = np.ascontiguousarray(ds.apply_plan_numpy(p)).astype(np.float32) / np.float32(255.0)
        assert np.array_equal(o, host), (p.flip_axis, p.flip_t, p.rot_k)
    with pytest.raises(ValueError):
        store.gather([ClipPlan(keys, [30] * 5, [0] * 5, F.CROP, 0, False, 0)])     # window outside


def test_train_source_feeds_the_training_step(tmp_path):
    """TrainSource = create_dataloader(opt, 'train') for BD: enlarged crop (S + 2*int(3 sigma)),
    drop_last batches, DistributedSampler sharding; its batches go straight into
    prepare_training_data + train()."""
    from tests.test_hip_train import make_opt
    from tecogan_pytorch_amd.models import define_model
    _env(tmp_path)
    opt = make_opt('FRVSR')
    opt['manual_seed'] = 3
    flt = tmp_path / 'sel.txt'
    flt.write_text('walk_000\n')                               # 8 frames of 40 x 48
    opt['dataset']['train'].update({'seq_dir': str(tmp_path), 'filter_file': str(flt), 'data_type': 'rgb',
                                    'crop_size': 32, 'batch_size_per_gpu': 2})
    opt['train']['tempo_extent'] = 4
    src = TrainSource(opt)
    assert len(src) == 8 // 2 and src.store.nbytes() >= 8 * 40 * 48 * 3
    batches = list(src.epoch(0))
    assert len(batches) == 4
    assert all(tuple(b['gt'].shape) == (2, 4, 3, 40, 40) and b['gt'].is_cuda for b in batches)
    assert float(batches[0]['gt'].min()) >= 0.0 and float(batches[0]['gt'].max()) <= 1.0
    m = define_model(opt)
    m.prepare_training_data(batches[0])
    assert tuple(m.lr_data.shape) == (2, 4, 3, 8, 8) and tuple(m.gt_data.shape) == (2, 4, 3, 32, 32)
    m.train()
    assert np.isfinite(m.log_dict['l_pix_G'])
    # two ranks see disjoint halves of the same permutation
    halves = []
    for rank in (0, 1):
        o = dict(opt, dist=True, rank=rank, world_size=2)
        s = TrainSource(o)
        n = len(s.dataset)
        g = torch.Generator()
        g.manual_seed(3)
        order = torch.randperm(n, generator=g).tolist()
        halves.append(order[rank:n:2])
        assert len(s) == (n // 2) // 2
    assert set(halves[0]).isdisjoint(halves[1]) and len(halves[0]) + len(halves[1]) == 8


# ------------------------------------------------------------------ paired (BI) sets, round 4
def test_paired_train_source_equals_reference_samples_and_feeds_bi_training(tmp_path, golden):
    """TrainSource for `degradation: BI` (codes/data/__init__.py:22-29): GT and LR clips gathered on the device
    from the two HBM-resident sets are bit-identical to the reference's PairedLMDBDataset samples (golden), and a
    batch goes straight into prepare_training_data + train() of the 2x BI model."""
    from tests.test_data_cpu import _make_paired_envs
    from tecogan_pytorch_amd.data import PairedLMDBDataset
    gt_dir, lr_dir = _make_paired_envs(tmp_path)
    g = golden('data_aug_paired')
    for tag in F.CONFIGS:
        moving, factor, pseed, nseed = F.CONFIGS[tag]
        ds = PairedLMDBDataset({'gt_seq_dir': gt_dir, 'lr_seq_dir': lr_dir, 'filter_file': None, 'data_type': 'rgb',
                                'gt_crop_size': F.PAIRED_GT_CROP},
                               scale=F.PAIRED_SCALE, tempo_extent=F.TEMPO, moving_first_frame=moving, moving_factor=factor)
        gt_store = DeviceClipStore.from_frames(F.all_frames().items())
        lr_store = DeviceClipStore.from_frames(F.all_lr_frames().items())
        random.seed(pseed)
        np.random.seed(nseed)
        plans = [ds.draw_plan(int(it)) for it in g[tag + '_items']]
        gt = gt_store.gather([p[0] for p in plans]).cpu().numpy()
        lr = lr_store.gather([p[1] for p in plans]).cpu().numpy()
        assert np.array_equal(gt, g[tag + '_gt_u8'].astype(np.float32) / np.float32(255.0)), tag
        assert np.array_equal(lr, g[tag + '_lr_u8'].astype(np.float32) / np.float32(255.0)), tag
    # the whole front end into a BI training step (2x, crop 16 -> LR 8)
    from tests.test_hip_train import make_opt
    from tecogan_pytorch_amd.models import define_model
    opt = make_opt('FRVSR')
    opt['scale'] = 2
    opt['dataset']['degradation'] = {'type': 'BI'}
    opt['dataset']['train'].update({'gt_seq_dir': gt_dir, 'lr_seq_dir': lr_dir, 'filter_file': None, 'data_type': 'rgb',
                                    'gt_crop_size': 16, 'batch_size_per_gpu': 2, 'name': 'REDS'})
    opt['train']['tempo_extent'] = 4
    src = TrainSource(opt)
    b = next(iter(src.epoch(0)))
    assert tuple(b['gt'].shape) == (2, 4, 3, 16, 16) and tuple(b['lr'].shape) == (2, 4, 3, 8, 8) and b['lr'].is_cuda
    m = define_model(opt)
    m.prepare_training_data(b)
    assert tuple(m.lr_data.shape) == (2, 4, 3, 8, 8) and tuple(m.gt_data.shape) == (2, 4, 3, 16, 16)
    m.train()
    assert np.isfinite(m.log_dict['l_pix_G'])
