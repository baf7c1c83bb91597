"""Training-data front end: the reference's LMDB training sets
(codes/data/__init__.py:11-64, unpaired_lmdb_dataset.py) feeding prepare_training_data."""
import torch

from .device_clip_store import DeviceClipStore
from .lmdb_io import LMDBReader, LMDBWriter, is_frame_key, make_key, parse_lmdb_key
from .paired_lmdb_dataset import PairedLMDBDataset
from .unpaired_lmdb_dataset import ClipPlan, UnpairedLMDBDataset


class TrainSource:
    """What codes/data/__init__.py:create_dataloader(opt, 'train', ...) provides: an iterable of
    {'gt': (n, t, 3, S + 2b, S + 2b) fp32} batches for BD training, {'gt': (n, t, 3, S, S), 'lr': (n, t, 3, S/s,
    S/s)} for BI (paired sets), here already on the device.  Epoch order = DataLoader(shuffle=True, drop_last=True), or under
    torch.distributed the DistributedSampler rule (permutation seeded with seed + epoch, padded
    to a multiple of the world size, every world-th index starting at rank)."""

    def __init__(self, opt):
        data_opt = opt['dataset']['train']
        self.batch = data_opt['batch_size_per_gpu']
        self.rank, self.world = opt.get('rank', 0), opt.get('world_size', 1)
        self.dist = bool(opt.get('dist', False))
        self.seed = opt.get('manual_seed', 0)
        deg = opt['dataset']['degradation']['type']
        dev = opt.get('device', 'cuda')
        common = dict(tempo_extent=opt['train']['tempo_extent'],
                      moving_first_frame=opt['train'].get('moving_first_frame', False),
                      moving_factor=opt['train'].get('moving_factor', 1.0))
        self.paired = deg == 'BI'
        if deg == 'BI':          # paired GT / LR sets (paired_lmdb_dataset.py; data/__init__.py:22-29)
            self.dataset = PairedLMDBDataset(data_opt, scale=opt['scale'], **common)
            seqs = {parse_lmdb_key(g)[0] for g, _ in self.dataset.gt_lr_keys}
            self.store = self._load(data_opt['gt_seq_dir'], seqs, dev)
            self.store_lr = self._load(data_opt['lr_seq_dir'], seqs, dev)
        elif deg == 'BD':        # unpaired GT, the crop enlarged by the blur border (:31-42)
            sigma = opt['dataset']['degradation'].get('sigma', 1.5)
            self.dataset = UnpairedLMDBDataset(data_opt, crop_size=data_opt['crop_size'] + 2 * int(sigma * 3.0), **common)
            seqs = {parse_lmdb_key(k)[0] for k in self.dataset.keys}
            self.store = self._load(data_opt['seq_dir'], seqs, dev)
        else:
            raise ValueError(f'Unrecognized degradation type: {deg}')

    @staticmethod
    def _load(seq_dir, seqs, device):
        """every frame any sample can touch -- the keys of the selected sequences -- into HBM"""
        reader = LMDBReader(seq_dir)
        frames = [k.decode('ascii') for k in reader.keys()]
        frames = [k for k in frames if is_frame_key(k) and parse_lmdb_key(k)[0] in seqs]
        store = DeviceClipStore.from_lmdb(reader, frames, device)
        reader.close()
        return store

    def __len__(self):
        per = len(self.dataset) if not self.dist else -(-len(self.dataset) // self.world)
        return per // self.batch

    def replay(self, epochs):
        """A resumed run: draw (and discard) the sample geometry of `epochs` whole epochs, so that the
        Python / numpy random streams the reference's __getitem__ consumes stand where an uninterrupted
        run would have them (the sampler order itself is a function of (seed, epoch) alone).  Exact for a
        resume in a NEW process seeded like the original one (main.setup) -- the random state is not
        part of the checkpoint."""
        for e in range(epochs):
            for _ in self.epoch(e, first_batch=1 << 60):
                pass

    def epoch(self, epoch=0, first_batch=0):
        """Batches of one epoch; `first_batch` > 0 skips that many: their sample geometry is still drawn
        (only the device gather is skipped), so the random streams stay aligned with an uninterrupted
        epoch.  Earlier EPOCHS of a resumed run are replayed with `replay()` (main --resume does)."""
        n = len(self.dataset)
        g = torch.Generator()
        g.manual_seed(self.seed + epoch)
        order = torch.randperm(n, generator=g).tolist()
        if self.dist:
            total = -(-n // self.world) * self.world
            order = (order + order[:total - n])[self.rank:total:self.world]
        for b in range(len(order) // self.batch):
            plans = [self.dataset.draw_plan(i) for i in order[b * self.batch:(b + 1) * self.batch]]
            if b >= first_batch:
                if self.paired:
                    yield {'gt': self.store.gather([p[0] for p in plans]),
                           'lr': self.store_lr.gather([p[1] for p in plans])}
                else:
                    yield {'gt': self.store.gather(plans)}

    def __iter__(self):
        return self.epoch(0)
