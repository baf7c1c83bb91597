"""Training-data front end: the reference's LMDB training sets
(codes/data/__init__.py:11-64, unpaired_lmdb_dataset.py) feeding prepare_training_data."""
import torch

from .device_clip_store import DeviceClipStore
from .lmdb_io import LMDBReader, LMDBWriter, make_key, parse_lmdb_key
from .unpaired_lmdb_dataset import ClipPlan, UnpairedLMDBDataset


class TrainSource:
    """What codes/data/__init__.py:create_dataloader(opt, 'train', ...) provides for BD
    training: an iterable of {'gt': (n, t, 3, S + 2b, S + 2b) fp32} batches, here already on
    the device.  Epoch order = DataLoader(shuffle=True, drop_last=True), or under
    torch.distributed the DistributedSampler rule (permutation seeded with seed + epoch, padded
    to a multiple of the world size, every world-th index starting at rank)."""

    def __init__(self, opt):
        data_opt = opt['dataset']['train']
        if opt['dataset']['degradation']['type'] != 'BD':
            raise NotImplementedError('LMDB front end: BD (unpaired GT) training sets; BI batches '
                                      'carry their own LR frames (paired_lmdb_dataset.py)')
        sigma = opt['dataset']['degradation'].get('sigma', 1.5)
        self.dataset = UnpairedLMDBDataset(
            data_opt, crop_size=data_opt['crop_size'] + 2 * int(sigma * 3.0),      # :33-34
            tempo_extent=opt['train']['tempo_extent'],
            moving_first_frame=opt['train'].get('moving_first_frame', False),
            moving_factor=opt['train'].get('moving_factor', 1.0))
        self.batch = data_opt['batch_size_per_gpu']
        self.rank, self.world = opt.get('rank', 0), opt.get('world_size', 1)
        self.dist = bool(opt.get('dist', False))
        self.seed = opt.get('manual_seed', 0)
        reader = LMDBReader(data_opt['seq_dir'])
        # every frame any sample can touch: the keys of the selected sequences
        seqs = {parse_lmdb_key(k)[0] for k in self.dataset.keys}
        frames = [k.decode('ascii') for k in reader.keys()]
        frames = [k for k in frames if k.count('_') >= 2 and parse_lmdb_key(k)[0] in seqs]
        self.store = DeviceClipStore.from_lmdb(reader, frames, opt.get('device', 'cuda'))
        reader.close()

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
                yield {'gt': self.store.gather(plans)}

    def __iter__(self):
        return self.epoch(0)
