"""Where does the HOST spend a training step?  cProfile over 10 steps (the GPU runs beside it)."""
import cProfile, pstats, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tecogan_pytorch_amd.models import define_model
crop = int(sys.argv[1]) if len(sys.argv) > 1 else 128
opt = {'scale': 4, 'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1, 'is_train': True,
       'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5}, 'train': {'crop_size': crop}},
       'model': {'name': 'TecoGAN', 'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10},
                 'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3}},
       'train': {'tempo_extent': 10, 'ckpt_dir': '/tmp', 'generator': {'lr': 5e-5, 'betas': [0.9, 0.999]},
                 'discriminator': {'update_policy': 'adaptive', 'update_threshold': 1e9, 'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                 'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'}, 'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                 'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'}, 'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
       'logger': {'decay': 0.99}}
torch.manual_seed(0)
m = define_model(opt)
data = {'gt': torch.rand(2, 10, 3, crop + 8, crop + 8).cuda()}
for _ in range(3):
    m.prepare_training_data(data); m.train()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    m.prepare_training_data(data); m.train()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats('tottime')
st.print_stats(28)
print(s.getvalue()[:6000])
