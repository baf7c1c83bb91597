import sys, os, torch, numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests/golden'); sys.path.insert(0, ROOT + '/tests')
from oracle import tecogan_oracle as O
from test_r3_parity import bi_opt, bi_batch, GT, T, SCALE
from procedural_weights import generator_state_dict, discriminator_state_dict
from tecogan_pytorch_amd.models import define_model
chain = os.environ.get('CHAIN', '1') == '1'
m = define_model(bi_opt())
m.net_G.srnet.chain_body = chain
sdG0 = generator_state_dict(scale=SCALE, degradation='BI'); sdD0 = discriminator_state_dict(spatial_size=GT, scale=SCALE, degradation='BI')
m.net_G.load_state_dict(sdG0, strict=True); m.net_D.load_state_dict(sdD0, strict=True)
b = bi_batch(300)
m.prepare_training_data(b); m.train()
sdG = generator_state_dict(scale=SCALE, degradation='BI'); sdD = discriminator_state_dict(spatial_size=GT, scale=SCALE, degradation='BI')
log, gG, gD = O.vsrgan_train_step(sdG, sdD, {}, {}, {}, b['lr'], b['gt'], SCALE, 'BI', GT, T)
print('chain', chain, 'log', {k: (round(m.log_dict[k], 6), round(log[k], 6)) for k in log})
for k, p in m.net_G.named_parameters():
    if p.grad is None: continue
    a, r = p.grad.cpu().double(), gG[k].double()
    print(f'{k:40s} rel-l2 {((a - r).norm() / (r.norm() + 1e-30)).item():.3e}  norm {r.norm().item():.3e}')
