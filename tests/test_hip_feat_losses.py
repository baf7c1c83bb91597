"""GPU: perceptual (VGG19 cosine) and feature-matching losses on the HIP path against the
reference's own VSRGANModel.train() (tests/golden/train_feat.npz) and the CPU oracle.
Tolerances as in test_hip_train.py; the cosine gradient at the 2x2 relu5_4 map is divided by
|a||b| ~ 1e-1..1e1, nothing ill-conditioned."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tecogan_oracle as O
from procedural_weights import (generator_state_dict, discriminator_state_dict, vgg19_state_dict)
from procedural_weights import smooth_clip

CROP, T, N, SCALE = 32, 4, 2, 4
WATCH_G = ['fnet.encoder1.0.weight', 'fnet.decoder1.2.bias', 'fnet.flow.2.weight',
           'srnet.conv_in.0.weight', 'srnet.resblocks.4.conv.2.weight', 'srnet.conv_up.2.weight',
           'srnet.conv_out.bias']
WATCH_D = ['conv_in.0.weight', 'discriminator_block.block2.0.weight',
           'discriminator_block.block3.1.weight', 'discriminator_block.block4.1.bias',
           'dense.weight', 'dense.bias']


def make_opt(model_name, thr=0.4):
    opt = {
        'scale': SCALE, 'dist': False, 'device': 'cuda', 'rank': 0, 'world_size': 1, 'is_train': True,
        'dataset': {'degradation': {'type': 'BD', 'sigma': 1.5}, 'train': {'crop_size': CROP}},
        'model': {'name': model_name,
                  'generator': {'name': 'FRNet', 'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 10,
                                'load_path': None},
                  'discriminator': {'name': 'STNet', 'in_nc': 3, 'tempo_range': 3, 'load_path': None}},
        'train': {'tempo_extent': T, 'ckpt_dir': '/tmp',
                  'generator': {'lr': 1e-4 if model_name == 'FRVSR' else 5e-5, 'betas': [0.9, 0.999]},
                  'discriminator': {'update_policy': 'adaptive', 'update_threshold': thr,
                                    'crop_border_ratio': 0.75, 'lr': 5e-5, 'betas': [0.9, 0.999]},
                  'pixel_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'warping_crit': {'type': 'CB', 'weight': 1, 'reduction': 'mean'},
                  'pingpong_crit': {'type': 'CB', 'weight': 0.5, 'reduction': 'mean'},
                  'gan_crit': {'type': 'GAN', 'weight': 0.01, 'reduction': 'mean'}},
        'logger': {'decay': 0.99},
    }
    if model_name == 'FRVSR':
        del opt['train']['pingpong_crit'], opt['train']['gan_crit']
    return opt


def batch(seed):
    return torch.stack([smooth_clip(T, 3, CROP + 8, CROP + 8, seed=seed + i, shift=1.0)
                        for i in range(N)])


def digest(v):
    v = v.detach().double().cpu().reshape(-1)
    return np.array([v.norm().item(), v.sum().item(), v[0].item(), v[v.numel() // 2].item(),
                     v[-1].item()])


def close_digest(mine, ref, rel, what):
    scale = abs(ref[0]) + 1e-12          # the tensor's L2 norm
    assert abs(mine[0] - ref[0]) <= rel * scale, (what, 'norm', mine, ref)
    assert np.all(np.abs(mine[2:] - ref[2:]) <= rel * scale), (what, 'samples', mine, ref)

DEV = 'cuda'


def rs(seed, shape, lo=0.0, hi=1.0):
    g = np.random.RandomState(seed)
    return torch.from_numpy(g.uniform(lo, hi, size=shape).astype(np.float32))


@pytest.fixture(scope='module')
def ops():
    from tecogan_pytorch_amd import ops as o
    return o


def test_channel_norm_fwd_bwd(ops):
    x = rs(1, (3, 3, 17, 23))
    mean = torch.tensor(O.VGG_MEAN)
    std = torch.tensor(O.VGG_STD)
    ref = (x - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    y = ops.channel_norm(x.to(DEV), mean.to(DEV), std.to(DEV))
    assert torch.equal(y.cpu(), ref)                      # same two IEEE operations
    g = rs(2, x.shape, -1, 1)
    dx = ops.channel_norm(g.to(DEV), None, std.to(DEV))
    assert torch.equal(dx.cpu(), g / std.view(1, 3, 1, 1))


@pytest.mark.parametrize('mode,kind', [(1, 'L1'), (2, 'MSE')])
def test_pixel_loss_vs_oracle(ops, mode, kind):
    x = rs(3, (2, 5, 9, 11), -1, 1).requires_grad_(True)
    y = rs(4, (2, 5, 9, 11), -1, 1)
    y.view(-1)[::7] = x.detach().view(-1)[::7]            # exact ties: sign(0) = 0 for L1
    loss = 0.7 * O.pointwise_criterion(kind, x, y, 'mean')
    loss.backward()
    acc = torch.zeros(1, device=DEV)
    sc = 0.7 / x.numel()
    dx = ops.pixel_loss(x.detach().to(DEV), y.to(DEV), mode, acc, sc, grad_scale=sc)
    assert abs(acc.item() - loss.item()) <= 1e-6 * abs(loss.item()) + 1e-8
    assert (dx.cpu() - x.grad).abs().max().item() <= 1e-9 + 1e-6 * x.grad.abs().max().item()


@pytest.mark.parametrize('shape', [(2, 128, 16, 16), (3, 512, 2, 2), (1, 64, 5, 7)])
def test_cosine_loss_vs_oracle(ops, shape):
    a = torch.relu(rs(5, shape, -0.5, 1.0)).requires_grad_(True)
    b = torch.relu(rs(6, shape, -0.5, 1.0))
    with torch.no_grad():
        a[0, :, 0, 0] = 0.0                               # |a| < eps: clamp, no norm gradient
        b[0, :, 1, 1] = 0.0
    loss = 0.2 * O.cosine_similarity_loss(a, b)
    loss.backward()
    acc = torch.zeros(1, device=DEV)
    sc = 0.2 / (shape[0] * shape[2] * shape[3])
    da = ops.cosine_loss(a.detach().to(DEV), b.to(DEV), acc, sc, grad_scale=sc)
    assert abs(acc.item() - loss.item()) <= 2e-6 * abs(loss.item()) + 1e-8
    ref = a.grad
    # the zero-norm pixel has gradient b/(eps |b|) ~ 1e8 * scale: compare relatively
    assert torch.isfinite(ref).all()
    assert ((da.cpu() - ref).abs() <= 1e-7 + 2e-5 * ref.abs().clamp_min(ref.abs().median())).all()


def _vgg(dev=DEV):
    from tecogan_pytorch_amd.models.networks.vgg_nets import VGGFeatureExtractor
    net = VGGFeatureExtractor([8, 17, 26, 35]).to(dev)
    net.load_vgg19_state_dict(vgg19_state_dict())
    return net


def test_vgg_features_vs_reference_and_gradient_vs_oracle(golden):
    """Forward: the reference's VGGFeatureExtractor outputs (golden).  Backward: gradient of the
    summed cosine losses w.r.t. the input image, reference autograd (golden)."""
    from tecogan_pytorch_amd import ops
    from tecogan_pytorch_amd.models import train_graph as TG
    g = golden('train_feat')
    net = _vgg()
    x, y = torch.from_numpy(g['op_x']).to(DEV), torch.from_numpy(g['op_y']).to(DEV)
    tape = TG.Tape()
    fx = net(x, tape)
    fy = net(y)
    for i, f in enumerate(fx):
        v = f.double()
        st = np.array([v.norm().item(), v.sum().item(), v.max().item()])
        assert np.all(np.abs(st - g[f'op_feat{i}_stats']) <= 2e-5 * np.abs(g[f'op_feat{i}_stats']) + 1e-5), i
    assert (fx[3].cpu() - torch.from_numpy(g['op_feat3'])).abs().max().item() <= 2e-5
    acc = torch.zeros(1, device=DEV)
    for a, b in zip(fx, fy):
        sc = 1.0 / (a.shape[0] * a.shape[2] * a.shape[3])
        tape.add_grad(a, ops.cosine_loss(a, b, acc, sc, grad_scale=sc))
    tape.backward()
    assert abs(acc.item() - float(g['op_loss'])) <= 2e-6 * abs(float(g['op_loss'])) + 1e-7
    gx = tape.grad(x).cpu().numpy()
    ref = g['op_grad_x']
    assert np.abs(gx - ref).max() <= 2e-3 * np.abs(ref).max(), (np.abs(gx - ref).max(), np.abs(ref).max())


def test_vgg_rejects_bad_inputs():
    from tecogan_pytorch_amd.models.networks.vgg_nets import VGGFeatureExtractor
    with pytest.raises(ValueError):
        VGGFeatureExtractor([7])                          # a conv layer, not a ReLU / pool
    net = VGGFeatureExtractor([8])
    sd = vgg19_state_dict()
    del sd['features.5.weight']
    with pytest.raises(KeyError):
        net.load_vgg19_state_dict(sd)


def test_feature_crit_without_weights_fails_loudly():
    from tecogan_pytorch_amd.models import define_model
    opt = make_opt('TecoGAN')
    opt['train']['feature_crit'] = {'type': 'CosineSimilarity', 'weight': 0.2}
    with pytest.raises(FileNotFoundError):
        define_model(opt)


VARIANTS = {
    'feat': dict(feature_crit={'type': 'CosineSimilarity', 'weight': 0.2, 'reduction': 'mean',
                               'feature_layers': [8, 17, 26, 35], 'init': 'default'}),
    'featfm': dict(feature_crit={'type': 'CosineSimilarity', 'weight': 0.2, 'reduction': 'mean',
                                 'feature_layers': [8, 17, 26, 35], 'init': 'default'},
                   feature_matching_crit={'type': 'CB', 'weight': 0.3, 'reduction': 'mean'}),
    'fm_l1': dict(feature_matching_crit={'type': 'L1', 'weight': 0.5, 'reduction': 'mean',
                                         'layer_norm': [10.0, 12.0, 20.0, 80.0]}),
}


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_tecogan_feature_losses_two_iterations(golden, tag):
    from tecogan_pytorch_amd.models import define_model
    g = golden('train_feat')
    opt = make_opt('TecoGAN')
    opt['train'].update(VARIANTS[tag])
    m = define_model(opt)
    m.net_G.load_state_dict(generator_state_dict(scale=SCALE, degradation='BD'), strict=True)
    m.net_D.load_state_dict(discriminator_state_dict(spatial_size=CROP, scale=SCALE, degradation='BD'),
                            strict=True)
    if 'feature_crit' in VARIANTS[tag]:
        m.net_F.load_vgg19_state_dict(vgg19_state_dict())
    keys = list(g[f'{tag}_log_keys'])
    for it in range(2):
        m.prepare_training_data({'gt': batch(300 + 10 * it)})
        m.train()
        assert list(m.log_dict.keys()) == keys          # same entries, same order as the reference
        ref = dict(zip(keys, g[f'{tag}_log{it}']))
        for k in keys:
            post_update = k in ('l_gan_G', 'p_fake_G', 'l_fm_G') or it > 0
            rtol, atol = (1e-2, 5e-4) if post_update else (5e-4, 2e-5)
            assert abs(m.log_dict[k] - ref[k]) <= rtol * abs(ref[k]) + atol, \
                (tag, it, k, m.log_dict[k], ref[k])
        if it == 0:
            pg = dict(m.net_G.named_parameters())
            for k in WATCH_G:
                close_digest(digest(pg[k].grad), g[f'{tag}_gradG_' + k], 2e-2, 'gradG ' + k)
