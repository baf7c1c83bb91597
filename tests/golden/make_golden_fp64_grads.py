"""fp64 triangulation of the full-size training step's gradients.

tests/test_hip_parity_long.py::test_fullsize_tecogan_train_step_vs_oracle compares the HIP step's gradients with
the oracle's fp32 autograd at a relative L2 of up to 1e-2 -- not an fp32-looking number.  This script runs the SAME
oracle step (oracle/tecogan_oracle.py::vsrgan_train_step, the restatement of vsrgan_model.py:98-286) on the SAME
seeded inputs and procedural weights in float64 and stores the watched gradients (strided to <= 16384 values per
tensor, as float32 of the fp64 result).  The test then reports ||HIP - fp64|| beside ||oracle-fp32 - fp64||: if they
are the same size, the 1e-2 is the conditioning of a 19-frame BPTT / of BatchNorm-through-a-sign-changing-sum in
fp32, not a defect of the HIP backward.

No reference import is needed (the oracle is pinned against the reference by the other fixtures).
    python tests/golden/make_golden_fp64_grads.py [crop ...]     -> tests/golden/train_fp64_grads.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import tecogan_oracle as O  # noqa: E402
from procedural_weights import discriminator_state_dict, generator_state_dict, smooth_clip  # noqa: E402

WATCH_G = ['fnet.encoder1.0.weight', 'fnet.decoder1.2.bias', 'fnet.flow.2.weight',
           'srnet.conv_in.0.weight', 'srnet.resblocks.4.conv.2.weight', 'srnet.conv_up.2.weight',
           'srnet.conv_out.bias']
WATCH_D = ['conv_in.0.weight', 'discriminator_block.block2.0.weight',
           'discriminator_block.block3.1.weight', 'discriminator_block.block4.1.bias',
           'dense.weight', 'dense.bias']
CAP = 16384


def pick(n):
    """the strided index set of a tensor of n elements stored in the fixture (also used by the test)"""
    step = max(1, -(-n // CAP))
    return np.arange(0, n, step)


def run(crop, dtype):
    n, tempo, s, deg = 2, 10, 4, 'BD'
    gt = torch.stack([smooth_clip(tempo, 3, crop + 8, crop + 8, seed=900 + i, shift=1.0) for i in range(n)])
    lr_o, gt_o = O.prepare_training_data(gt, s, deg)          # fp32, exactly what the test feeds both sides
    sdg = {k: v.clone().to(dtype) if v.is_floating_point() else v.clone()
           for k, v in generator_state_dict(scale=s, degradation=deg).items()}
    sdd = {k: v.clone().to(dtype) if v.is_floating_point() else v.clone()
           for k, v in discriminator_state_dict(spatial_size=crop, scale=s, degradation=deg).items()}
    ref, rG, rD = O.vsrgan_train_step(sdg, sdd, {}, {}, {}, lr_o.to(dtype), gt_o.to(dtype), s, deg,
                                      spatial_size=crop, tempo_extent=tempo, update_threshold=0.4)
    assert ref['l_gan_D'] != 0.0
    return ref, rG, rD


def main():
    crops = [int(a) for a in sys.argv[1:]] or [128]
    out = {}
    path = os.path.join(HERE, 'train_fp64_grads.npz')
    if os.path.isfile(path):
        out.update(dict(np.load(path)))
    for crop in crops:
        ref, rG, rD = run(crop, torch.float64)
        for tag, names, grads in (('G', WATCH_G, rG), ('D', WATCH_D, rD)):
            for k in names:
                g = grads[k].detach().double().reshape(-1).numpy()
                out['c%d_%s_%s' % (crop, tag, k)] = g[pick(g.size)].astype(np.float32)
                out['c%d_%s_%s_norm' % (crop, tag, k)] = np.float64(np.linalg.norm(g))
        for k, v in ref.items():
            out['c%d_log_%s' % (crop, k)] = np.float64(v)
        print('crop', crop, {k: float(v) for k, v in ref.items()})
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
