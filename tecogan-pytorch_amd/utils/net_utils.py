"""Hot functional ops, same names/signatures as the reference's
codes/utils/net_utils.py, each a single HIP kernel launch."""
import torch
import torch.nn as nn

from .. import ops


def space_to_depth(x, scale):
    """Equivalent of net_utils.py:36-47 (tf.space_to_depth channel order)."""
    return ops.space_to_depth(x.contiguous(), scale)


def backward_warp(x, flow, mode='bilinear', padding_mode='border'):
    """net_utils.py:50-82: x (n,c,h,w), flow (n,2,h,w) in pixels."""
    if mode != 'bilinear' or padding_mode != 'border':
        raise ValueError('only bilinear / border is implemented (the only use on the path)')
    return ops.backward_warp(x.contiguous(), flow.contiguous())


class BicubicUpsampler(nn.Module):
    """net_utils.py:101-156.  The `kernels` buffer is kept (state-dict key
    `upsample_func.kernels`), the kernel evaluates the same Keys weights."""

    def __init__(self, scale_factor, a=-0.75):
        super().__init__()
        if a != -0.75 or scale_factor not in (2, 4):
            raise ValueError('HIP bicubic path: a=-0.75, scale 2 or 4')
        cubic = torch.tensor([[0, a, -2 * a, a], [1, 0, -(a + 3), a + 2],
                              [0, -a, (2 * a + 3), -(a + 2)], [0, 0, a, -a]],
                             dtype=torch.float32)
        ks = [cubic @ torch.tensor([1, s, s ** 2, s ** 3], dtype=torch.float32)
              for s in [1.0 * d / scale_factor for d in range(scale_factor)]]
        self.scale_factor = scale_factor
        self.register_buffer('kernels', torch.stack(ks))

    def forward(self, input):
        return ops.upsample(input.contiguous(), self.scale_factor, ops.UP_BICUBIC)


class BilinearUpsampler:
    """functools.partial(F.interpolate, bilinear, align_corners=False) stand-in
    (net_utils.py:86-89); not a Module, so it adds no state-dict entries."""

    def __init__(self, scale_factor):
        self.scale_factor = scale_factor

    def __call__(self, input):
        return ops.upsample(input.contiguous(), self.scale_factor, ops.UP_BILINEAR)


def get_upsampling_func(scale=4, degradation='BI'):
    if degradation == 'BI':
        return BilinearUpsampler(scale)
    elif degradation == 'BD':
        return BicubicUpsampler(scale_factor=scale)
    raise ValueError(f'Unrecognized degradation type: {degradation}')
