"""Host-side helpers mirroring codes/utils/data_utils.py for the path."""
import numpy as np


def float32_to_uint8(inputs):
    """data_utils.py:80-87 (host version, for arrays already on the CPU).
    The device path is ops.quantize_u8_hwc."""
    return np.uint8(np.clip(np.round(inputs * 255), 0, 255))


def gaussian_kernel2d(sigma, ksize=None):
    """2-D Gaussian of create_kernel (data_utils.py:11-20), one channel."""
    if ksize is None:
        ksize = 1 + 2 * int(sigma * 3.0)
    xs = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    g1 = np.exp(-0.5 * (xs / sigma) ** 2)
    g2 = np.outer(g1, g1)
    return np.float32(g2 / g2.sum())
