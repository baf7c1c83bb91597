"""PSNR on the Y channel, the reference's in-loop quality metric
(codes/metrics/metric_calculator.py:228-244, codes/utils/data_utils.py:56-77)."""
import numpy as np

_T = np.array([[0.256788235294118, -0.148223529411765, 0.439215686274510],
               [0.504129411764706, -0.290992156862745, -0.367788235294118],
               [0.097905882352941, 0.439215686274510, -0.071427450980392]], dtype=np.float64)
_O = np.array([16, 128, 128], dtype=np.float64)


def rgb_to_ycbcr(img):
    res = np.matmul(img.astype(np.float64), _T) + _O
    return res.clip(0, 255).round().astype(np.uint8)


def compute_psnr(true_img, pred_img, colorspace='y'):
    """hwc uint8 frames -> dB (inf when identical)."""
    if colorspace == 'y':
        true_img = rgb_to_ycbcr(true_img)[..., 0]
        pred_img = rgb_to_ycbcr(pred_img)[..., 0]
    diff = true_img.astype(np.float64) - pred_img.astype(np.float64)
    rmse = np.sqrt(np.mean(np.power(diff, 2)))
    return np.inf if rmse == 0 else 20 * np.log10(255.0 / rmse)


def compute_psnr_device(true_u8, pred_u8, colorspace='y'):
    """compute_PSNR for whole clips resident on the GPU: (t,h,w,3) uint8 CUDA tensors ->
    list of per-frame dB values.  The squared-error sums are exact integers computed by the
    HIP kernel; only the final 20*log10(255/rmse) per frame is host arithmetic."""
    from .. import ops
    sse = ops.psnr_sse_u8(true_u8, pred_u8, y_only=(colorspace == 'y')).tolist()
    t, h, w, c = true_u8.shape
    count = h * w * (1 if colorspace == 'y' else c)
    out = []
    for s_ in sse:
        rmse = np.sqrt(s_ / count)
        out.append(np.inf if rmse == 0 else 20 * np.log10(255.0 / rmse))
    return out
