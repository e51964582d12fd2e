"""
Image quality metrics on the device - the counterpart of the reference's helpers/metrics.py (ssim :9-25, psnr :28-44,
mse :47-65, mae :68-86, batch :89-92), which wraps skimage on host arrays.  Same call shapes: two (H,W,C) images return
a scalar, two (N,H,W,C) batches return one value per image; inputs may be numpy arrays or DeviceArrays.
"""
import numpy as np
import torch

from .. import ops
from ..device import DeviceArray, default_device, to_device


def _pair(a, b):
    a = a.t if isinstance(a, DeviceArray) else a
    b = b.t if isinstance(b, DeviceArray) else b
    dev = a.device if isinstance(a, torch.Tensor) and a.is_cuda else (
        b.device if isinstance(b, torch.Tensor) and b.is_cuda else default_device())
    a, b = to_device(a, dev), to_device(b, dev)
    if a.ndim == 4 and a.shape[0] == 1:
        a = a[0]
    if b.ndim == 4 and b.shape[0] == 1:
        b = b[0]
    if a.ndim != b.ndim or a.ndim not in (3, 4):
        raise ValueError('Incompatible tensor shapes! {} and {}'.format(tuple(a.shape), tuple(b.shape)))
    single = a.ndim == 3
    if single:
        a, b = a[None], b[None]
    return a.contiguous(), b.contiguous(), single


def _ret(v, single):
    v = v.double().cpu().numpy()
    return float(v[0]) if single else v


def ssim(a, b):
    """skimage.metrics.structural_similarity(a, b, multichannel=True, data_range=1): 7x7 uniform window."""
    a, b, single = _pair(a, b)
    return _ret(ops.ssim(a, b, mode='skimage', max_val=1.0), single)


def _mean_per_image(a, b, fn):
    d = fn(a.double() - b.double())
    return d.reshape(d.shape[0], -1).mean(dim=1)


def mse(a, b):
    a, b, single = _pair(a, b)
    return _ret(_mean_per_image(a, b, lambda d: d * d), single)


def mae(a, b):
    a, b, single = _pair(a, b)
    return _ret(_mean_per_image(a, b, torch.abs), single)


def psnr(a, b):
    """skimage.metrics.peak_signal_noise_ratio(a, b, data_range=1) = 10 log10(1 / mse)."""
    a, b, single = _pair(a, b)
    return _ret(10.0 * torch.log10(1.0 / _mean_per_image(a, b, lambda d: d * d)), single)


def batch(a, b, metric=ssim):
    assert a.ndim == 4 and b.ndim == 4, 'Input arrays need to be 4-dim: batch, height, width, channels'
    assert len(a) == len(b), 'Image batches must be of the same length'
    return float(np.mean(metric(a, b)))
