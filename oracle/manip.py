"""
Photo manipulations restated (helpers/tf_helpers.py:68-184).  TEST INFRASTRUCTURE.
"""
import numpy as np
import torch

from . import tables
from . import tfops as T


def manipulation_sharpen(x, strength=1, hsv=True):
    """tf_helpers.py:156-184: SYMMETRIC pad 1 -> rgb_to_hsv -> 3x3 per-channel filter (S zeroed, corner tap
    [2,2]=1 => S shifted by (+1,+1); H and V sharpened) -> hsv_to_rgb -> hard clip."""
    gf = torch.tensor(tables.sharpen_filter(strength, hsv).astype(np.float32), dtype=x.dtype)   # tf.constant(.., tf.float32)
    y = T.pad2d(x, 1, 'SYMMETRIC')
    if hsv:
        y = T.rgb_to_hsv(y)
    y = T.conv2d(y, gf, None, 1, 'VALID')
    if hsv:
        y = T.hsv_to_rgb(y)
    return torch.clamp(y, 0, 1)


def manipulation_resample(x, factor=50, method='bilinear'):
    """tf_helpers.py:68-76: both dims sized from shape[1]."""
    if 0 < factor <= 1:
        factor = 100 * factor
    s = x.shape[1] * int(factor) // 100
    if method in T.RESIZE_AXIS:
        resize = lambda t, oh, ow: T.resize_separable(t, oh, ow, method)
    else:
        resize = {'bilinear': T.resize_bilinear, 'nearest': T.resize_nearest}[method]
    down = resize(x, s, s)
    return resize(down, x.shape[1], x.shape[1])


def residual(x):
    """tf_helpers.py:127-154 with hsv=False: REFLECT pad 1, the fixed 3x3 high-pass per channel, no clip."""
    gk = np.array([[-0.0833, -0.1667, -0.0833], [-0.1667, 1, -0.1667], [-0.0833, -0.1667, -0.0833]])
    gf = torch.tensor(tables.repeat_2dfilter(gk, 3).astype(np.float32), dtype=x.dtype)
    return T.conv2d(T.pad2d(x, 1, 'REFLECT'), gf, None, 1, 'VALID')


def manipulation_gaussian(x, kernel=5, std=0.83, skip_clip=False):
    """tf_helpers.py:113-125: REFLECT pad, depthwise (diagonal) gaussian, clip."""
    kernel = int(kernel)
    gk = tables.gkern(kernel, std)
    gf = np.zeros((kernel, kernel, 3, 3))
    for r in range(3):
        gf[:, :, r, r] = gk
    gf = torch.tensor(gf.astype(np.float32), dtype=x.dtype)       # tf.constant(gfilter, tf.float32)
    y = T.conv2d(T.pad2d(x, kernel // 2, 'REFLECT'), gf, None, 1, 'VALID')
    return y if skip_clip else torch.clamp(y, 0, 1)


def manipulation_awgn(x, strength=0.025, noise=None):
    """tf_helpers.py:79-82 with injectable noise (tf.random.normal is not reproducible)."""
    if noise is None:
        noise = torch.randn_like(x)
    y = T.soft_quantization(x + strength * noise)
    return torch.clamp(y, 0, 1)


def manipulation_gamma(x, strength=2.0):
    """tf_helpers.py:85-88"""
    y = T.soft_quantization(torch.pow(x, strength))
    return torch.pow(torch.clamp(y, 1.0 / 255, 1), 1 / strength)


def manipulation_median(x, kernel=3):
    """tf_helpers.py:91-110"""
    kernel = int(kernel)
    if kernel % 2 == 0:
        kernel += 1
    kernel = max(kernel, 1)
    xp = T.pad2d(x, kernel // 2, 'REFLECT')
    n, h, w, c = x.shape
    patches = torch.stack([xp[:, i:i + h, j:j + w, :] for i in range(kernel) for j in range(kernel)], dim=-1)
    area = kernel ** 2
    return torch.sort(patches, dim=-1, descending=True).values[..., (area + 1) // 2 - 1]
