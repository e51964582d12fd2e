"""
Constant filter kernels of the channel (host side, numpy).  Same functions, names and argument meaning as the
reference's helpers/kernels.py:94-123; values are pinned against the reference in tests/test_golden.py.
"""
import numpy as np


def gkern(kernlen=5, std=0.83):
    """2-D Gaussian kernel, normalised (helpers/kernels.py:94-98)."""
    n = np.arange(kernlen) - (kernlen - 1.0) / 2.0
    g1 = np.exp(-0.5 * (n / std) ** 2)
    g2 = np.outer(g1, g1)
    return g2 / g2.sum()


def repeat_2dfilter(f, channels=3, pad=0):
    """Repeat a 2-D filter on the channel diagonal -> (k,k,channels,channels)  (helpers/kernels.py:101-114)."""
    rf = np.zeros((f.shape[0] + 2 * pad, f.shape[1] + 2 * pad, channels, channels))
    for r in range(channels):
        rf[:, :, r, r] = np.pad(f, [pad, pad], 'constant')
    return rf


def center_mask_2dfilter(f_size, channels):
    """Centre-tap indicator on the channel diagonal (helpers/kernels.py:117-123)."""
    indicator = np.zeros((f_size, f_size, channels, channels))
    for r in range(channels):
        indicator[f_size // 2, f_size // 2, r, r] = 1
    return indicator


def residual_init_filter():
    """Initial value of the constrained residual filter (models/layers.py:40-43)."""
    f = np.array([[0, 0, 0, 0, 0], [0, -1, -2, -1, 0], [0, -2, 12, -2, 0], [0, -1, -2, -1, 0], [0, 0, 0, 0, 0]])
    return repeat_2dfilter(f, 3)


def sharpen_kernel(strength=1.0):
    """3x3 H/V sharpening taps of manipulation_sharpen (helpers/tf_helpers.py:158-160)."""
    gk = np.array([[-0.0833, -0.1667, -0.0833], [-0.1667, 0, -0.1667], [-0.0833, -0.1667, -0.0833]])
    gk = strength * gk / np.abs(gk.sum())
    gk[1, 1] = strength + 1
    return gk


def bilinear_axis_matrix(in_size, out_size):
    """Dense (out,in) matrix of tf.image.resize(bilinear, half-pixel centres, no antialias) along one axis."""
    m = np.zeros((out_size, in_size), np.float64)
    scale = in_size / out_size
    for o in range(out_size):
        src = (o + 0.5) * scale - 0.5
        f = int(np.floor(src))
        lo, hi, t = max(f, 0), min(int(np.ceil(src)), in_size - 1), src - f
        m[o, lo] += 1.0 - t
        m[o, hi] += t
    return m
