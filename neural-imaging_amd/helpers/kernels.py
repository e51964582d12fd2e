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


def nearest_axis_matrix(in_size, out_size):
    """Dense (out,in) 0/1 matrix of tf.image.resize(nearest) along one axis: half-pixel centres, source index
    floor((o + 0.5) * in/out) in float32, clamped to the last pixel (ResizeNearestNeighbor, half_pixel_centers=True)."""
    m = np.zeros((out_size, in_size), np.float64)
    scale = np.float32(in_size) / np.float32(out_size)
    for o in range(out_size):
        src = int(np.floor((np.float32(o) + np.float32(0.5)) * scale))
        m[o, min(src, in_size - 1)] = 1.0
    return m


def residual_kernel():
    """3x3 high-pass taps of residual() (helpers/tf_helpers.py:131)."""
    return np.array([[-0.0833, -0.1667, -0.0833], [-0.1667, 1, -0.1667], [-0.0833, -0.1667, -0.0833]])


def upsampling_kernel(cfa_pattern='gbrg'):
    """(4,12) 1x1 kernel that scatters the RAW planes [R, G (first in raster order), G (second), B] of a 2x2 Bayer cell
    to the (position, colour) slots depth_to_space(2) expects - slot = 2*dy + dx, colour R/G/B = 0/1/2 - for the CFA
    layouts 'gbrg' | 'rggb' | 'bggr' (helpers/kernels.py:9-46 of the reference)."""
    cfa = cfa_pattern.lower()
    if cfa not in ('gbrg', 'rggb', 'bggr'):
        raise ValueError('Unsupported CFA pattern: {}'.format(cfa_pattern))
    upk = np.zeros((4, 12))
    greens = [s for s, ch in enumerate(cfa) if ch == 'g']
    for plane, (slot, colour) in enumerate([(cfa.index('r'), 0), (greens[0], 1), (greens[1], 1), (cfa.index('b'), 2)]):
        upk[plane, 3 * slot + colour] = 1
    return upk


def bilin_kernel(kernel=3):
    """(k,k,3,3) bilinear demosaicing filter on the zero-filled colour planes: the green plane with the 4-neighbour cross,
    red / blue with the separable [1/2, 1, 1/2] tent; channel-diagonal; zero-padded from 3x3 to k x k."""
    tent = np.array([0.5, 1.0, 0.5])
    rb = np.outer(tent, tent)
    g = np.array([[0, 0.25, 0], [0.25, 1, 0.25], [0, 0.25, 0]])
    dmf = np.zeros((3, 3, 3, 3), np.float32)
    dmf[:, :, 0, 0], dmf[:, :, 1, 1], dmf[:, :, 2, 2] = rb, g, rb
    if kernel > 3:
        pad = (kernel - 3) // 2
        dmf = np.pad(dmf, ((pad, pad), (pad, pad), (0, 0), (0, 0)), 'constant')
    return dmf


def gamma_kernels():
    """Weights of the reference's pre-trained toy gamma MLP (1 -> 4 tanh -> 1 per colour channel; the numbers are model
    data from helpers/kernels.py:49-71), expanded to block-diagonal (3,12) / (12,3) 1x1 kernels."""
    d1k = np.array([2.9542332, 17.780445, 0.6280197, 0.40384966])
    d1b = np.array([0.4047071, 1.1489044, -0.17624384, 0.47826886])
    d2k = np.array([0.44949612, 0.78081024, 0.97692937, -0.24265033])
    d2b = -0.4702738
    k1, b1, k2, b2 = np.zeros((3, 12)), np.zeros(12), np.zeros((12, 3)), np.zeros(3)
    for ch in range(3):
        k1[ch, 4 * ch:4 * ch + 4], b1[4 * ch:4 * ch + 4] = d1k, d1b
        k2[4 * ch:4 * ch + 4, ch], b2[ch] = d2k, d2b
    return k1, b1, k2, b2


# example camera-RGB -> sRGB conversion INet starts from (pipelines.py:262-264 of the reference; model data)
SRGB_EXAMPLE = np.array([[1.82691061, -0.65497452, -0.17193617],
                         [-0.00683982, 1.33216381, -0.32532394],
                         [0.06269717, -0.40055895, 1.33786178]]).T
