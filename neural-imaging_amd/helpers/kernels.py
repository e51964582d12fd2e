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


# tf.image.resize's other methods (image_ops_impl.py resize_images_v2 with antialias=False, the call of the reference's
# manipulation_resample, helpers/tf_helpers.py:68-76): every one of them is a separable linear map, so each becomes an (out, in)
# matrix per axis like the two above and runs through the same banded-operator kernels.  The weights are computed in float32 in
# the order of TensorFlow's kernels (resize_bicubic_op.cc, resize_area_op.cc, scale_and_translate_op.cc + sampling_kernels.h).
def bicubic_axis_matrix(in_size, out_size):
    """ResizeBicubic(half_pixel_centers=True): Keys cubic (a = -0.5) read from the kernel's 1024-step coefficient table, the four
    taps around floor(src); taps that fall outside the image weigh 0 and the others are renormalised to sum 1."""
    f32, steps, a = np.float32, 1024, -0.5
    x = (np.arange(steps + 1) * 1.0 / steps).astype(f32)
    near = (((a + 2) * x.astype(np.float64) - (a + 3)) * x * x + 1).astype(f32)              # |t| <= 1
    x1 = (x + f32(1.0)).astype(np.float64)
    far = (((a * x1 - 5 * a) * x1 + 8 * a) * x1 - 4 * a).astype(f32)                         # 1 <= |t| <= 2
    o = np.arange(out_size, dtype=f32)
    src = (o + f32(0.5)) * (f32(in_size) / f32(out_size)) - f32(0.5)
    base = np.floor(src)
    off = np.rint((src - base) * f32(steps)).astype(np.int64)
    base = base.astype(np.int64)
    idx = base[:, None] + np.array([-1, 0, 1, 2])
    w = np.stack([far[off], near[off], near[steps - off], far[steps - off]], 1)
    w = np.where((idx >= 0) & (idx < in_size), w, f32(0.0))
    tot = ((w[:, 0] + w[:, 1]) + w[:, 2]) + w[:, 3]
    ok = np.abs(tot) >= 1000.0 * np.finfo(f32).tiny
    w = np.where(ok[:, None], w * (f32(1.0) / np.where(ok, tot, f32(1.0)))[:, None], w).astype(f32)
    m = np.zeros((out_size, in_size), np.float64)
    np.add.at(m, (np.repeat(np.arange(out_size), 4), np.clip(idx, 0, in_size - 1).ravel()), w.ravel().astype(np.float64))
    return m


def area_axis_matrix(in_size, out_size):
    """ResizeArea: the mean of the input over [o s, (o + 1) s), s = in / out - edge pixels by the covered fraction, clamped indices."""
    f32 = np.float32
    s = f32(in_size) / f32(out_size)
    lo = (np.arange(out_size, dtype=f32) * s).astype(f32)
    hi = (np.arange(1, out_size + 1, dtype=f32) * s).astype(f32)
    m = np.zeros((out_size, in_size), np.float64)
    for o in range(out_size):                                                    # spans differ in length: one short loop per output
        v = np.arange(int(np.floor(lo[o])), int(np.ceil(hi[o])), dtype=np.int64)
        vf = v.astype(f32)
        before, beyond = vf < lo[o], vf + f32(1.0) > hi[o]
        cover = np.where(before, np.where(beyond, s, vf + f32(1.0) - lo[o]), np.where(beyond, hi[o] - vf, f32(1.0))).astype(f32)
        np.add.at(m[o], np.clip(v, 0, in_size - 1), cover.astype(np.float64) / float(s))
    return m


def _sampling_kernel(name):
    f32 = np.float32
    pi = f32(3.14159265359)
    if name in ('lanczos3', 'lanczos5'):
        r = f32(3.0 if name == 'lanczos3' else 5.0)

        def fn(x):
            xs = np.where(x <= f32(1e-3), f32(1.0), x)                            # (the guarded lanes are overwritten below)
            y = r * np.sin(pi * xs, dtype=f32) * np.sin(pi * xs / r, dtype=f32) / (pi * pi * (xs * xs))
            return np.where(x > r, f32(0.0), np.where(x <= f32(1e-3), f32(1.0), y)).astype(f32)
        return r, fn
    if name == 'gaussian':
        r = f32(1.5)
        sigma = float(r / f32(3.0))
        return r, lambda x: np.where(x >= r, 0.0, np.exp(-x.astype(np.float64) ** 2 / (2.0 * sigma * sigma))).astype(f32)
    if name == 'mitchellcubic':
        def fn(x):
            outer = ((f32(-7.0) / f32(18.0) * x + f32(2.0)) * x - f32(10.0) / f32(3.0)) * x + f32(16.0) / f32(9.0)
            inner = ((f32(7.0) / f32(6.0) * x - f32(2.0)) * x) * x + f32(8.0) / f32(9.0)
            return np.where(x >= f32(2.0), f32(0.0), np.where(x >= f32(1.0), outer, inner)).astype(f32)
        return f32(2.0), fn
    raise ValueError(name)


def scale_translate_axis_matrix(in_size, out_size, kernel):
    """ScaleAndTranslate(kernel_type, antialias=False), no translation: output o samples the input at (o + 0.5) in / out with the
    kernel at its native width (no widening on down-sampling), over the pixel centres inside the radius and the image; weights
    normalised per output."""
    f32 = np.float32
    radius, fn = _sampling_kernel(kernel)
    inv_scale = f32(1.0) / (f32(out_size) / f32(in_size))
    sample = ((np.arange(out_size, dtype=f32) + f32(0.5)) * inv_scale).astype(f32)
    first = np.clip(np.ceil(sample - radius - f32(0.5)).astype(np.int64), 0, in_size - 1)
    last = np.clip(np.floor(sample + radius - f32(0.5)).astype(np.int64), 0, in_size - 1)
    width = int((last - first).max()) + 1
    src = first[:, None] + np.arange(width)
    live = (src <= last[:, None]) & ((sample >= 0) & (sample <= in_size))[:, None]
    w = np.where(live, fn(np.abs(src.astype(f32) + f32(0.5) - sample[:, None]).astype(f32)), f32(0.0)).astype(f32)
    tot = np.zeros(out_size, f32)
    for k in range(width):                                                       # the kernel's left-to-right float32 sum
        tot = (tot + w[:, k]).astype(f32)
    ok = np.abs(tot) >= 1000.0 * np.finfo(f32).tiny
    w = np.where(ok[:, None], w * (f32(1.0) / np.where(ok, tot, f32(1.0)))[:, None], f32(0.0)).astype(f32)
    m = np.zeros((out_size, in_size), np.float64)
    rows = np.repeat(np.arange(out_size), width)
    np.add.at(m, (rows, np.clip(src, 0, in_size - 1).ravel()), np.where(live, w, f32(0.0)).ravel().astype(np.float64))
    return m


RESIZE_AXIS_MATRIX = {'bilinear': bilinear_axis_matrix, 'nearest': nearest_axis_matrix, 'bicubic': bicubic_axis_matrix,
                      'area': area_axis_matrix}
for _name in ('lanczos3', 'lanczos5', 'gaussian', 'mitchellcubic'):
    RESIZE_AXIS_MATRIX[_name] = (lambda i, o, _name=_name: scale_translate_axis_matrix(i, o, _name))


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
