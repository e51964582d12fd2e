"""
Photo manipulations and image losses of the channel on the HIP kernels.  Same function names and argument meaning as
the reference's helpers/tf_helpers.py (manipulation_* :68-184, mse :31-32); each manipulation also exists as an object
with forward(x, strength, out) / backward(ctx, dy), which is what the workflow's training step uses.

Implemented: sharpen (hsv=True incl. the S-channel corner-tap quirk, and hsv=False), resample (down + up with any method string
tf.image.resize takes - bilinear, nearest, bicubic, area, lanczos3, lanczos5, gaussian, mitchellcubic - any factor), gaussian (any
odd kernel, any std; the workflow's 5x5 has its own LDS-tiled kernel), residual (hsv=False), awgn (device-side noise), gamma,
median (odd kernels up to 9).
"""
from collections import OrderedDict

import numpy as np
import torch

from .. import ops
from ..device import DeviceArray, to_device, default_device
from . import kernels as hk


class _TapCache(OrderedDict):
    """Device copies of small filter tables keyed by (strength, device), least recently used dropped beyond `limit` entries:
    augmentation draws a fresh strength every step, an unbounded cache would grow by one device allocation per step."""

    def __init__(self, limit=8):
        super().__init__()
        self.limit = limit

    def fetch(self, key, build):
        if key in self:
            self.move_to_end(key)
            return ops.pin(self[key])
        self[key] = value = build()
        while len(self) > self.limit:
            self.popitem(last=False)           # a captured step that replays on the evicted table holds its own reference
        return ops.pin(value)


class Sharpen(object):
    """manipulation_sharpen(x, strength, hsv)  (tf_helpers.py:156-184); hsv=True is the workflow's form (fused rgb->hsv->filter
    ->rgb kernel), hsv=False the plain per-channel 3x3 filter with a SYMMETRIC border."""

    def __init__(self, hsv=True):
        self.hsv = bool(hsv)
        self._cache = _TapCache()

    def taps(self, strength, device):
        # tf.constant(gfilter, tf.float32)
        return self._cache.fetch((float(strength), str(device)), lambda: torch.from_numpy(
            hk.sharpen_kernel(float(strength)).astype(np.float32).reshape(-1)).to(device))

    def forward(self, x, strength=1, out=None, training=False):
        gk = self.taps(strength, x.device)
        if not self.hsv:
            y, mask = ops.dwfilter_fwd(x, gk, 3, 'SYMMETRIC', out=out, clip=True, want_mask=training)
            return y, ({'mask': mask, 'gk': gk} if training else None)
        y, aux, mask = ops.sharpen_fwd(x, gk, out=out, want_aux=training)
        return y, ({'x': x, 'aux': aux, 'mask': mask, 'gk': gk} if training else None)

    def backward(self, ctx, dy):
        if not self.hsv:
            return ops.dwfilter_bwd(dy, ctx['mask'], ctx['gk'], 3, 'SYMMETRIC')
        return ops.sharpen_bwd(ctx['x'], dy, ctx['aux'], ctx['mask'], ctx['gk'])


class Gaussian(object):
    """manipulation_gaussian(x, kernel, std)  (tf_helpers.py:113-125).  kernel = 5 (the workflow's, :114) runs the LDS-tiled 5x5
    kernel; any other odd size the generic per-channel filter (REFLECT border).  An even kernel is refused: tf.pad by kernel//2
    followed by a VALID convolution would return an image one pixel larger than its input."""

    def __init__(self, kernel=5):
        self.kernel = int(kernel)
        if self.kernel < 1 or self.kernel % 2 == 0 or self.kernel > 31:
            raise ValueError('gaussian kernel: an odd size in 1 .. 31, got {}'.format(kernel))
        self._cache = _TapCache()

    def taps(self, std, device):
        return self._cache.fetch((float(std), str(device)), lambda: torch.from_numpy(
            hk.gkern(self.kernel, float(std)).astype(np.float32).reshape(-1)).to(device))

    def forward(self, x, std=0.83, out=None, training=False, skip_clip=False):
        gk = self.taps(std, x.device)
        if self.kernel == 5:
            y, mask = ops.gaussian_fwd(x, gk, out=out, clip=not skip_clip, want_mask=training)
        else:
            y, mask = ops.dwfilter_fwd(x, gk, self.kernel, 'REFLECT', out=out, clip=not skip_clip, want_mask=training)
        return y, ({'mask': mask, 'gk': gk} if training else None)

    def backward(self, ctx, dy):
        if self.kernel == 5:
            return ops.gaussian_bwd(dy, ctx['mask'], ctx['gk'])
        return ops.dwfilter_bwd(dy, ctx['mask'], ctx['gk'], self.kernel, 'REFLECT')


class Residual(object):
    """residual(x, hsv=False) (tf_helpers.py:127-154): the fixed 3x3 high-pass filter, REFLECT border, no clip."""

    def __init__(self):
        self._cache = _TapCache()

    def forward(self, x, out=None, training=False):
        gk = self._cache.fetch(str(x.device), lambda: torch.from_numpy(
            hk.residual_kernel().astype(np.float32).reshape(-1)).to(x.device))
        y, _ = ops.dwfilter_fwd(x, gk, 3, 'REFLECT', out=out, clip=False, want_mask=False)
        return y, ({'gk': gk} if training else None)

    def backward(self, ctx, dy):
        return ops.dwfilter_bwd(dy, None, ctx['gk'], 3, 'REFLECT')


class Resample(object):
    """manipulation_resample(x, factor, method) (tf_helpers.py:68-76): down to floor(H*factor/100) and back up, both dims sized
    from shape[1].  Every method of tf.image.resize is a separable linear map (helpers/kernels.py RESIZE_AXIS_MATRIX), so the
    composition is one banded linear operator M per axis: y = M x M^T, dx = M^T dy M."""

    METHODS = hk.RESIZE_AXIS_MATRIX          # every method string tf.image.resize takes

    def __init__(self, method='bilinear'):
        if method not in self.METHODS:
            raise NotImplementedError('tf.image.resize method {!r}: only {} are built'.format(method, sorted(self.METHODS)))
        self.method = method
        self._cache = {}

    def operator(self, size, factor, device):
        if 0 < factor <= 1:
            factor = 100 * factor
        small = size * int(factor) // 100
        key = (size, small, str(device))
        if key not in self._cache:
            if small < 1:
                raise ValueError('resample factor {} leaves no pixels of a {}-pixel patch'.format(factor, size))
            axis = self.METHODS[self.method]
            m = axis(small, size) @ axis(size, small)
            self._cache[key] = ops.AxisOperator(m, device)
        return self._cache[key]

    def forward(self, x, factor=50, out=None, training=False):
        if x.shape[1] != x.shape[2]:
            raise ValueError('manipulation_resample assumes square patches (tf_helpers.py:73-76)')
        op = self.operator(x.shape[1], factor, x.device)
        tmp = ops.sparse_axis_apply(x, op.fwd, 0, op.out_size)
        y = ops.sparse_axis_apply(tmp, op.fwd, 1, op.out_size, out=out)
        return y, ({'op': op} if training else None)

    def backward(self, ctx, dy):
        op = ctx['op']
        tmp = ops.sparse_axis_apply(dy, op.bwd, 0, op.in_size)
        return ops.sparse_axis_apply(tmp, op.bwd, 1, op.in_size)


_sharpen, _sharpen_rgb, _gaussian, _resample, _residual = Sharpen(), Sharpen(hsv=False), Gaussian(), Resample(), Residual()
_gaussians, _resamples = {5: _gaussian}, {'bilinear': _resample}


def _dev(x):
    x = x.t if isinstance(x, DeviceArray) else x
    return x.device if isinstance(x, torch.Tensor) and x.is_cuda else default_device()


def manipulation_sharpen(x, strength=1, hsv=True):
    return DeviceArray((_sharpen if hsv else _sharpen_rgb).forward(to_device(x, _dev(x)), strength)[0])


def residual(x, hsv=False):
    if hsv:
        raise NotImplementedError('residual(hsv=True) has no caller in the reference')
    return DeviceArray(_residual.forward(to_device(x, _dev(x)))[0])


def manipulation_resample(x, factor=50, method='bilinear'):
    if method not in _resamples:
        _resamples[method] = Resample(method)
    return DeviceArray(_resamples[method].forward(to_device(x, _dev(x)), factor)[0])


def manipulation_gaussian(x, kernel, std, skip_clip=False):
    kernel = int(kernel)
    if kernel not in _gaussians:
        _gaussians[kernel] = Gaussian(kernel)
    return DeviceArray(_gaussians[kernel].forward(to_device(x, _dev(x)), std, skip_clip=skip_clip)[0])


class Awgn(object):
    """manipulation_awgn(x, strength / 255) (tf_helpers.py:79-82; the workflow passes strength/255, workflows/...:122).
    The noise is drawn on the device per call from an explicit generator - seeded per model and PER RANK under data parallelism
    (parallel.rank_generator: the ranks' shards get different noise, rank 0 reproduces the single-process stream) - and kept
    for the backward; a test injects `noise`."""

    def __init__(self, seed=9731):
        self._seed, self._gen = int(seed), None

    def forward(self, x, strength=5.1, out=None, training=False, noise=None):
        s = float(strength) / 255.0
        if noise is None:
            if self._gen is None or self._gen.device != x.device:
                from .. import parallel
                self._gen = parallel.rank_generator(self._seed, x.device)
            noise = torch.randn(x.shape, device=x.device, dtype=x.dtype, generator=self._gen)
        y, mask = ops.awgn_fwd(x, noise, s, out=out, want_mask=training)
        return y, ({'x': x, 'noise': noise, 'mask': mask, 's': s} if training else None)

    def backward(self, ctx, dy):
        return ops.awgn_bwd(ctx['x'], ctx['noise'], dy, ctx['mask'], ctx['s'])


class Gamma(object):
    """manipulation_gamma(x, strength) (tf_helpers.py:85-88)"""

    def forward(self, x, strength=3, out=None, training=False):
        y = ops.gamma_fwd(x, strength, out=out)
        return y, ({'x': x, 'g': float(strength)} if training else None)

    def backward(self, ctx, dy):
        return ops.gamma_bwd(ctx['x'], dy, ctx['g'])


class Median(object):
    """manipulation_median(x, kernel) (tf_helpers.py:91-110): even kernels are bumped to the next odd size."""

    @staticmethod
    def _k(kernel):
        kernel = int(kernel)
        if kernel % 2 == 0:
            kernel += 1
        return max(kernel, 1)

    def forward(self, x, kernel=3, out=None, training=False):
        k = self._k(kernel)
        y, sel = ops.median_fwd(x, k, out=out, want_sel=training)
        return y, ({'sel': sel, 'k': k} if training else None)

    def backward(self, ctx, dy):
        return ops.median_bwd(dy, ctx['sel'], ctx['k'])


def manipulation_awgn(x, strength=0.025):
    return DeviceArray(Awgn().forward(to_device(x, _dev(x)), 255.0 * strength)[0])


def manipulation_gamma(x, strength=2.0):
    return DeviceArray(Gamma().forward(to_device(x, _dev(x)), strength)[0])


def manipulation_median(x, kernel=3):
    return DeviceArray(Median().forward(to_device(x, _dev(x)), kernel)[0])


def mse(a, b):
    """mean((255a - 255b)^2)  (tf_helpers.py:31-32)"""
    d = _dev(a)
    return DeviceArray(ops.mse255(to_device(a, d), to_device(b, d))[0])


def mae(a, b):
    """mean(|255a - 255b|)  (tf_helpers.py:35-36)"""
    d = _dev(a)
    return DeviceArray(ops.mae255(to_device(a, d), to_device(b, d))[0])


def ssim_loss(a, b):
    """mean(255 (1 - tf.image.ssim(a, b, 1.0)))  (tf_helpers.py:39-40)"""
    d = _dev(a)
    return DeviceArray(ops.ssim_loss(to_device(a, d), to_device(b, d))[0])


def msssim_loss(a, b):
    """mean(255 (1 - tf.image.ssim_multiscale(a, b, 1.0)))  (tf_helpers.py:43-44)"""
    d = _dev(a)
    return DeviceArray(ops.msssim_loss(to_device(a, d), to_device(b, d))[0])
