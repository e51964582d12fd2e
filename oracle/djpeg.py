"""
Differentiable JPEG restated.  TEST INFRASTRUCTURE - see oracle/__init__.py.

Follows models/jpeg.py:47-159 (DifferentiableJPEG) and models/jpeg.py:202-251 (JPEG.process
quality resolution).  Two implementations of the same arithmetic:

  * djpeg_torch  - torch (float64 by default), differentiable through autograd; the 'soft'
                   rounding uses the custom backward of oracle.tfops._SoftRound (layers.py:126-128)
  * djpeg_numpy_fwd / djpeg_numpy_bwd - numpy float64 forward + hand-derived backward, an
                   independent second opinion for the gradient (the HIP backward kernel is
                   checked against both).

Reference quirks kept on purpose: level shift 127 (jpeg.py:105,154), the 4-decimal
DCT matrix (jpeg.py:78-85), [Y.., Cb.., Cr..] per-image Q tiling without chroma subsampling
(jpeg.py:125-128), hard output clip with zero gradient outside [0,1] (jpeg.py:157).
"""
import math

import numpy as np
import torch

from . import tables
from .tfops import quantization


def _blocks(x):          # (N,H,W,3) -> (N,3,H/8,W/8,8,8)
    n, h, w, c = x.shape
    return x.reshape(n, h // 8, 8, w // 8, 8, c).permute(0, 5, 1, 3, 2, 4)


def _unblocks(b):        # inverse of _blocks
    n, c, hb, wb, _, _ = b.shape
    return b.permute(0, 2, 4, 3, 5, 1).reshape(n, hb * 8, wb * 8, c)


def qtables_torch(quality, dtype=torch.float64):
    ql = torch.tensor(tables.jpeg_qtable(quality, 0), dtype=dtype)
    qc = torch.tensor(tables.jpeg_qtable(quality, 1), dtype=dtype)
    return torch.stack([ql, qc, qc])          # per channel (Y, Cb, Cr)


def djpeg_torch(x, quality=None, mode='soft', q=None):
    """x: (N,H,W,3) in [0,1]. Returns (y, X_dequantised, idx) where idx = quantised X/Q (the integer index
    tensor when mode is 'soft'/'round')."""
    dt = x.dtype
    if q is None:
        q = qtables_torch(quality, dt) if quality is not None else torch.ones(3, 8, 8, dtype=dt)
    cf = torch.tensor(tables.COLOR_F, dtype=dt)
    ci = torch.tensor(tables.COLOR_I, dtype=dt)
    Fm = torch.tensor(tables.DCT_F, dtype=dt)
    # rgb -> ycbcr : 1x1 conv over [1, 255*r, 255*g, 255*b]   (jpeg.py:99-100)
    xc = torch.cat([torch.ones_like(x[..., :1]), 255.0 * x], dim=-1)
    ycbcr = xc @ cf.t()
    b = _blocks(ycbcr - 127)                                   # jpeg.py:105-114
    X = Fm @ b @ Fm.t()                                        # jpeg.py:118-119
    Qb = q.reshape(1, 3, 1, 1, 8, 8)
    idx = quantization(X / Qb, mode)                           # jpeg.py:129-130
    Xd = idx * Qb                                              # jpeg.py:131
    xi = Fm.t() @ Xd @ Fm                                      # jpeg.py:135-136
    qq = _unblocks(xi)
    qc = torch.cat([torch.ones_like(qq[..., :1]), qq + 127], dim=-1)
    y = (qc @ ci.t()) / 255.0                                  # jpeg.py:154-156
    y = torch.clamp(y, 0, 1)                                   # hard clip, jpeg.py:157
    return y, Xd, idx


def resolve_quality(quality, default=None, rng=np.random):
    """JPEG.process quality resolution (models/jpeg.py:210-225)."""
    quality = default if quality is None else quality
    is_num = isinstance(quality, (int, float, np.integer, np.floating)) and not isinstance(quality, bool)
    if is_num and 1 <= quality <= 100:
        return int(quality)
    if hasattr(quality, '__getitem__') and len(quality) > 1 and all(1 <= v <= 100 for v in quality):
        if len(quality) > 2:
            return int(rng.choice(quality))
        return int(rng.randint(quality[0], quality[1]))
    raise ValueError('Invalid or unspecified JPEG quality!')


# ---------------------------------------------------------------------------------------------
# numpy float64, hand-derived backward (independent of autograd)
def djpeg_numpy_fwd(x, q, mode='soft'):
    """x (N,H,W,3) float, q (3,8,8). Returns y, cache."""
    x = np.asarray(x, np.float64)
    n, h, w, _ = x.shape
    cf = tables.COLOR_F.astype(np.float64)
    ci = tables.COLOR_I.astype(np.float64)
    Fm = tables.DCT_F.astype(np.float64)
    ycc = 255.0 * x @ cf[:, 1:].T + cf[:, 0]
    b = (ycc - 127).reshape(n, h // 8, 8, w // 8, 8, 3).transpose(0, 5, 1, 3, 2, 4)
    X = Fm @ b @ Fm.T
    u = X / q.reshape(1, 3, 1, 1, 8, 8)
    if mode in ('soft', 'round'):
        r = np.rint(u)
    elif mode == 'sin':
        r = u - np.sin(2 * np.pi * u) / (2 * np.pi)
    elif mode == 'harmonic':
        r = u - np.sin(2 * np.pi * u) / np.pi
    elif mode == 'identity':
        r = u
    else:
        raise ValueError(mode)
    Xd = r * q.reshape(1, 3, 1, 1, 8, 8)
    xi = Fm.T @ Xd @ Fm
    qq = xi.transpose(0, 2, 4, 3, 5, 1).reshape(n, h, w, 3) + 127
    ypre = (qq @ ci[:, 1:].T + ci[:, 0]) / 255.0
    y = np.clip(ypre, 0, 1)
    return y, dict(u=u, ypre=ypre, q=q, mode=mode, idx=r)


def djpeg_numpy_bwd(gy, cache):
    """Gradient wrt x given gy = dL/dy."""
    u, ypre, q, mode = cache['u'], cache['ypre'], cache['q'], cache['mode']
    n, h, w, _ = gy.shape
    cf = tables.COLOR_F.astype(np.float64)
    ci = tables.COLOR_I.astype(np.float64)
    Fm = tables.DCT_F.astype(np.float64)
    g = np.where((ypre >= 0) & (ypre <= 1), gy, 0.0) / 255.0        # tf.clip_by_value grad: pass inside [0,1]
    gq = g @ ci[:, 1:]                                               # d/d(qq)
    gb = gq.reshape(n, h // 8, 8, w // 8, 8, 3).transpose(0, 5, 1, 3, 2, 4)
    gXd = Fm @ gb @ Fm.T                                             # xi = F^T Xd F  => dXd = F g F^T
    if mode == 'soft' or mode == 'sin':
        dr = 1 - np.cos(2 * np.pi * u)
    elif mode == 'harmonic':
        dr = 1 - 2 * np.cos(2 * np.pi * u)
    elif mode == 'identity':
        dr = np.ones_like(u)
    elif mode == 'round':
        dr = np.zeros_like(u)
    gX = gXd * dr                                                    # (xQ then /Q cancel)
    gblk = Fm.T @ gX @ Fm                                            # X = F b F^T => db = F^T g F
    gycc = gblk.transpose(0, 2, 4, 3, 5, 1).reshape(n, h, w, 3)
    return 255.0 * (gycc @ cf[:, 1:])
