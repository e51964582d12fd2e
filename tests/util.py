"""Shared helpers for the parity tests (oracle on CPU float64 vs HIP kernels on the GPU)."""
import numpy as np
import torch


def natural_images(n, h, w, seed=0):
    """Natural-image-like synthetic RGB in [0,1]: low-pass blocks + ramp + noise, quantised to k/255."""
    rng = np.random.default_rng(seed)
    base = rng.random((n, h // 8 + 1, w // 8 + 1, 3))
    img = np.kron(base, np.ones((1, 8, 8, 1)))[:, :h, :w, :]
    ramp = np.linspace(0, 1, w)[None, None, :, None]
    img = 0.7 * img + 0.3 * ramp + rng.normal(0, 0.03, size=(n, h, w, 3))
    return (np.round(np.clip(img, 0, 1) * 255) / 255).astype(np.float32)


def bayer_from_rgb(rgb):
    """Position-ordered Bayer stack (N,h/2,w/2,4) of an RGB batch on a GBRG mosaic: planes taken at (0,0), (0,1), (1,0), (1,1)
    = G, B, R, G.  A synthetic RAW input for the learned pipelines (which do not care about the plane order); the parity
    tolerances of the workflow tests were tuned on it.  The reference's own order is stack_bayer() below."""
    g1 = rgb[:, 0::2, 0::2, 1]
    b = rgb[:, 0::2, 1::2, 2]
    r = rgb[:, 1::2, 0::2, 0]
    g2 = rgb[:, 1::2, 1::2, 1]
    return np.stack([g1, b, r, g2], axis=-1).astype(np.float32)


def stack_bayer(rgb):
    """RGGB-ordered stack of a GBRG mosaic like the reference's helpers/raw.py:204-225 `stack_bayer(image, 'GBRG')`: planes
    (R, G1, G2, B) taken at (1,0), (0,0), (1,1), (0,1) - what upsampling_kernel('gbrg') puts back in place."""
    return np.stack([rgb[:, 1::2, 0::2, 0], rgb[:, 0::2, 0::2, 1], rgb[:, 1::2, 1::2, 1], rgb[:, 0::2, 1::2, 2]],
                    axis=-1).astype(np.float32)


def to64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0, 0.0
    d = np.abs(a - b)
    return float(d.max()), float(d.max() / (np.abs(b).max() + 1e-30))


def assert_close(a, b, atol, rtol_max=None, what=''):
    """|a-b| <= atol  OR  (optionally) max|a-b| <= rtol_max * max|b| (gradient tensors have arbitrary scale)."""
    mx, rel = err(a, b)
    ok = mx <= atol or (rtol_max is not None and rel <= rtol_max)
    assert ok, '{}: max abs err {:.3e}, rel-to-max {:.3e} (atol {}, rtol_max {})'.format(what, mx, rel, atol, rtol_max)
    return mx, rel
