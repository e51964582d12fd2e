"""
Oracle self-consistency (CPU, seconds): two independent restatements must agree, analytic known answers hold
(SURVEY 8c "Analytic KATs"), and the C float32 dJPEG agrees with the float64 one except at rounding ties.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import djpeg as odj
from oracle import manip as om
from oracle import nets as onets
from oracle import tables as ot
from oracle import tfops as T
from oracle import workflow as owf

from util import natural_images, to64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clib():
    path = os.path.join(ROOT, 'oracle', 'libdjpeg_ref.so')
    if not os.path.isfile(path):
        pytest.skip('oracle/libdjpeg_ref.so not built (make -C oracle)')
    return ctypes.CDLL(path)


def _q3(q):
    return np.stack([ot.jpeg_qtable(q, 0), ot.jpeg_qtable(q, 1), ot.jpeg_qtable(q, 1)]).astype(np.float32)


def c_djpeg(x, q):
    lib = _clib()
    x = np.ascontiguousarray(x, np.float32)
    n, h, w, _ = x.shape
    y = np.zeros_like(x)
    idx = np.zeros((n, 3, h // 8, w // 8, 8, 8), np.int16)
    xd = np.zeros((n, 3, h // 8, w // 8, 8, 8), np.float32)
    qq = _q3(q)
    rc = lib.djpeg_ref_forward(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                               qq.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p),
                               xd.ctypes.data_as(ctypes.c_void_p), n, h, w)
    assert rc == 0
    return y, idx, xd


@pytest.mark.parametrize('q', [10, 50, 95])
def test_c_djpeg_matches_float64(q):
    x = natural_images(2, 64, 64, seed=q)
    y, idx, _ = c_djpeg(x, q)
    y64, _, idx64 = odj.djpeg_torch(to64(x), q, 'soft')
    assert np.abs(y - y64.numpy()).max() < 2e-6
    mism = (idx != idx64.numpy()).mean()
    assert mism < 1e-3, 'index mismatches vs float64 should only happen at rounding ties: {}'.format(mism)


def test_djpeg_numpy_vs_torch_forward_backward():
    x = torch.rand(1, 16, 24, 3, dtype=torch.float64, requires_grad=True)
    for mode in ('soft', 'sin', 'harmonic', 'identity'):
        x.grad = None
        y, _, _ = odj.djpeg_torch(x, 50, mode)
        gy = torch.rand_like(y)
        (y * gy).sum().backward()
        yn, cache = odj.djpeg_numpy_fwd(x.detach().numpy(), _q3(50).astype(np.float64), mode)
        gn = odj.djpeg_numpy_bwd(gy.numpy(), cache)
        assert np.abs(yn - y.detach().numpy()).max() < 1e-12
        assert np.abs(gn - x.grad.numpy()).max() < 1e-9


def test_djpeg_analytic_kats():
    x = torch.rand(1, 32, 32, 3, dtype=torch.float64)
    # Q == 1 and identity rounding is NOT an identity: the 4-decimal DCT matrix is not orthonormal
    y, _, _ = odj.djpeg_torch(x, None, 'identity')
    d = (y - x).abs().max().item()
    assert 1e-5 < d < 1e-3
    # soft forward == round forward
    ys, _, i1 = odj.djpeg_torch(x, 50, 'soft')
    yr, _, i2 = odj.djpeg_torch(x, 50, 'round')
    assert torch.equal(ys, yr) and torch.equal(i1, i2)
    assert float((i1 - i1.round()).abs().max()) == 0.0
    F = ot.DCT_F.astype(np.float64)
    assert 1e-4 < np.abs(F @ F.T - np.eye(8)).max() < 5e-4


def test_depth_to_space_is_dcr():
    x = torch.arange(2 * 3 * 4 * 8, dtype=torch.float64).reshape(2, 3, 4, 8)
    y = T.depth_to_space(x, 2)
    for (n, yy, xx, i, j, c) in [(0, 1, 2, 0, 1, 1), (1, 2, 3, 1, 0, 0), (1, 0, 0, 1, 1, 1)]:
        assert y[n, yy * 2 + i, xx * 2 + j, c] == x[n, yy, xx, (i * 2 + j) * 2 + c]
    assert torch.equal(T.space_to_depth(y, 2), x)


def test_tf_same_padding_and_resize_tables():
    assert T.same_pads(256, 5, 2) == (1, 2) and T.same_pads(128, 3, 1) == (1, 1) and T.same_pads(7, 5, 2) == (2, 2)
    lo, hi, t = T.resize_axis_table(256, 128)
    assert (lo == 2 * np.arange(128)).all() and (hi == lo + 1).all() and np.allclose(t, 0.5)
    lo, hi, t = T.resize_axis_table(128, 256)
    assert lo[0] == 0 and hi[0] == 0 and np.isclose(t[1], 0.25) and np.isclose(t[2], 0.75)
    x = torch.rand(1, 8, 8, 3, dtype=torch.float64)
    ref = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(5, 5), mode='bilinear', align_corners=False)
    assert (T.resize_bilinear(x, 5, 5) - ref.permute(0, 2, 3, 1)).abs().max() < 1e-12


def test_constrained_kernel_kats():
    k = torch.tensor(ot.fan_residual_init()) + 0.01 * torch.rand(5, 5, 3, 3, dtype=torch.float64)
    m = torch.tensor(ot.center_mask_2dfilter(5, 3))
    nf = T.constrained_kernel(k, m)
    assert torch.allclose(nf.sum(dim=(0, 1, 2)), torch.zeros(3, dtype=torch.float64), atol=1e-9)
    for i in range(3):
        assert nf[2, 2, i, i] == -100
    out = T.constrained_conv(torch.full((1, 16, 16, 3), 0.37, dtype=torch.float64), k, m)
    assert out.abs().max() < 1e-10


def test_ce_entropy_kats():
    p = torch.full((4, 5), 0.2, dtype=torch.float64)
    assert abs(float(T.sparse_ce_from_probs(p, [0, 1, 2, 3])) - np.log(5)) < 1e-12
    cb = torch.tensor(ot.codebook(5))
    assert float(T.entropy(torch.zeros(1000), cb)[0]) < 1e-5
    vals = cb.repeat(100)
    assert abs(float(T.entropy(vals, cb)[0]) - 5.0) < 1e-3


def test_hsv_roundtrip_and_sharpen_quirk():
    x = torch.rand(2, 12, 12, 3, dtype=torch.float64)
    assert (T.hsv_to_rgb(T.rgb_to_hsv(x)) - x).abs().max() < 1e-12
    import colorsys
    v = x[0, 3, 4].tolist()
    ref = colorsys.rgb_to_hsv(*v)
    assert np.allclose(T.rgb_to_hsv(x)[0, 3, 4].numpy(), ref, atol=1e-12)
    # strength 0 => H,V untouched, S shifted by (+1,+1) with SYMMETRIC edge handling
    y0 = om.manipulation_sharpen(x, 0.0)
    hsv = T.rgb_to_hsv(T.pad2d(x, 1, 'SYMMETRIC'))
    shifted = torch.stack([hsv[:, 1:-1, 1:-1, 0], hsv[:, 2:, 2:, 1], hsv[:, 1:-1, 1:-1, 2]], dim=-1)
    assert (y0 - torch.clamp(T.hsv_to_rgb(shifted), 0, 1)).abs().max() < 1e-12


def test_hsv_known_answers_against_colorsys():
    """Third-party KAT for oracle/tfops.py rgb_to_hsv / hsv_to_rgb (tf.image.rgb_to_hsv, helpers/tf_helpers.py:133-160): the
    standard library's colorsys on every pixel of a random batch plus the branch corners (grey, black, white, pure and
    two-channel-tied colours, where tf's and colorsys' selection order could differ)."""
    import colorsys
    rng = np.random.default_rng(5)
    px = rng.random((400, 3))
    corners = [(0, 0, 0), (1, 1, 1), (.5, .5, .5), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1),
               (.3, .3, .7), (.7, .3, .3), (.3, .7, .3), (.7, .7, .3), (.3, .7, .7), (.7, .3, .7), (.2, .2, .2000001)]
    px = np.concatenate([px, np.array(corners, np.float64)])
    hsv = T.rgb_to_hsv(torch.tensor(px)).numpy()
    want = np.array([colorsys.rgb_to_hsv(*p) for p in px])
    assert np.abs(hsv - want).max() < 1e-12
    back = T.hsv_to_rgb(torch.tensor(want)).numpy()
    want_rgb = np.array([colorsys.hsv_to_rgb(*p) for p in want])
    assert np.abs(back - want_rgb).max() < 1e-12 and np.abs(back - px).max() < 1e-12
    # hue wheel: colorsys.hsv_to_rgb over a grid of (h, s, v), including h = 0 and the sector borders k / 6
    grid = np.array([(h, s, v) for h in np.linspace(0, 1, 25)[:-1] for s in (0., .25, 1.) for v in (0., .6, 1.)])
    assert np.abs(T.hsv_to_rgb(torch.tensor(grid)).numpy() - np.array([colorsys.hsv_to_rgb(*p) for p in grid])).max() < 1e-12


def test_pad_modes_against_numpy_and_explicit_indices():
    """tf.pad SYMMETRIC (edge repeated: helpers/tf_helpers.py:84,151) vs REFLECT (edge not repeated: tf_helpers.py:110,
    pipelines.py:276): explicit index tables for the 1-D case and numpy's np.pad (same two definitions) on images."""
    v = torch.arange(5, dtype=torch.float64).view(1, 5, 1, 1).expand(1, 5, 5, 1).contiguous()
    assert T.pad2d(v, 2, 'SYMMETRIC')[0, :, 2, 0].tolist() == [1, 0, 0, 1, 2, 3, 4, 4, 3]
    assert T.pad2d(v, 2, 'REFLECT')[0, :, 2, 0].tolist() == [2, 1, 0, 1, 2, 3, 4, 3, 2]
    rng = np.random.default_rng(11)
    x = rng.random((2, 7, 9, 3))
    for p in (1, 2, 3, 5):
        for mode, npmode in (('SYMMETRIC', 'symmetric'), ('REFLECT', 'reflect')):
            want = np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)), mode=npmode)
            assert np.array_equal(T.pad2d(torch.tensor(x), p, mode).numpy(), want), (mode, p)


def test_keras_adam_first_step():
    p = [torch.ones(3, dtype=torch.float64)]
    g = [torch.tensor([0.5, -2.0, 0.0], dtype=torch.float64)]
    m, v = [torch.zeros(3, dtype=torch.float64)], [torch.zeros(3, dtype=torch.float64)]
    T.adam_step(p, g, m, v, 1, 1e-3)
    # first step: m_hat/sqrt(v_hat) = sign(g) up to epsilon placement
    assert torch.allclose(p[0], torch.tensor([1 - 1e-3, 1 + 1e-3, 1.0], dtype=torch.float64), atol=1e-8)


def test_param_counts_and_workflow_step():
    assert onets.count_params(onets.unet_init()) == 7763820
    assert onets.count_params(onets.fan_init(5)) == 1145382
    assert onets.count_params(onets.dcn_init()) == 2533293
    wf = owf.Workflow()
    rgb = natural_images(2, 64, 64, seed=1)
    from util import bayer_from_rgb
    bx, by = to64(bayer_from_rgb(rgb)), to64(rgb)
    l0, parts = wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)
    l1, _ = wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)
    assert np.isfinite(l0) and l1 < l0 and abs(parts['ce'] - np.log(5)) < 0.5


def test_gradcheck_manipulations():
    """Finite-difference check of the oracle's own backward (it is the reference for the HIP backward kernels)."""
    torch.manual_seed(0)
    x = (0.2 + 0.6 * torch.rand(1, 8, 8, 3, dtype=torch.float64)).requires_grad_(True)
    for fn in (lambda t: om.manipulation_gaussian(t, 5, 0.83), lambda t: om.manipulation_resample(t, 50),
               lambda t: om.manipulation_sharpen(t, 1.0)):
        assert torch.autograd.gradcheck(fn, (x,), eps=1e-6, atol=1e-5, nondet_tol=0.0)


def test_oracle_ssim_flavours():
    """The two SSIM restatements agree with a direct evaluation of the definition on one window, are 1 for identical
    images and decrease with noise."""
    rng = np.random.RandomState(3)
    a = rng.rand(7, 7, 1)
    b = np.clip(a + 0.1 * rng.randn(7, 7, 1), 0, 1)
    x, y = a[..., 0].ravel(), b[..., 0].ravel()
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    vx, vy, vxy = x.var(ddof=1), y.var(ddof=1), np.cov(x, y, ddof=1)[0, 1]
    direct = (2 * x.mean() * y.mean() + c1) * (2 * vxy + c2) / ((x.mean() ** 2 + y.mean() ** 2 + c1) * (vx + vy + c2))
    assert abs(T.ssim_skimage(a, b) - direct) < 1e-12
    img = rng.rand(2, 24, 24, 3)
    noisy = np.clip(img + 0.05 * rng.randn(*img.shape), 0, 1)
    s_same = T.ssim_tf(torch.from_numpy(img), torch.from_numpy(img))
    s_noisy = T.ssim_tf(torch.from_numpy(img), torch.from_numpy(noisy))
    assert torch.allclose(s_same, torch.ones(2, dtype=torch.float64)) and bool((s_noisy < 0.99).all())
    assert abs(T.ssim_skimage(img[0], img[0]) - 1.0) < 1e-12 and T.ssim_skimage(img[0], noisy[0]) < 0.99


def test_oracle_msssim_known_answers():
    """ssim_multiscale: 1 for identical images, symmetric, equal to the hand-combined per-scale statistics (weights 0.0448 ..
    0.1333, cs of scales 0-3, ssim of scale 4), a single-scale ssim_tf consistent with its per-channel parts."""
    import torch.nn.functional as F
    rng = np.random.RandomState(8)
    a = torch.from_numpy(natural_images(2, 176, 192, seed=3).astype(np.float64))
    b = (a + 0.06 * torch.from_numpy(rng.randn(*a.shape))).clamp(0, 1)
    assert torch.allclose(T.ssim_multiscale(a, a), torch.ones(2, dtype=torch.float64))
    ms = T.ssim_multiscale(a, b)
    assert torch.allclose(ms, T.ssim_multiscale(b, a)) and bool((ms < 0.999).all()) and bool((ms > 0).all())
    ssim_pc, cs_pc = T._ssim_per_channel(a, b)
    assert torch.allclose(ssim_pc.mean(dim=1), T.ssim_tf(a, b))
    vals, x, y = [], a, b
    for k in range(5):
        if k:
            x = F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            y = F.avg_pool2d(y.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        s_, c_ = T._ssim_per_channel(x, y)
        vals.append(s_ if k == 4 else c_)
    want = torch.ones_like(vals[0])
    for v, w in zip(vals, T.MSSSIM_WEIGHTS):
        want = want * torch.relu(v) ** w
    assert torch.allclose(ms, want.mean(dim=1))
    assert abs(float(T.msssim_loss255(a, b)) - float((255 * (1 - ms)).mean())) < 1e-12


def test_oracle_classic_isp_known_answers():
    """classic_isp_forward: with alpha = 0 (or no CNN) the residual ISP is bilinear demosaicing -> colour matrix -> gamma of
    the clipped value; the gamma stage keeps a gradient below 1/255 (straight-through clip)."""
    from oracle import tables as ot2
    from util import stack_bayer
    from neural_imaging_amd.helpers import kernels as hk
    col = np.array([0.8, 0.35, 0.002])                      # a flat colour: bilinear demosaicing must return it everywhere
    rgb = np.broadcast_to(col, (1, 16, 16, 3)).astype(np.float32)
    raw = stack_bayer(rgb).astype(np.float64)
    p = {'up/kernel': to64(hk.upsampling_kernel('gbrg').reshape(1, 1, 4, 12)),
         'bilinear/kernel': to64(np.asarray(hk.bilin_kernel(5)).reshape(5, 5, 3, 3)),
         'srgb/kernel': to64(np.eye(3).reshape(1, 1, 3, 3)), 'demosaicing/alpha': to64(np.array([0.1]))}
    y = onets.classic_isp_forward(p, to64(raw), residual=True)
    want = np.maximum(col.astype(np.float32).astype(np.float64), 1 / 255) ** (1 / 2.2)
    assert y.shape == (1, 16, 16, 3) and np.abs(y.numpy()[:, 2:-2, 2:-2] - want).max() < 1e-6      # REFLECT border aside
    x = torch.tensor([0.0, 1e-3, 0.5, 2.0], dtype=torch.float64, requires_grad=True)
    g = x + (torch.clamp(x, 1 / 255, 1) - x).detach()
    out = torch.pow(g, 1 / 2.2)
    out.sum().backward()
    assert float(x.grad[0]) > 1.0 and abs(float(x.grad[3]) - 1 / 2.2) < 1e-12       # clipped values keep d/dx of pow


def test_oracle_datafeed_policy_consistency():
    """oracle/datafeed.select over the candidate list that sample_patch itself drew gives sample_patch's answer (the two share
    the Policy the device kernel is checked against), and cut_batch normalises like dataset.py:124-126."""
    from oracle import datafeed as odf
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'datafeed_sample_patch.npz'))
    img = d['images'][0]

    class Rec(object):                      # records the draws of one sample_patch call
        def __init__(self, seed):
            self.r = np.random.RandomState(seed); self.xy = []; self.u = []; self._pend = []
        def randint(self, lo, hi):
            v = self.r.randint(lo, hi); self._pend.append(v)
            if len(self._pend) == 2:
                self.xy.append((2 * (self._pend[0] // 2), 2 * (self._pend[1] // 2))); self.u.append(0.75); self._pend = []
            return v
        def uniform(self):
            v = self.r.uniform(); self.u[-1] = v
            return v
    for mode in ('flat', 'flat-aggressive', 'dark-n-textured'):
        for seed in range(6):
            rec = Rec(seed)
            want = odf.sample_patch(img, 32, mode, 6, rng=rec)
            got, used = odf.select(img, rec.xy, rec.u, 32, mode, 6)
            assert got == tuple(want) and used == len(rec.xy), (mode, seed)
    raw = np.arange(2 * 8 * 8 * 4, dtype=np.uint16).reshape(2, 8, 8, 4) * 100
    rgb = (np.arange(2 * 16 * 16 * 3) % 256).astype(np.uint8).reshape(2, 16, 16, 3)
    x, y = odf.cut_batch(raw, rgb, [1, 0], [(2, 4), (0, 0)], 8)
    assert x.dtype == np.float32 and np.array_equal(x[0], (raw[1, 2:6, 1:5] / 65535.0).astype(np.float32))
    assert np.array_equal(y[1], (rgb[0, :8, :8] / 255.0).astype(np.float32))


def test_stride2_conv_is_a_conv_over_the_space_to_depth_image():
    """The identity behind the throughput-mode form of the codec's strided layers (models/compression.py:217-229): a 5x5 stride-2
    TF-SAME convolution over an even-sized image equals a 3x3 stride-1 SAME convolution over its space-to-depth image with the
    re-arranged kernel - forward and, through autograd, both gradients (the kernel gradient maps back by the transposed gather)."""
    g = torch.Generator().manual_seed(5)
    for (n, h, w, cin, cout, cp) in ((2, 16, 24, 3, 8, 16), (1, 8, 8, 5, 4, None)):
        x = torch.randn((n, h, w, cin), dtype=torch.float64, generator=g).requires_grad_(True)
        w5 = torch.randn((5, 5, cin, cout), dtype=torch.float64, generator=g).requires_grad_(True)
        b = torch.randn((cout,), dtype=torch.float64, generator=g)
        ref = T.conv2d(x, w5, b, 2, 'SAME')
        got = T.conv2d(T.space_to_depth2(x, cp), T.s2d_conv_weights(w5, cp), b, 1, 'SAME')
        assert ref.shape == got.shape == (n, h // 2, w // 2, cout) and float((ref - got).abs().max()) < 1e-12
        dy = torch.randn(ref.shape, dtype=torch.float64, generator=g)
        gx, gw = torch.autograd.grad((ref * dy).sum(), (x, w5))
        gx2, gw2 = torch.autograd.grad((got * dy).sum(), (x, w5))
        assert float((gx - gx2).abs().max()) < 1e-12 and float((gw - gw2).abs().max()) < 1e-12



def test_codec_channel_gradient_conditioning_under_bf16_weights():
    """What a bf16-operand implementation can be asked for on configs[4] at random initialisation (the yardstick behind the UNet
    floors of tests/test_gpu_models.py::test_learned_codec_channel_at_the_full_patch_size).  The float64 oracle itself, exact
    arithmetic throughout, evaluated once with its weights as they are and once with every weight rounded to bf16: the codec's l2
    term dominates the loss, the UNet's deep levels only see the nearly cancelling low-frequency part of the gradient image, and a
    coherent 2^-9 perturbation of the linear maps in front of it leaves little of their direction (levels 4 - 5: cosine < 0.5) while
    the FAN / codec kernels and UNet level 1 keep theirs.  The product keeps float32 master weights and rounds operands only; it
    measures 0.36 - 0.69 on the same tensors."""
    from util import bayer_from_rgb
    rgb = natural_images(1, 256, 256, seed=9)
    raw = bayer_from_rgb(rgb)
    ref = owf.Workflow(codec='dcn', trainable=('nip', 'dcn'))
    names = list(ref.fan.keys()) + list(ref.nip.keys()) + list(ref.dcn.keys())
    g0 = dict(zip(names, [g.numpy().copy() for g in ref.loss_and_grads(to64(raw), to64(rgb), 0.1, 0.01)[3]]))
    rb = lambda v: v.to(torch.bfloat16).to(torch.float64)
    r2 = owf.Workflow(codec='dcn', trainable=('nip', 'dcn'))
    r2.nip = onets.OrderedDict((k, rb(v)) for k, v in ref.nip.items())
    r2.fan = onets.OrderedDict((k, rb(v)) for k, v in ref.fan.items())
    r2.dcn = onets.OrderedDict((k, rb(v)) for k, v in ref.dcn.items())
    g2 = dict(zip(names, [g.numpy() for g in r2.loss_and_grads(to64(raw), to64(rgb), 0.1, 0.01)[3]]))
    cos = {}
    for k in names:
        if k.endswith('/kernel'):
            a, b = g2[k].ravel(), g0[k].ravel()
            cos[k.split('/')[0]] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
    assert all(cos[k] < 0.5 for k in ('ec51', 'ec52', 'dct1', 'ec41')), cos
    assert all(cos[k] > 0.9 for k in ('dc42', 'ec11', 'conv3', 'conv4', 'dense', 'e2', 'er2a', 'd256')), cos


def test_resize_method_matrices_known_answers():
    """The axis matrices of tf.image.resize's other methods (oracle/tfops.py RESIZE_AXIS) against values that follow from the
    kernels' definitions by hand, and against the product's independent builder (helpers/kernels.py)."""
    import importlib
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd.helpers import kernels as hk
    # Keys cubic a = -0.5 half-way between two pixels: (-1/16, 9/16, 9/16, -1/16); at the border the outside tap is dropped and
    # the remaining three are renormalised by 1 / (1 + 1/16)
    m = T.resize_bicubic_axis(16, 8)
    assert np.allclose(m[3, 5:9], [-0.0625, 0.5625, 0.5625, -0.0625], atol=1e-7)
    assert np.allclose(m[0, :3], np.array([0.5625, 0.5625, -0.0625]) / 1.0625, atol=1e-7)
    assert np.array_equal(T.resize_bicubic_axis(9, 9), np.eye(9))
    # area: 2:1 is the mean of two pixels, 1:2 repeats each pixel, 3:2 mixes thirds
    assert np.array_equal(T.resize_area_axis(8, 4), np.kron(np.eye(4), [[0.5, 0.5]]))
    assert np.array_equal(T.resize_area_axis(4, 8), np.kron(np.eye(4), [[1.0], [1.0]]))
    assert np.allclose(T.resize_area_axis(3, 2), [[2 / 3, 1 / 3, 0], [0, 1 / 3, 2 / 3]], atol=1e-6)
    # lanczos3 two-to-one WITHOUT antialiasing: taps at +-0.5, 1.5, 2.5 of L(x) = 3 sin(pi x) sin(pi x / 3) / (pi x)^2, normalised
    lz = lambda x: 3 * np.sin(np.pi * x) * np.sin(np.pi * x / 3) / (np.pi * x) ** 2
    w = np.array([lz(2.5), lz(1.5), lz(0.5), lz(0.5), lz(1.5), lz(2.5)])
    assert np.allclose(T.resize_scale_translate_axis(32, 16, 'lanczos3')[7, 12:18], w / w.sum(), atol=1e-6)
    # gaussian (sigma 0.5, radius 1.5) and Mitchell-Netravali at the same size are NOT the identity: taps at -1, 0, 1
    gw = np.array([np.exp(-2.0), 1.0, np.exp(-2.0)])
    assert np.allclose(T.resize_scale_translate_axis(12, 12, 'gaussian')[5, 4:7], gw / gw.sum(), atol=1e-6)
    assert np.allclose(T.resize_scale_translate_axis(12, 12, 'mitchellcubic')[5, 4:7], [1 / 18, 8 / 9, 1 / 18], atol=1e-6)
    for name, axis in T.RESIZE_AXIS.items():
        for i, o in ((256, 128), (128, 256), (64, 46), (46, 64), (10, 3), (3, 10), (200, 31), (7, 7), (1, 4), (4, 1)):
            a = axis(i, o)
            assert np.allclose(a.sum(1), 1.0, atol=3e-6), (name, i, o)             # every output is a weighted mean
            assert np.abs(a - hk.RESIZE_AXIS_MATRIX[name](i, o)).max() < 1e-6, (name, i, o)
    # the manipulation (tf_helpers.py:68-76): constants stay constants, and the method reaches both resizes
    x = torch.full((1, 12, 12, 3), 0.25, dtype=torch.float64)
    for name in T.RESIZE_AXIS:
        assert torch.allclose(om.manipulation_resample(x, 50, name), x, atol=1e-6), name
    ramp = torch.linspace(0, 1, 12, dtype=torch.float64).view(1, 1, 12, 1).expand(1, 12, 12, 3)
    assert not torch.allclose(om.manipulation_resample(ramp, 50, 'area'), om.manipulation_resample(ramp, 50, 'bicubic'), atol=1e-3)
