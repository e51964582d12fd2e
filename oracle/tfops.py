"""
TensorFlow-2.1 op semantics restated on torch-CPU tensors (NHWC at the interface).
TEST INFRASTRUCTURE - see oracle/__init__.py.

PARITY UNPINNED: these ops' arithmetic lives in tensorflow-gpu==2.1.2 (requirements.txt:2),
which is absent from /root/reference and cannot be installed here.  Semantics follow the TF
documentation / kernels as listed in SURVEY.md section 7 "hard parts":
  * SAME padding is asymmetric for strided convs (extra pad goes after)
  * depth_to_space is DCR, Conv2DTranspose kernels are (kh,kw,Cout,Cin)
  * tf.round is round-half-to-even; tf.image.resize uses half-pixel centres, no antialias
  * tf.pad SYMMETRIC repeats the edge sample, REFLECT does not
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# layout helpers
def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def same_pads(size, k, s):
    """TF SAME: pad_total = max((ceil(in/s)-1)*s + k - in, 0); before = total//2, rest after."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


# ---------------------------------------------------------------------------------------------
# convolution family
def conv2d(x, w, b=None, stride=1, padding='SAME'):
    """tf.nn.conv2d / keras Conv2D. x NHWC, w HWIO (kh,kw,Cin,Cout)."""
    kh, kw = w.shape[0], w.shape[1]
    xc = _nchw(x)
    if padding == 'SAME':
        pt, pb = same_pads(x.shape[1], kh, stride)
        pl, pr = same_pads(x.shape[2], kw, stride)
        xc = F.pad(xc, (pl, pr, pt, pb))
    y = F.conv2d(xc, w.permute(3, 2, 0, 1), b, stride=stride)
    return _nhwc(y)


def conv2d_transpose_2x2(x, w, b=None):
    """keras Conv2DTranspose(k=2, s=2, SAME). w is (kh,kw,Cout,Cin):
       out[n,2y+i,2x+j,co] = sum_ci in[n,y,x,ci] * w[i,j,co,ci] + b[co]   (models/pipelines.py:205)."""
    y = F.conv_transpose2d(_nchw(x), w.permute(3, 2, 0, 1), b, stride=2)
    return _nhwc(y)


def max_pool2(x):
    """MaxPool2D([2,2]) - SAME == VALID on even sizes (pipelines.py:197, forensics.py:70)."""
    return _nhwc(F.max_pool2d(_nchw(x), 2))


def avg_pool(x, f):
    return _nhwc(F.avg_pool2d(_nchw(x), f))


def depth_to_space(x, bs):
    """tf.nn.depth_to_space, DCR: out[n, y*bs+i, x*bs+j, c] = in[n, y, x, (i*bs+j)*C + c]."""
    n, h, w, c = x.shape
    co = c // (bs * bs)
    x = x.reshape(n, h, w, bs, bs, co).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(n, h * bs, w * bs, co)


def space_to_depth(x, bs):
    n, h, w, c = x.shape
    x = x.reshape(n, h // bs, bs, w // bs, bs, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(n, h // bs, w // bs, bs * bs * c)


def leaky_relu(x, alpha=0.2):
    """tf.keras.layers.LeakyReLU(alpha=0.2) (tf_helpers.py:23); tf.nn.leaky_relu default alpha is 0.2 too."""
    return torch.where(x > 0, x, alpha * x)


def activation(x, kind='leaky_relu'):
    """helpers/tf_helpers.py:22-28 `activation_mapping`: LeakyReLU(0.2) | relu | tanh | sigmoid | softsign (x / (1 + |x|))."""
    if kind == 'leaky_relu':
        return leaky_relu(x)
    if kind == 'relu':
        return torch.relu(x)
    if kind == 'tanh':
        return torch.tanh(x)
    if kind == 'sigmoid':
        return torch.sigmoid(x)
    if kind == 'softsign':
        return x / (1 + x.abs())
    raise ValueError('unknown activation {}'.format(kind))


def clip_ste(y):
    """stop_gradient(clip(y,0,1) - y) + y  (pipelines.py:223, compression.py:271)."""
    return (torch.clamp(y, 0, 1) - y).detach() + y


def pad2d(x, p, mode):
    """tf.pad on H,W by p with mode SYMMETRIC | REFLECT."""
    if p == 0:
        return x
    n, h, w, c = x.shape
    if mode == 'REFLECT':
        iy = list(range(p, 0, -1)) + list(range(h)) + list(range(h - 2, h - 2 - p, -1))
        ix = list(range(p, 0, -1)) + list(range(w)) + list(range(w - 2, w - 2 - p, -1))
    elif mode == 'SYMMETRIC':
        iy = list(range(p - 1, -1, -1)) + list(range(h)) + list(range(h - 1, h - 1 - p, -1))
        ix = list(range(p - 1, -1, -1)) + list(range(w)) + list(range(w - 1, w - 1 - p, -1))
    else:
        raise ValueError(mode)
    return x[:, iy][:, :, ix]


def resize_axis_table(in_size, out_size):
    """tf.image.resize(bilinear, half_pixel_centers, antialias=False) interpolation table for one axis:
       in = (o + 0.5) * in/out - 0.5; lo = max(floor(in),0); hi = min(ceil(in), in_size-1); t = in - floor(in)."""
    scale = in_size / out_size
    lo, hi, t = [], [], []
    for o in range(out_size):
        src = (o + 0.5) * scale - 0.5
        f = math.floor(src)
        lo.append(max(f, 0))
        hi.append(min(math.ceil(src), in_size - 1))
        t.append(src - f)
    return np.array(lo), np.array(hi), np.array(t)


def resize_bilinear(x, out_h, out_w):
    n, h, w, c = x.shape
    ylo, yhi, yt = resize_axis_table(h, out_h)
    xlo, xhi, xt = resize_axis_table(w, out_w)
    yt = torch.tensor(yt, dtype=x.dtype).view(1, -1, 1, 1)
    xt = torch.tensor(xt, dtype=x.dtype).view(1, 1, -1, 1)
    top = x[:, ylo]
    bot = x[:, yhi]
    tl, tr = top[:, :, xlo], top[:, :, xhi]
    bl, br = bot[:, :, xlo], bot[:, :, xhi]
    t = tl + (tr - tl) * xt
    b = bl + (br - bl) * xt
    return t + (b - t) * yt


def space_to_depth2(x, cp=None):
    """(N,H,W,C) -> (N,H/2,W/2,cp): block channel (2 pr + pc) C + c = pixel (2by + pr, 2bx + pc); channels >= 4 C are zero."""
    n, h, w, c = x.shape
    cp = 4 * c if cp is None else cp
    y = torch.zeros((n, h // 2, w // 2, cp), dtype=x.dtype)
    for pr in range(2):
        for pc in range(2):
            y[..., (2 * pr + pc) * c:(2 * pr + pc + 1) * c] = x[:, pr::2, pc::2, :]
    return y


def s2d_conv_weights(w5, cp=None):
    """The 3x3 stride-1 kernel over the space-to-depth image that equals a 5x5 stride-2 TF-SAME convolution over an even-sized
    image: tap ky = 2 dy + pr - 1 of block offset dy (0 outside 0..4).  Test-side statement of the identity the throughput mode
    runs the codec's strided layers through (neural-imaging_amd/csrc/latent.hip s2d_conv_weights_kernel)."""
    cin, cout = w5.shape[2], w5.shape[3]
    cp = 4 * cin if cp is None else cp
    w3 = torch.zeros((3, 3, cp, cout), dtype=w5.dtype)
    for dy in range(3):
        for dx in range(3):
            for pr in range(2):
                for pc in range(2):
                    ky, kx = 2 * dy + pr - 1, 2 * dx + pc - 1
                    if 0 <= ky <= 4 and 0 <= kx <= 4:
                        w3[dy, dx, (2 * pr + pc) * cin:(2 * pr + pc + 1) * cin, :] = w5[ky, kx]
    return w3


def resize_nearest(x, out_h, out_w):
    """tf.image.resize(method='nearest') = ResizeNearestNeighbor(half_pixel_centers=True, align_corners=False): source index
    min(floor((o + 0.5f) * (in / out)), in - 1), evaluated in float32 (tensorflow/core/kernels/image/resize_nearest_neighbor_op.cc
    with HalfPixelScalerForNN; not vendored in /root/reference - restated from the published kernel, parity unpinned here)."""
    n, h, w, c = x.shape

    def table(in_size, out_size):
        scale = np.float32(in_size) / np.float32(out_size)
        return np.array([min(int(math.floor((np.float32(o) + np.float32(0.5)) * scale)), in_size - 1) for o in range(out_size)])

    return x[:, table(h, out_h)][:, :, table(w, out_w)]


# tf.image.resize's remaining methods (tensorflow/python/ops/image_ops_impl.py resize_images_v2, antialias=False as
# helpers/tf_helpers.py:74-76 calls it): 'bicubic' -> ResizeBicubic(half_pixel_centers=True), 'area' -> ResizeArea,
# 'lanczos3' | 'lanczos5' | 'gaussian' | 'mitchellcubic' -> ScaleAndTranslate(kernel_type, antialias=False).  All are separable
# linear maps; each is restated as ONE (out, in) matrix per axis, built the way the kernel builds its per-output weights - float32
# where the kernel computes in float - and applied to both axes.  Not vendored in /root/reference: restated from the published
# kernels (tensorflow/core/kernels/image/resize_bicubic_op.cc, resize_area_op.cc, scale_and_translate_op.cc, sampling_kernels.h),
# PARITY UNPINNED like the rest of this file.
_f = np.float32


def _bound(v, limit):
    return min(limit - 1, max(0, v))


def resize_bicubic_axis(in_size, out_size):
    """ResizeBicubic, half_pixel_centers=True: Keys cubic a = -0.5 from a 1024-entry coefficient table indexed by
    lrintf(delta * 1024); taps in_loc - 1 .. in_loc + 2, a tap outside the image gets weight 0 and the rest are renormalised
    (GetWeightsAndIndices<HalfPixelScaler, true>)."""
    a, table = -0.5, 1 << 10
    co = np.zeros(2 * (table + 1), np.float32)
    for i in range(table + 1):
        x = _f(i * 1.0 / table)                                   # float x = i * 1.0 / kTableSize
        co[2 * i] = ((a + 2) * x - (a + 3)) * x * x + 1           # double arithmetic, stored as float
        x = _f(x + _f(1.0))
        co[2 * i + 1] = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    scale = _f(in_size) / _f(out_size)
    m = np.zeros((out_size, in_size), np.float64)
    for o in range(out_size):
        in_loc_f = _f(_f(_f(o) + _f(0.5)) * scale) - _f(0.5)
        in_loc = int(math.floor(in_loc_f))
        delta = _f(in_loc_f - _f(in_loc))
        offset = int(np.rint(_f(delta * _f(table))))
        taps = [(in_loc - 1, co[offset * 2 + 1]), (in_loc, co[offset * 2]),
                (in_loc + 1, co[(table - offset) * 2]), (in_loc + 2, co[(table - offset) * 2 + 1])]
        w = [_f(wt) if _bound(i, in_size) == i else _f(0.0) for i, wt in taps]
        tot = _f(_f(_f(w[0] + w[1]) + w[2]) + w[3])
        if abs(tot) >= 1000.0 * np.finfo(np.float32).tiny:
            inv = _f(1.0) / tot
            w = [_f(v * inv) for v in w]
        for (i, _), v in zip(taps, w):
            m[o, _bound(i, in_size)] += float(v)
    return m


def resize_area_axis(in_size, out_size):
    """ResizeArea: output o averages the input interval [o * scale, (o + 1) * scale) - boundary pixels by their covered
    fraction, pixels in between whole, indices clamped - and the sum is divided by scale (resize_area_op.cc, per axis)."""
    scale = _f(in_size) / _f(out_size)
    m = np.zeros((out_size, in_size), np.float64)

    def frac(v, lo, hi):
        if v < lo:
            return scale if v + 1 > hi else _f(v + 1 - lo)
        return _f(hi - v) if v + 1 > hi else _f(1.0)

    for o in range(out_size):
        lo, hi = _f(o * scale), _f((o + 1) * scale)
        for v in range(int(math.floor(lo)), int(math.ceil(hi))):
            m[o, _bound(v, in_size)] += float(frac(v, lo, hi)) / float(scale)
    return m


def _sampling_kernel(name):
    """(radius, f) of tensorflow/core/kernels/image/sampling_kernels.h; x >= 0, float32."""
    pi = _f(3.14159265359)
    if name in ('lanczos3', 'lanczos5'):
        r = _f(3.0 if name == 'lanczos3' else 5.0)

        def f(x):
            if x > r:
                return _f(0.0)
            if x <= _f(1e-3):
                return _f(1.0)
            return _f(r * _f(np.sin(_f(pi * x))) * _f(np.sin(_f(_f(pi * x) / r))) / _f(_f(pi * pi) * _f(x * x)))
        return r, f
    if name == 'gaussian':
        r, sigma = _f(1.5), _f(1.5) / _f(3.0)

        def f(x):
            return _f(0.0) if x >= r else _f(np.exp(-float(x) * float(x) / (2.0 * float(sigma) * float(sigma))))
        return r, f
    if name == 'mitchellcubic':
        def f(x):
            if x >= _f(2.0):
                return _f(0.0)
            if x >= _f(1.0):
                return _f(_f(_f(_f(_f(_f(-7.0) / _f(18.0)) * x + _f(2.0)) * x - _f(10.0) / _f(3.0)) * x) + _f(16.0) / _f(9.0))
            return _f(_f(_f(_f(_f(7.0) / _f(6.0)) * x - _f(2.0)) * x) * x + _f(8.0) / _f(9.0))
        return _f(2.0), f
    raise ValueError(name)


def resize_scale_translate_axis(in_size, out_size, kernel):
    """ScaleAndTranslate, antialias=False, translation 0 (ComputeSpansCore): the sample point of output o is (o + 0.5) / scale with
    scale = out / in; the span is the input pixels whose centres lie within the kernel radius (NOT widened when down-sampling),
    clamped to the image; weights kernel(|centre - sample|) normalised to sum 1."""
    radius, kern = _sampling_kernel(kernel)
    scale = _f(out_size) / _f(in_size)
    inv_scale = _f(1.0) / scale
    m = np.zeros((out_size, in_size), np.float64)
    for o in range(out_size):
        sample = _f(_f(_f(o) + _f(0.5)) * inv_scale)
        if sample < 0 or sample > in_size:
            continue
        first = int(math.ceil(_f(_f(sample - radius) - _f(0.5))))
        last = int(math.floor(_f(_f(sample + radius) - _f(0.5))))
        first, last = min(max(first, 0), in_size - 1), min(max(last, 0), in_size - 1)
        w = [kern(_f(abs(_f(_f(_f(src) + _f(0.5)) - sample)))) for src in range(first, last + 1)]
        tot = _f(0.0)
        for v in w:
            tot = _f(tot + v)
        if abs(tot) >= 1000.0 * np.finfo(np.float32).tiny:
            inv = _f(1.0) / tot
            for src, v in zip(range(first, last + 1), w):
                m[o, src] = float(_f(v * inv))
    return m


RESIZE_AXIS = {'bicubic': resize_bicubic_axis, 'area': resize_area_axis}
for _k in ('lanczos3', 'lanczos5', 'gaussian', 'mitchellcubic'):
    RESIZE_AXIS[_k] = (lambda i, o, _k=_k: resize_scale_translate_axis(i, o, _k))


def resize_separable(x, out_h, out_w, method):
    """tf.image.resize(x, (out_h, out_w), method) for the methods of RESIZE_AXIS: rows then columns with the axis matrices."""
    my = torch.tensor(RESIZE_AXIS[method](x.shape[1], out_h), dtype=x.dtype)
    mx = torch.tensor(RESIZE_AXIS[method](x.shape[2], out_w), dtype=x.dtype)
    return torch.einsum('oh,nhwc->nowc', my, torch.einsum('pw,nhwc->nhpc', mx, x))


# ---------------------------------------------------------------------------------------------
# colour spaces (tf.image.rgb_to_hsv / hsv_to_rgb; tensorflow/core/kernels/colorspace_op.h)
def rgb_to_hsv(x):
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    # max / min with an explicit selection order so that the (sub)gradient goes to ONE channel:
    # R first, then G, then B - the same order the H branch uses.
    v = torch.where((r >= g) & (r >= b), r, torch.where(g >= b, g, b))
    mn = torch.where((r <= g) & (r <= b), r, torch.where(g <= b, g, b))
    rng = v - mn
    zero = torch.zeros_like(v)
    safe_v = torch.where(v > 0, v, torch.ones_like(v))
    s = torch.where(v > 0, rng / safe_v, zero)
    safe_rng = torch.where(rng > 0, rng, torch.ones_like(rng))
    norm = 1.0 / (6.0 * safe_rng)
    h = torch.where(r == v, norm * (g - b),
                    torch.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
    h = torch.where(rng > 0, h, zero)
    h = torch.where(h < 0, h + 1, h)
    return torch.stack([h, s, v], dim=-1)


def hsv_to_rgb(x):
    h, s, v = x[..., 0], x[..., 1], x[..., 2]
    dh = h * 6
    dr = torch.clamp(torch.abs(dh - 3) - 1, 0, 1)
    dg = torch.clamp(2 - torch.abs(dh - 2), 0, 1)
    db = torch.clamp(2 - torch.abs(dh - 4), 0, 1)
    one_s = 1 - s
    return torch.stack([(one_s + s * dr) * v, (one_s + s * dg) * v, (one_s + s * db) * v], dim=-1)


# ---------------------------------------------------------------------------------------------
# quantisation (models/layers.py:118-172)
class _SoftRound(torch.autograd.Function):
    """'soft': forward round(x) [half-to-even], backward d/dx (x - sin(2 pi x)/(2 pi)) = 1 - cos(2 pi x)
       (layers.py:126-128)."""
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.round(x)      # torch.round is half-to-even like tf.round

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return g * (1 - torch.cos(2 * math.pi * x))


def quantization(x, mode='soft', taylor_terms=1):
    if mode == 'round':
        return torch.round(x).detach() + 0 * x
    if mode == 'sin':
        return x - torch.sin(2 * math.pi * x) / (2 * math.pi)
    if mode == 'soft':
        return _SoftRound.apply(x)
    if mode == 'harmonic':
        xa = x - torch.sin(2 * math.pi * x) / math.pi
        for k in range(2, taylor_terms):
            xa = xa + (-1.0) ** k * torch.sin(2 * math.pi * k * x) / (k * math.pi)
        return xa
    if mode == 'identity':
        return x
    raise ValueError('Unsupported quantization: {}'.format(mode))


def soft_quantization(x, alpha=255):
    """helpers/tf_helpers.py:271-277"""
    return _SoftRound.apply(alpha * x) / alpha


def _codebook_weights(values, codebook, v, gamma):
    """float64 kernel weights (layers.py:146-158 / tf_helpers.py:311-323)."""
    eps = 1e-72
    vals = values.reshape(-1, 1).to(torch.float64)
    cb = codebook.reshape(1, -1).to(torch.float64)
    dff = vals - cb
    if v <= 0:
        w = torch.exp(-gamma * dff ** 2)
    else:
        dff = gamma * dff
        w = (1 + dff ** 2 / v) ** (-(v + 1) / 2)
    return (w + eps) / (w + eps).sum(dim=1, keepdim=True)


def soft_codebook(x, codebook, v=50, gamma=25):
    """Quantization('soft-codebook') (layers.py:139-170): float64 inside, float32 (x.dtype) out."""
    w = _codebook_weights(x, codebook, v, gamma)
    cb = codebook.reshape(-1).to(torch.float64)
    soft = (w @ cb.reshape(-1, 1)).mean(dim=1).to(x.dtype).reshape(x.shape)
    hard = codebook.reshape(-1)[torch.argmax(w, dim=1)].to(x.dtype).reshape(x.shape)
    return (hard - soft).detach() + soft


def entropy(values, codebook, v=50, gamma=25):
    """Differentiable entropy (helpers/tf_helpers.py:290-333). Returns (entropy[float32-like], histogram)."""
    w = _codebook_weights(values, codebook, v, gamma)
    hist = w.mean(dim=0)
    hist = torch.clamp(hist, min=1e-9)
    hist = hist / hist.sum()
    ent = -(hist * torch.log(hist)).sum() / 0.6931
    return ent.to(values.dtype), hist


# ---------------------------------------------------------------------------------------------
# ConstrainedConv2D (models/layers.py:45-57)
def constrained_kernel(kernel, mask, strength=100.0):
    nf = kernel * (1 - mask)
    df = nf.sum(dim=(0, 1, 2)).reshape(1, 1, 1, -1)
    nf = strength * nf / df
    return nf - strength * mask


def constrained_conv(x, kernel, mask, strength=100.0):
    nf = constrained_kernel(kernel, mask, strength)
    xp = pad2d(x, 2, 'SYMMETRIC')
    return conv2d(xp, nf, None, 1, 'VALID')


# ---------------------------------------------------------------------------------------------
# classifier head + losses
def sparse_ce_from_probs(probs, labels):
    """tf.keras.losses.SparseCategoricalCrossentropy() on probabilities, eager path
       (keras/backend.py sparse_categorical_crossentropy, from_logits=False): clip to [1e-7, 1-1e-7],
       log, then sparse_softmax_cross_entropy_with_logits(log p) ; reduction = mean over the batch."""
    eps = 1e-7
    p = torch.clamp(probs, eps, 1 - eps)
    logp = torch.log(p)
    lse = torch.logsumexp(logp, dim=1)
    idx = torch.as_tensor(labels, dtype=torch.long)
    return (lse - logp[torch.arange(p.shape[0]), idx]).mean()


def mse255(a, b):
    """helpers/tf_helpers.py:31-32"""
    return ((255 * a - 255 * b) ** 2).mean()


def mae255(a, b):
    """helpers/tf_helpers.py:35-36"""
    return (255 * a - 255 * b).abs().mean()


def l2_loss(d):
    """tf.nn.l2_loss = sum(d^2)/2"""
    return (d ** 2).sum() / 2


# ---------------------------------------------------------------------------------------------
# Keras Adam (tf.keras.optimizers.Adam defaults: beta1 .9, beta2 .999, eps 1e-7, no amsgrad)
def adam_step(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """In-place on lists of tensors. t is the 1-based step count AFTER increment.
       theta -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)     (epsilon OUTSIDE the corrected sqrt)."""
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    for p, g, mi, vi in zip(params, grads, m, v):
        mi.mul_(beta1).add_(g, alpha=1 - beta1)
        vi.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        p.sub_(lr_t * mi / (vi.sqrt() + eps))


# --------------------------------------------------------------------------------------------------------------
# SSIM (Wang, Bovik, Sheikh, Simoncelli 2004) in the two flavours the reference calls
def ssim_skimage(a, b, data_range=1.0):
    """skimage.metrics.structural_similarity(a, b, multichannel=True, data_range=1) as helpers/metrics.py:9-25 calls it
    (scikit-image is not installed here; restated from its documentation): per channel, 7x7 uniform window
    (scipy.ndimage.uniform_filter), sample covariance N/(N-1), K1 = 0.01, K2 = 0.03, mean over the interior that the
    window covers completely (crop of (win-1)//2), then the mean over channels.  a, b: (H,W,C) numpy arrays."""
    from scipy.ndimage import uniform_filter
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    win, pad = 7, 3
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    vals = []
    for ch in range(a.shape[-1]):
        x, y = a[..., ch], b[..., ch]
        ux, uy = uniform_filter(x, win), uniform_filter(y, win)
        uxx, uyy, uxy = uniform_filter(x * x, win), uniform_filter(y * y, win), uniform_filter(x * y, win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def ssim_tf(a, b, max_val=1.0):
    """tf.image.ssim(a, b, max_val) as models/compression.py:89 calls it (TF 2.1 image_ops_impl: 11x11 Gaussian window,
    sigma 1.5, built as a softmax of -(x^2+y^2)/(2 sigma^2); VALID depthwise filtering; luminance x contrast-structure
    from population moments; mean over positions, then channels).  a, b: (N,H,W,C) torch tensors -> (N,)."""
    co = torch.arange(11, dtype=a.dtype) - 5.0
    g = -0.5 * (co[:, None] ** 2 + co[None, :] ** 2) / 1.5 ** 2
    g = torch.softmax(g.reshape(-1), 0).reshape(1, 1, 11, 11)
    c = a.shape[-1]
    k = g.repeat(c, 1, 1, 1)
    f = lambda t: F.conv2d(t.permute(0, 3, 1, 2), k, groups=c)
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    m0, m1 = f(a), f(b)
    num0, den0 = m0 * m1 * 2.0, m0 * m0 + m1 * m1
    lum = (num0 + c1) / (den0 + c1)
    num1, den1 = f(a * b) * 2.0, f(a * a + b * b)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(dim=(2, 3)).mean(dim=1)


def ssim_loss255(a, b):
    """helpers/tf_helpers.py:39-40: mean over the batch of 255 (1 - tf.image.ssim(a, b, 1.0))."""
    return torch.mean(255 * (1 - ssim_tf(a, b, 1.0)))


def _ssim_per_channel(a, b, max_val=1.0):
    """tf.image _ssim_per_channel: per (image, channel) means of luminance x contrast-structure and of contrast-structure."""
    co = torch.arange(11, dtype=a.dtype) - 5.0
    g = -0.5 * (co[:, None] ** 2 + co[None, :] ** 2) / 1.5 ** 2
    g = torch.softmax(g.reshape(-1), 0).reshape(1, 1, 11, 11)
    c = a.shape[-1]
    k = g.repeat(c, 1, 1, 1)
    f = lambda t: F.conv2d(t.permute(0, 3, 1, 2), k, groups=c)
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    m0, m1 = f(a), f(b)
    num0, den0 = m0 * m1 * 2.0, m0 * m0 + m1 * m1
    lum = (num0 + c1) / (den0 + c1)
    num1, den1 = f(a * b) * 2.0, f(a * a + b * b)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(dim=(2, 3)), cs.mean(dim=(2, 3))


MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def ssim_multiscale(a, b, max_val=1.0):
    """tf.image.ssim_multiscale (TF 2.1 image_ops_impl): five scales, each a 2x2 VALID average pooling of the previous one
    (even sizes assumed - TF pads odd ones symmetrically), relu(cs) of scales 0..3 and relu(ssim) of scale 4 raised to the
    power factors, product over the scales, mean over the channels.  -> (N,)"""
    mcs = []
    for k in range(len(MSSSIM_WEIGHTS)):
        if k > 0:
            a = F.avg_pool2d(a.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            b = F.avg_pool2d(b.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        ssim_pc, cs = _ssim_per_channel(a, b, max_val)
        mcs.append(torch.relu(cs))
    mcs.pop()
    stack = torch.stack(mcs + [torch.relu(ssim_pc)], dim=-1)
    w = torch.tensor(MSSSIM_WEIGHTS, dtype=a.dtype)
    return torch.prod(stack ** w, dim=-1).mean(dim=-1)


def msssim_loss255(a, b):
    """helpers/tf_helpers.py:43-44."""
    return torch.mean(255 * (1 - ssim_multiscale(a, b, 1.0)))


IMAGE_LOSSES = {'L2': mse255, 'L1': mae255, 'SSIM': ssim_loss255, 'MS-SSIM': msssim_loss255}

