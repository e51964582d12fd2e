"""
UNet / FAN / TwitterDCN graphs restated on the oracle ops.  TEST INFRASTRUCTURE.

Parameters are ordered dicts {name: tensor} in the Keras layouts the reference's checkpoints use:
Conv2D kernel (kh,kw,Cin,Cout), Conv2DTranspose kernel (kh,kw,Cout,Cin), Dense kernel (in,out).

  unet_*   models/pipelines.py:175-226
  fan_*    models/forensics.py:29-94  (+ ConstrainedConv2D models/layers.py:36-57)
  dcn_*    models/compression.py:49-138, 197-279 (+ DiscreteLatent models/layers.py:183-203)
"""
from collections import OrderedDict

import numpy as np
import torch

from . import tables
from . import tfops as T


# ---------------------------------------------------------------------------------------------
# initialisation (Keras defaults: glorot_uniform kernels, zero biases) with an explicit generator
def _glorot(gen, shape, fan_in, fan_out, dtype):
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * limit).to(dtype)


def _conv_param(p, name, gen, kh, kw, cin, cout, dtype, transpose=False):
    rf = kh * kw
    shape = (kh, kw, cout, cin) if transpose else (kh, kw, cin, cout)
    p[name + '/kernel'] = _glorot(gen, shape, cin * rf, cout * rf, dtype)
    p[name + '/bias'] = torch.zeros(cout, dtype=dtype)


def unet_init(seed=1234, in_channels=4, n_steps=5, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    cin = in_channels
    for n in range(1, n_steps + 1):
        c = 32 * 2 ** (n - 1)
        _conv_param(p, 'ec{}1'.format(n), gen, 3, 3, cin, c, dtype)
        _conv_param(p, 'ec{}2'.format(n), gen, 3, 3, c, c, dtype)
        cin = c
    for n in range(1, n_steps):
        c = 32 * 2 ** (n_steps - n - 1)
        _conv_param(p, 'dct{}'.format(n), gen, 2, 2, cin, c, dtype, transpose=True)
        _conv_param(p, 'dc{}1'.format(n), gen, 3, 3, 2 * c, c, dtype)
        _conv_param(p, 'dc{}2'.format(n), gen, 3, 3, c, c, dtype)
        cin = c
    _conv_param(p, 'dc{}'.format(n_steps), gen, 3, 3, cin, 12, dtype)
    return p


def unet_forward(p, x, n_steps=5, return_tensors=False, activation='leaky_relu'):
    """models/pipelines.py:175-226; `activation` is the constructor's (helpers/tf_helpers.py:22-28), applied behind every 3x3 layer
    but the last."""
    act = lambda v: T.activation(v, activation)
    t = OrderedDict()
    t['ep0'] = x
    for n in range(1, n_steps + 1):
        a = act(T.conv2d(t['ep{}'.format(n - 1)], p['ec{}1/kernel'.format(n)], p['ec{}1/bias'.format(n)]))
        t['ec{}1'.format(n)] = a
        a = act(T.conv2d(a, p['ec{}2/kernel'.format(n)], p['ec{}2/bias'.format(n)]))
        t['ec{}2'.format(n)] = a
        if n < n_steps:
            t['ep{}'.format(n)] = T.max_pool2(a)
    t['dc02'] = t['ec{}2'.format(n_steps)]
    for n in range(1, n_steps):
        up = T.conv2d_transpose_2x2(t['dc{}2'.format(n - 1)], p['dct{}/kernel'.format(n)], p['dct{}/bias'.format(n)])
        t['dct{}'.format(n)] = up
        cat = torch.cat([up, t['ec{}2'.format(n_steps - n)]], dim=-1)        # [upsampled, skip]  pipelines.py:211
        a = act(T.conv2d(cat, p['dc{}1/kernel'.format(n)], p['dc{}1/bias'.format(n)]))
        t['dc{}1'.format(n)] = a
        a = act(T.conv2d(a, p['dc{}2/kernel'.format(n)], p['dc{}2/bias'.format(n)]))
        t['dc{}2'.format(n)] = a
    z = T.conv2d(t['dc{}2'.format(n_steps - 1)], p['dc{}/kernel'.format(n_steps)], p['dc{}/bias'.format(n_steps)])
    t['dc{}'.format(n_steps)] = z
    y = T.clip_ste(T.depth_to_space(z, 2))
    t['y'] = y
    return (y, t) if return_tensors else y


# ---------------------------------------------------------------------------------------------
def fan_init(n_classes, seed=4321, n_filters=32, n_fscale=2, n_convolutions=4, kernel=5, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    p['constrained/kernel'] = torch.tensor(tables.fan_residual_init(), dtype=dtype)
    cin, nf = 3, n_filters
    for i in range(n_convolutions):
        _conv_param(p, 'conv{}'.format(i + 1), gen, kernel, kernel, cin, nf, dtype)
        cin, nf = nf, int(nf * n_fscale)
    nf = int(nf // n_fscale)
    _conv_param(p, 'conv1x1', gen, 1, 1, cin, nf, dtype)
    p['dense/kernel'] = _glorot(gen, (nf, n_classes), nf, n_classes, dtype)
    p['dense/bias'] = torch.zeros(n_classes, dtype=dtype)
    return p


def _dense_names(p):
    """Keras names of the Dense layers in creation order: dense, dense_1, ... (the last one is the classifier)."""
    names = [k[:-len('/kernel')] for k in p if k.startswith('dense') and k.endswith('/kernel')]
    return sorted(names, key=lambda n: int(n.split('_')[1]) if '_' in n else 0)


def fan_forward(p, x, n_convolutions=4, return_tensors=False, use_gap=True, dropout=0.0, dropout_masks=None,
                activation='leaky_relu'):
    """models/forensics.py:62-90: constrained conv -> n x [conv + LReLU -> pool] -> 1x1 conv + LReLU -> GAP | Flatten ->
    hidden Dense + LReLU layers (n_dense) -> Dense softmax."""
    act = lambda v: T.activation(v, activation)
    t = OrderedDict()
    mask = torch.tensor(tables.center_mask_2dfilter(5, 3), dtype=x.dtype)
    net = T.constrained_conv(x, p['constrained/kernel'], mask)
    t['constrained'] = net
    for i in range(n_convolutions):
        net = act(T.conv2d(net, p['conv{}/kernel'.format(i + 1)], p['conv{}/bias'.format(i + 1)]))
        t['conv{}'.format(i + 1)] = net
        net = T.max_pool2(net)
        t['pool{}'.format(i + 1)] = net
    net = act(T.conv2d(net, p['conv1x1/kernel'], p['conv1x1/bias']))
    t['conv1x1'] = net
    feat = net.mean(dim=(1, 2)) if use_gap else net.reshape(net.shape[0], -1)       # Flatten is NHWC row-major
    t['gap'] = feat
    dn = _dense_names(p)
    for li, name in enumerate(dn[:-1]):
        feat = act(feat @ p[name + '/kernel'] + p[name + '/bias'])
        t[name] = feat
        if dropout_masks is not None:                    # Keras Dropout at training time (forensics.py:88), given mask
            feat = feat * dropout_masks[li].to(feat.dtype) / (1.0 - dropout)
    logits = feat @ p[dn[-1] + '/kernel'] + p[dn[-1] + '/bias']
    t['logits'] = logits
    probs = torch.softmax(logits, dim=1)
    return (probs, t) if return_tensors else probs


# ---------------------------------------------------------------------------------------------
def dcn_init(seed=777, n_features=32, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    _conv_param(p, 'e1', gen, 5, 5, 3, 64, dtype)
    _conv_param(p, 'e2', gen, 5, 5, 64, 128, dtype)
    for b in range(1, 4):
        _conv_param(p, 'er{}a'.format(b), gen, 3, 3, 128, 128, dtype)
        _conv_param(p, 'er{}b'.format(b), gen, 3, 3, 128, 128, dtype)
    _conv_param(p, 'elat', gen, 5, 5, 128, n_features, dtype)
    p['latent_scaling'] = torch.ones((), dtype=dtype)
    _conv_param(p, 'd512', gen, 3, 3, n_features, 512, dtype)
    for b in range(1, 4):
        _conv_param(p, 'dr{}a'.format(b), gen, 3, 3, 128, 128, dtype)
        _conv_param(p, 'dr{}b'.format(b), gen, 3, 3, 128, 128, dtype)
    _conv_param(p, 'd256', gen, 3, 3, 128, 256, dtype)
    _conv_param(p, 'd12', gen, 3, 3, 64, 12, dtype)
    return p


def dcn_encode(p, x, latent_bpf=5, rounding='soft-codebook', v=50, gamma=25, activation='leaky_relu'):
    """compression.py:219-241. Returns (latent, entropy, pre-quantisation latent).  `activation` replaces every layer activation
    EXCEPT the tf.nn.leaky_relu in front of the first residual block (:224), which the reference hard-codes."""
    act = lambda t_: T.activation(t_, activation)
    net = 2 * (x - 0.5)
    net = act(T.conv2d(net, p['e1/kernel'], p['e1/bias'], stride=2))
    net = T.conv2d(net, p['e2/kernel'], p['e2/bias'], stride=2)
    # block 1: fed LReLU(net), skip adds the PRE-activation net (compression.py:224-227)
    r = act(T.conv2d(T.leaky_relu(net), p['er1a/kernel'], p['er1a/bias']))
    r = T.conv2d(r, p['er1b/kernel'], p['er1b/bias'])
    net = net + r
    for b in (2, 3):
        r = act(T.conv2d(net, p['er{}a/kernel'.format(b)], p['er{}a/bias'.format(b)]))
        r = T.conv2d(r, p['er{}b/kernel'.format(b)], p['er{}b/bias'.format(b)])
        net = net + r
    z = T.conv2d(net, p['elat/kernel'], p['elat/bias'], stride=2)
    cb = torch.tensor(tables.codebook(latent_bpf), dtype=x.dtype)
    zs = z * p['latent_scaling']                                   # layers.py:195-198 (scale always trainable)
    if rounding == 'soft-codebook':
        lat = T.soft_codebook(zs, cb, v, gamma)
    else:
        lat = T.quantization(zs, rounding)
    ent, _ = T.entropy(lat, cb, v, gamma)                          # layers.py:201
    return lat, ent, zs


def dcn_decode(p, lat, activation='leaky_relu'):
    """compression.py:245-271"""
    act = lambda t_: T.activation(t_, activation)
    net = T.depth_to_space(T.conv2d(lat, p['d512/kernel'], p['d512/bias']), 2)
    for b in (1, 2, 3):
        r = act(T.conv2d(net, p['dr{}a/kernel'.format(b)], p['dr{}a/bias'.format(b)]))
        r = T.conv2d(r, p['dr{}b/kernel'.format(b)], p['dr{}b/bias'.format(b)])
        net = net + r
    net = T.depth_to_space(act(T.conv2d(net, p['d256/kernel'], p['d256/bias'])), 2)
    net = T.depth_to_space(T.conv2d(net, p['d12/kernel'], p['d12/bias']), 2)
    return T.clip_ste((net + 1) / 2)


def dcn_forward(p, x, **kw):
    lat, ent, _ = dcn_encode(p, x, **kw)
    return dcn_decode(p, lat, activation=kw.get('activation', 'leaky_relu')), ent, lat


def dcn_loss(x, y, ent, entropy_weight=250.0):
    """compression.py:92-93"""
    return T.l2_loss(x - y) + entropy_weight * ent


class DCNTrainer(object):
    """DCN.training_step restated (models/compression.py:123-138): tape over l2_loss(x - y) + entropy_weight * H, Keras Adam
    (default learning rate 1e-3 unless assigned), returns {'loss': sqrt(2 loss), 'ssim', 'entropy'}."""

    def __init__(self, params, entropy_weight=250.0):
        self.p, self.entropy_weight = params, entropy_weight
        self._m = self._v = None
        self._t, self.lr = 0, 1e-3

    def training_step(self, x, learning_rate=None):
        ps = list(self.p.values())
        for q in ps:
            q.requires_grad_(True)
        y, ent, _ = dcn_forward(self.p, x)
        loss = dcn_loss(x, y, ent, self.entropy_weight)
        grads = torch.autograd.grad(loss, ps)
        for q in ps:
            q.requires_grad_(False)
        if learning_rate is not None:
            self.lr = learning_rate
        if self._m is None:
            self._m = [torch.zeros_like(q) for q in ps]
            self._v = [torch.zeros_like(q) for q in ps]
        self._t += 1
        with torch.no_grad():
            T.adam_step(ps, list(grads), self._m, self._v, self._t, self.lr)
            ssim = float(T.ssim_tf(x, y.detach()).mean()) if min(x.shape[1], x.shape[2]) >= 11 else float('nan')
        return {'loss': float(np.sqrt(2 * float(loss.detach()))), 'ssim': ssim, 'entropy': float(ent.detach())}


def count_params(p):
    return int(sum(int(np.prod(v.shape)) for v in p.values()))


# ---------------------------------------------------------------------------------------------
def inet_forward(p, x):
    """INet (models/pipelines.py:233-295): 1x1 up-sampling -> depth_to_space(2) -> REFLECT pad + VALID k x k demosaicing ->
    1x1 colour conversion -> 1x1 (3->12) + tanh -> 1x1 (12->3) -> straight-through clip.  p: names up, demosaic, srgb,
    gamma1, gamma2 (+ '/kernel', '/bias')."""
    k = p['demosaic/kernel'].shape[0]
    pad = (k - 1) // 2
    h12 = T.conv2d(x, p['up/kernel'], None, 1, 'VALID')
    bayer = T.depth_to_space(h12, 2)
    bp = torch.nn.functional.pad(bayer.permute(0, 3, 1, 2), (pad, pad, pad, pad), mode='reflect').permute(0, 2, 3, 1)
    rgb = T.conv2d(bp, p['demosaic/kernel'], None, 1, 'VALID')
    srgb = T.conv2d(rgb, p['srgb/kernel'], None, 1, 'VALID')
    g0 = torch.tanh(T.conv2d(srgb, p['gamma1/kernel'], p['gamma1/bias'], 1, 'VALID'))
    y = T.conv2d(g0, p['gamma2/kernel'], p['gamma2/bias'], 1, 'VALID')
    return y + (torch.clamp(y, 0, 1) - y).detach()


def classic_isp_forward(p, x, residual=True):
    """ClassicISP (models/pipelines.py:416-453 `_ClassicISP.call`; demosaicing = models/layers.py:238-258).  p: up, srgb,
    [bilinear], demosaicing/alpha, demosaicing/conv{i}, demosaicing/out ('/kernel', '/bias'); the CNN is present iff
    'demosaicing/out/kernel' is."""
    ste = lambda t, lo, hi: t + (torch.clamp(t, lo, hi) - t).detach()
    bayer = T.depth_to_space(T.conv2d(x, p['up/kernel'], None, 1, 'VALID'), 2)
    f = None
    if 'demosaicing/out/kernel' in p:
        f = bayer
        n = len([k for k in p if k.startswith('demosaicing/conv') and k.endswith('/kernel')])
        for i in range(n):
            f = T.leaky_relu(T.conv2d(f, p['demosaicing/conv{}/kernel'.format(i)], p['demosaicing/conv{}/bias'.format(i)],
                                      1, 'SAME'))
        f = T.conv2d(f, p['demosaicing/out/kernel'], p['demosaicing/out/bias'], 1, 'VALID')
        f = torch.tanh(f) if residual else torch.sigmoid(f)
    if residual:
        k = p['bilinear/kernel'].shape[0]
        pad = (k - 1) // 2
        bp = torch.nn.functional.pad(bayer.permute(0, 3, 1, 2), (pad,) * 4, mode='reflect').permute(0, 2, 3, 1)
        rgb = T.conv2d(bp, p['bilinear/kernel'], None, 1, 'VALID')
        if f is not None:
            rgb = rgb - p['demosaicing/alpha'] * f
    else:
        rgb = f
    rgb = ste(rgb, 0.0, 1.0)
    srgb = T.conv2d(rgb, p['srgb/kernel'], None, 1, 'VALID')
    return torch.pow(ste(srgb, 1.0 / 255, 1.0), 1 / 2.2)


def dnet_forward(p, x):
    """DNet (models/pipelines.py:298-349).  p: conv0..conv{n-1}, up, proj, out ('/kernel', '/bias')."""
    refl = lambda t, pad: torch.nn.functional.pad(t.permute(0, 3, 1, 2), (pad,) * 4, mode='reflect').permute(0, 2, 3, 1)
    n_layers = len([k for k in p if k.startswith('conv') and k.endswith('/kernel')])
    pad = (p['conv0/kernel'].shape[0] - 1) // 2
    deep = x
    for r in range(n_layers):
        deep = refl(torch.relu(T.conv2d(deep, p['conv{}/kernel'.format(r)], p['conv{}/bias'.format(r)], 1, 'VALID')), pad)
    bayer = T.depth_to_space(T.conv2d(x, p['up/kernel'], None, 1, 'VALID'), 2)
    feat = T.depth_to_space(deep, 2)
    pu = torch.relu(T.conv2d(torch.cat((feat, bayer), dim=3), p['proj/kernel'], p['proj/bias'], 1, 'VALID'))
    y = T.conv2d(refl(pu, pad), p['out/kernel'], None, 1, 'VALID')
    return y + (torch.clamp(y, 0, 1) - y).detach()
