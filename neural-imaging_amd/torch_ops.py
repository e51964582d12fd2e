"""
PyTorch-ROCm custom-op registration of the C-ABI kernels (torch.library, namespace `nimg`).

The model classes of this package drive libnimg.so through explicit forward / backward calls (ops.py); this module exposes the
same entry points to PyTorch itself, so that a caller who composes the channel with torch modules gets dispatcher-visible ops
with autograd formulas:

    import neural_imaging_amd.torch_ops                     # registers the ops
    y = torch.ops.nimg.djpeg(x, qtab, 'soft')               # models/jpeg.py:91-159, DIFFERENTIABLE (nimg_djpeg_fwd / _bwd)
    z = torch.ops.nimg.conv2d(x, w, b, 1, 'leaky_relu')      # Conv2D SAME (+bias, LeakyReLU 0.2), DIFFERENTIABLE at stride 1:
                                                            #   nimg_conv2d_* forward / input gradient / weight gradient
    c = torch.ops.nimg.cconv3(x, nf, 1)                     # ConstrainedConv2D core (models/layers.py:56-57), forward only
    d = torch.ops.nimg.cconv3_dgrad(dy, nf)                 #   its input gradient as an op of its own (nimg_cconv3 + _dgrad_border)
    p, k = torch.ops.nimg.conv_lrelu_pool(x, w, b)           # FAN feature stage (models/forensics.py:69-70), forward only

Round 5: the rest of the channel's differentiable stages are registered the same way, each with an Autograd kernel over the
library's own forward / backward entry points -
    manipulation_sharpen / _gaussian / _resample / _gamma / _median / _awgn   (helpers/tf_helpers.py:68-184)
    max_pool2, conv_transpose2x2, depth_to_space_clip                         (models/pipelines.py:190-223)
    constrained_conv (gradients to the image AND the 5x5x3x3 filter)           (models/layers.py:12-57)
    mse255 (the NIP loss), fan_head (GAP -> Dense -> softmax -> CE on probabilities: loss + probabilities)  (forensics.py:80-94)
so that a torch module graph can be assembled from `torch.ops.nimg.*` alone and trained with torch.autograd
(tests/test_gpu_ops.py::test_torch_ops_autograd_matches_the_explicit_backward).

Autograd: `djpeg` and `conv2d` carry Autograd kernels (autograd.Function over the raw ops djpeg_fwd / djpeg_bwd and conv2d_fwd /
conv2d_dgrad / conv2d_wgrad).  The forward-only ops REFUSE inputs that require grad (NotImplementedError) instead of silently
cutting the graph; the package's own models run these stages through explicit backward calls (models/layers.py,
models/forensics.py), not through the dispatcher.

Tensors are contiguous float32 NHWC on the GPU; weights are Keras HWIO.  No CPU implementation is registered: calling an op
with CPU tensors raises (there is no fallback anywhere in this package).
"""
import torch

from . import ops

_LIB = 'nimg'


def _define():
    lib = torch.library.Library(_LIB, 'DEF')
    lib.define('djpeg(Tensor x, Tensor qtab, str rounding) -> Tensor')
    lib.define('djpeg_fwd(Tensor x, Tensor qtab, str rounding) -> (Tensor, Tensor)')
    lib.define('djpeg_bwd(Tensor x, Tensor gy, Tensor mask, Tensor qtab, str rounding) -> Tensor')
    lib.define('conv2d(Tensor x, Tensor w, Tensor? bias, int stride, str act) -> Tensor')
    lib.define('conv2d_fwd(Tensor x, Tensor w, Tensor? bias, int stride, str act) -> Tensor')
    lib.define('conv2d_dgrad(Tensor dz, Tensor w, int h, int w_) -> Tensor')
    lib.define('conv2d_wgrad(Tensor x, Tensor dz, int ks, int stride) -> (Tensor, Tensor)')
    lib.define('cconv3(Tensor x, Tensor w, int pad_mode) -> Tensor')
    lib.define('cconv3_dgrad(Tensor dy, Tensor w) -> Tensor')
    lib.define('conv_lrelu_pool(Tensor x, Tensor w, Tensor bias) -> (Tensor, Tensor)')
    return lib


def _cuda_only(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('nimg ops run on the GPU only (libnimg.so has no CPU path)')


def _djpeg_fwd(x, qtab, rounding):
    _cuda_only(x, qtab)
    y, mask, _, _ = ops.djpeg_fwd(x.contiguous(), qtab.contiguous(), rounding, want_mask=True)
    return y, mask


def _djpeg_bwd(x, gy, mask, qtab, rounding):
    _cuda_only(x, gy)
    return ops.djpeg_bwd(x.contiguous(), gy.contiguous(), mask, qtab.contiguous(), rounding)


class _DJpeg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, qtab, rounding):
        y, mask = torch.ops.nimg.djpeg_fwd(x, qtab, rounding)
        ctx.save_for_backward(x, mask, qtab)
        ctx.rounding = rounding
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mask, qtab = ctx.saved_tensors
        return torch.ops.nimg.djpeg_bwd(x, gy, mask, qtab, ctx.rounding), None, None


def _conv2d(x, w, bias, stride, act):
    _cuda_only(x, w, bias)
    return ops.conv2d(x.contiguous(), w.contiguous(), None if bias is None else bias.contiguous(), stride=stride,
                      act=act if act else None)


def _conv2d_dgrad(dz, w, h, w_):
    _cuda_only(dz, w)
    return ops.conv2d_dgrad(dz.contiguous(), w.contiguous(), (h, w_))


def _conv2d_wgrad(x, dz, ks, stride):
    _cuda_only(x, dz)
    db = torch.empty((dz.shape[-1],), dtype=torch.float32, device=dz.device)
    dw = ops.conv2d_wgrad(x.contiguous(), dz.contiguous(), ks, stride=stride, db=db)
    return dw, db


class _Conv2D(torch.autograd.Function):
    """Conv2D(SAME) + bias + optional LeakyReLU(0.2); backward through the library's input- and weight-gradient kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, act):
        y = torch.ops.nimg.conv2d_fwd(x, w, bias, stride, act)
        ctx.save_for_backward(x, w, y)
        ctx.stride, ctx.act, ctx.has_bias = stride, act, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        dz = ops.lrelu_bwd(gy.contiguous(), y) if ctx.act == 'leaky_relu' else gy.contiguous()
        if ctx.stride != 1:
            raise NotImplementedError('autograd of the strided convolution: use models/compression.py (zero-insertion dgrad)')
        dx = torch.ops.nimg.conv2d_dgrad(dz, w, x.shape[1], x.shape[2]) if ctx.needs_input_grad[0] else None
        dw, db = torch.ops.nimg.conv2d_wgrad(x, dz, w.shape[0], ctx.stride)
        return dx, dw, (db if ctx.has_bias else None), None, None


def _cconv3(x, w, pad_mode):
    _cuda_only(x, w)
    return ops.cconv3(x.contiguous(), w.contiguous(), pad_mode=pad_mode)[0]


def _cconv3_dgrad(dy, w):
    _cuda_only(dy, w)
    return ops.cconv3_dgrad(dy.contiguous(), w.contiguous())


def _conv_lrelu_pool(x, w, bias):
    _cuda_only(x, w, bias)
    return ops.conv2d_pool(x.contiguous(), w.contiguous(), bias.contiguous())


def _register():
    lib = _define()
    impl = torch.library.Library(_LIB, 'IMPL', 'CUDA')
    impl.impl('djpeg_fwd', _djpeg_fwd)
    impl.impl('djpeg_bwd', _djpeg_bwd)
    impl.impl('conv2d', _conv2d)
    impl.impl('conv2d_fwd', _conv2d)
    impl.impl('conv2d_dgrad', _conv2d_dgrad)
    impl.impl('conv2d_wgrad', _conv2d_wgrad)
    impl.impl('cconv3', _cconv3)
    impl.impl('cconv3_dgrad', _cconv3_dgrad)
    impl.impl('conv_lrelu_pool', _conv_lrelu_pool)
    # differentiable front doors: autograd formulas over the raw kernels
    auto = torch.library.Library(_LIB, 'IMPL', 'Autograd')
    auto.impl('djpeg', lambda x, qtab, rounding: _DJpeg.apply(x, qtab, rounding))
    impl.impl('djpeg', lambda x, qtab, rounding: _djpeg_fwd(x, qtab, rounding)[0])
    auto.impl('conv2d', lambda x, w, bias, stride, act: _Conv2D.apply(x, w, bias, stride, act))

    def forward_only(name):
        def kernel(*args):
            if torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad for a in args):
                raise NotImplementedError('torch.ops.nimg.{} is forward-only: no autograd formula is registered (the models of '
                                          'this package run its backward pass explicitly)'.format(name))
            with torch._C._AutoDispatchBelowAutograd():
                return getattr(torch.ops.nimg, name)(*args)
        return kernel
    for name in ('cconv3', 'cconv3_dgrad', 'conv_lrelu_pool'):
        auto.impl(name, forward_only(name))
    return lib, impl, auto


# ----------------------------------------------------------------------------------------------------------------------
# round 5: manipulations, pooling, transposed convolution, constrained filter, losses, classifier head
def _manip_objects():
    from .helpers import tf_helpers as th
    return {'sharpen': th._sharpen, 'gaussian': th._gaussian, 'resample': th._resample, 'gamma': th.Gamma(), 'median': th.Median()}


_MANIP = {}


class _Manip(torch.autograd.Function):
    """One photo manipulation (helpers/tf_helpers.py manipulation_*): forward(x, strength) / backward(ctx, dy) of the object the
    workflow itself uses."""

    @staticmethod
    def forward(ctx, kind, x, strength):
        if not _MANIP:
            _MANIP.update(_manip_objects())
        _cuda_only(x)
        y, mctx = _MANIP[kind].forward(x.contiguous(), strength, training=True)
        ctx.kind, ctx.mctx = kind, mctx
        return y

    @staticmethod
    def backward(ctx, gy):
        return None, _MANIP[ctx.kind].backward(ctx.mctx, gy.contiguous()), None


class _Awgn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, noise, strength):
        _cuda_only(x, noise)
        y, mask = ops.awgn_fwd(x.contiguous(), noise.contiguous(), float(strength), want_mask=True)
        ctx.save_for_backward(x, noise, mask)
        ctx.s = float(strength)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, noise, mask = ctx.saved_tensors
        if ctx.needs_input_grad[1]:        # forward-only input (module docstring): refuse instead of cutting the graph silently
            raise RuntimeError('nimg::awgn: the noise tensor is a forward-only input (no gradient is defined for it)')
        return ops.awgn_bwd(x, noise, gy.contiguous(), mask, ctx.s), None, None


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _cuda_only(x)
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2_bwd(gy.contiguous(), x, None, apply_mask=False)      # first maximum of each window, like tf


class _ConvT2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        _cuda_only(x, w, bias)
        ctx.save_for_backward(x.contiguous(), w.contiguous())
        return ops.convt2x2(x.contiguous(), w.contiguous(), bias.contiguous())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        dx = ops.convt2x2_dgrad(gy, w) if ctx.needs_input_grad[0] else None
        dw = ops.convt2x2_wgrad(x, gy) if ctx.needs_input_grad[1] else None
        db = ops.bias_grad(gy) if ctx.needs_input_grad[2] else None
        return dx, dw, db


class _D2SClip(torch.autograd.Function):
    """tf.nn.depth_to_space(x, 2) followed by the straight-through clip to [0, 1] (pipelines.py:218-223)."""

    @staticmethod
    def forward(ctx, x):
        _cuda_only(x)
        return ops.d2s_clip(x.contiguous(), 1.0, 0.0, True)

    @staticmethod
    def backward(ctx, gy):
        return ops.d2s_clip_bwd(gy.contiguous(), 1.0)


class _ConstrainedConv(torch.autograd.Function):
    """ConstrainedConv2D (models/layers.py:12-57): kernel re-normalisation + SYMMETRIC-padded 5x5 filter, gradients to both."""

    @staticmethod
    def forward(ctx, x, kernel, strength):
        _cuda_only(x, kernel)
        x, kernel = x.contiguous(), kernel.contiguous()
        nf = ops.constrained_kernel(kernel, float(strength))
        ctx.save_for_backward(x, kernel, nf)
        ctx.strength = float(strength)
        return ops.cconv3(x, nf, pad_mode=1)[0]

    @staticmethod
    def backward(ctx, gy):
        x, kernel, nf = ctx.saved_tensors
        gy = gy.contiguous()
        n, h, w, _ = gy.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.cconv3_dgrad(gy, nf) if min(h, w) >= 4 else ops.fold_pad(
                ops.conv2d(gy, ops.flip_weights(nf), None, pads=(4, 4), out_hw=(h + 4, w + 4)), 2, 1)
        dk = torch.empty_like(kernel)
        ops.constrained_kernel_bwd(kernel, ops.conv2d_wgrad(x, gy, 5, pads=(2, 2), pad_mode=1), dk, ctx.strength)
        return dx, dk, None


class _Mse255(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _cuda_only(a, b)
        loss, g = ops.mse255(a.contiguous(), b.contiguous(), grad_scale=1.0)
        ctx.save_for_backward(g)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        ga = g * gl                        # d loss / d a; the loss is symmetric in a - b, so d loss / d b = -ga
        return (ga if ctx.needs_input_grad[0] else None), (-ga if ctx.needs_input_grad[1] else None)


class _FanHead(torch.autograd.Function):
    """GlobalAveragePooling -> Dense -> softmax -> SparseCategoricalCrossentropy on the probabilities (forensics.py:80-94):
    returns (mean loss, probabilities); gradients to the feature map, the Dense kernel and bias (through the loss only)."""

    @staticmethod
    def forward(ctx, act, w, b, labels):
        _cuda_only(act, w, b, labels)
        act, w, b = act.contiguous(), w.contiguous(), b.contiguous()
        n = act.shape[0]
        gap, probs, loss_per, dlogits = ops.fan_head_fwd(act, w, b, labels.to(torch.int32).contiguous(), 1.0 / n)
        ctx.mark_non_differentiable(probs)
        if not any(ctx.needs_input_grad[:3]):          # inference through the CUDA key: the loss only, no gradient kernels
            return loss_per.mean().reshape(()), probs      # (glue outside the hot path, like the g * gl products below)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        dact, loss = ops.fan_head_bwd(act, gap, w, dlogits, loss_per, 1.0 / n, dw, db, alpha=1.0)    # slope 1: no LeakyReLU' folded in
        ctx.save_for_backward(dact, dw, db)
        return loss.reshape(()), probs

    @staticmethod
    def backward(ctx, gl, _gp):
        dact, dw, db = ctx.saved_tensors
        return dact * gl, dw * gl, db * gl, None


def _register_round5():
    lib = torch.library.Library(_LIB, 'FRAGMENT')
    lib.define('manipulation_sharpen(Tensor x, float strength) -> Tensor')
    lib.define('manipulation_gaussian(Tensor x, float std) -> Tensor')
    lib.define('manipulation_resample(Tensor x, float factor) -> Tensor')
    lib.define('manipulation_gamma(Tensor x, float strength) -> Tensor')
    lib.define('manipulation_median(Tensor x, int kernel) -> Tensor')
    lib.define('manipulation_awgn(Tensor x, Tensor noise, float strength) -> Tensor')
    lib.define('max_pool2(Tensor x) -> Tensor')
    lib.define('conv_transpose2x2(Tensor x, Tensor w, Tensor bias) -> Tensor')
    lib.define('depth_to_space_clip(Tensor x) -> Tensor')
    lib.define('constrained_conv(Tensor x, Tensor kernel, float strength) -> Tensor')
    lib.define('mse255(Tensor a, Tensor b) -> Tensor')
    lib.define('fan_head(Tensor act, Tensor w, Tensor b, Tensor labels) -> (Tensor, Tensor)')
    # one implementation per op under the Autograd key: the Function runs the library's kernels in forward and backward, so the
    # same callable serves inference (no graph recorded) and training
    auto = torch.library.Library(_LIB, 'IMPL', 'Autograd')
    for kind, name in (('sharpen', 'manipulation_sharpen'), ('gaussian', 'manipulation_gaussian'),
                       ('resample', 'manipulation_resample'), ('gamma', 'manipulation_gamma'), ('median', 'manipulation_median')):
        auto.impl(name, (lambda k: (lambda x, s: _Manip.apply(k, x, s)))(kind))
    auto.impl('manipulation_awgn', lambda x, noise, s: _Awgn.apply(x, noise, s))
    auto.impl('max_pool2', lambda x: _MaxPool2.apply(x))
    auto.impl('conv_transpose2x2', lambda x, w, b: _ConvT2x2.apply(x, w, b))
    auto.impl('depth_to_space_clip', lambda x: _D2SClip.apply(x))
    auto.impl('constrained_conv', lambda x, k, s: _ConstrainedConv.apply(x, k, s))
    auto.impl('mse255', lambda a, b: _Mse255.apply(a, b))
    auto.impl('fan_head', lambda act, w, b, labels: _FanHead.apply(act, w, b, labels))
    # the same callables under the backend key: inference-mode callers (Autograd keys excluded) reach the kernels too
    cuda = torch.library.Library(_LIB, 'IMPL', 'CUDA')
    for kind, name in (('sharpen', 'manipulation_sharpen'), ('gaussian', 'manipulation_gaussian'),
                       ('resample', 'manipulation_resample'), ('gamma', 'manipulation_gamma'), ('median', 'manipulation_median')):
        cuda.impl(name, (lambda k: (lambda x, s: _Manip.apply(k, x, s)))(kind))
    cuda.impl('manipulation_awgn', lambda x, noise, s: _Awgn.apply(x, noise, s))
    cuda.impl('max_pool2', lambda x: _MaxPool2.apply(x))
    cuda.impl('conv_transpose2x2', lambda x, w, b: _ConvT2x2.apply(x, w, b))
    cuda.impl('depth_to_space_clip', lambda x: _D2SClip.apply(x))
    cuda.impl('constrained_conv', lambda x, k, s: _ConstrainedConv.apply(x, k, s))
    cuda.impl('mse255', lambda a, b: _Mse255.apply(a, b))
    cuda.impl('fan_head', lambda act, w, b, labels: _FanHead.apply(act, w, b, labels))
    return lib, auto, cuda


# keep the Library objects alive (dropping them de-registers the ops) - and register ONCE per process: the package is importable
# under two names ('neural-imaging_amd' and its alias 'neural_imaging_amd'), a second execution of this module must not
# define the namespace again
_HANDLES = getattr(torch, '_nimg_torch_ops_handles', None)
if _HANDLES is None:
    _HANDLES = _register() + _register_round5()
    torch._nimg_torch_ops_handles = _HANDLES


def conv2d(x, w, bias=None, stride=1, act=''):
    """Differentiable Conv2D(SAME) on the library's kernels = torch.ops.nimg.conv2d."""
    return torch.ops.nimg.conv2d(x, w, bias, stride, act)


def djpeg(x, qtab, rounding='soft'):
    """Differentiable JPEG (models/jpeg.py:91-159) as a dispatcher op: torch.ops.nimg.djpeg."""
    return torch.ops.nimg.djpeg(x, qtab, rounding)
