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


_HANDLES = _register()          # keep the Library objects alive: dropping them de-registers the ops


def conv2d(x, w, bias=None, stride=1, act=''):
    """Differentiable Conv2D(SAME) on the library's kernels = torch.ops.nimg.conv2d."""
    return torch.ops.nimg.conv2d(x, w, bias, stride, act)


def djpeg(x, qtab, rounding='soft'):
    """Differentiable JPEG (models/jpeg.py:91-159) as a dispatcher op: torch.ops.nimg.djpeg."""
    return torch.ops.nimg.djpeg(x, qtab, rounding)
