"""
Layer building blocks with explicit forward / backward passes over the HIP kernels (no autograd tape):
the counterpart of the Keras layers the reference composes (tf.keras.layers.Conv2D / Conv2DTranspose / MaxPool2D) and
of its custom layers in models/layers.py (ConstrainedConv2D :12-57, Quantization :60-172).

Convention: a layer's backward receives dz = gradient w.r.t. its PRE-activation output (the LeakyReLU derivative has
already been applied by whoever produced dz - it is fused into the producing kernel's epilogue) and
  * writes its parameter gradients into the model's flat gradient buffer,
  * returns the gradient w.r.t. its input, optionally already multiplied by the previous layer's LeakyReLU'.
"""
import numpy as np
import torch

from .. import ops
from ..helpers import kernels as hk


FUSED_ACTIVATIONS = (None, 'leaky_relu')       # what the convolution epilogues apply themselves; the other members of
                                               # helpers/tf_helpers.activation_mapping (relu, tanh, sigmoid, softsign) run as an
                                               # element-wise pass behind the (activation-free) convolution, on float32 tensors


def act_forward(y, activation):
    """activation(y) in place for an activation the kernels do not fuse."""
    if isinstance(y, tuple):                   # (out, bf16 copy): only the throughput-mode chains ask for copies
        raise RuntimeError('a bf16 copy cannot be combined with activation {}'.format(activation))
    return ops.activation(y, activation, out=y)


class Conv2D(object):
    """Conv2D(cout, ks, stride, 'SAME', activation) with (kh,kw,Cin,Cout) kernel + bias.
    mask_activation: the activation of the layer that FEEDS this one - what an `act_mask` handed to backward_input stands for
    (the models set it to their `activation` hyper-parameter; a call may override it)."""

    def __init__(self, name, ks, cin, cout, activation=None, stride=1, cin2=0, mask_activation='leaky_relu'):
        self.name, self.ks, self.cin, self.cin2, self.cout = name, ks, cin, cin2, cout
        self.activation, self.stride = activation, stride
        self.mask_activation = mask_activation

    def specs(self):
        return [(self.name + '/kernel', (self.ks, self.ks, self.cin + self.cin2, self.cout)),
                (self.name + '/bias', (self.cout,))]

    def init(self, store, gen):
        from .tfmodel import glorot_uniform_
        rf = self.ks * self.ks
        k = torch.empty(store.p[self.name + '/kernel'].shape, dtype=torch.float32)
        glorot_uniform_(k, (self.cin + self.cin2) * rf, self.cout * rf, gen)
        store.p[self.name + '/kernel'].copy_(k)
        store.p[self.name + '/bias'].zero_()

    def forward(self, store, x, x2=None, out_bf16=False, residual=None, bf16_copy=False, d2s_out=False, copy_lrelu=False,
                s2d_out=False):
        """residual: the skip tensor of a residual block, added to the layer's output in the same pass (ops.conv2d);
        bf16_copy: returns (out, bf16 copy of out or None); d2s_out / s2d_out: returns tf.nn.depth_to_space(out, 2) /
        tf.nn.space_to_depth(out, 2), see ops.conv2d."""
        fused = self.activation in FUSED_ACTIVATIONS
        y = ops.conv2d(x, store.p[self.name + '/kernel'], store.p[self.name + '/bias'], x2=x2,
                       stride=self.stride, act=self.activation if fused else None, out_bf16=out_bf16, residual=residual,
                       bf16_copy=bf16_copy, d2s_out=d2s_out, copy_lrelu=copy_lrelu, s2d_out=s2d_out)
        # (an element-wise activation commutes with the depth_to_space / space_to_depth permutations of the output)
        return y if fused else act_forward(y, self.activation)

    def can_pool(self, x):
        return self.stride == 1 and self.cin2 == 0 and self.ks in (3, 5) and self.cout % 4 == 0 and \
            x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and self.activation == 'leaky_relu'

    def forward_pool(self, store, x, want_idx=True, out_bf16=False):
        """conv -> activation -> MaxPool2D(2) in one pass; returns (pooled, argmax bytes)."""
        return ops.conv2d_pool(x, store.p[self.name + '/kernel'], store.p[self.name + '/bias'], act=self.activation,
                               want_idx=want_idx, out_bf16=out_bf16)

    def forward_and_pool(self, store, x, out_bf16=False):
        """-> (activation, MaxPool2D(2) of it): one pass where the bf16 epilogue can write both (ops.conv2d_and_pool)."""
        w = store.p[self.name + '/kernel']
        if out_bf16 and self.can_pool(x) and ops.conv2d_and_pool_ok(x, w):
            return ops.conv2d_and_pool(x, w, store.p[self.name + '/bias'], act=self.activation)
        y = self.forward(store, x, out_bf16=out_bf16)
        return y, ops.maxpool2(y)

    def backward_params(self, store, x, dz, x2=None):
        # on the side stream: it only needs (x, dz), which the input gradient on the launch stream reads as well
        ops.conv2d_wgrad(x, dz, self.ks, x2=x2, stride=self.stride, dw=store.g[self.name + '/kernel'],
                         db=store.g[self.name + '/bias'], side=True)

    # -- a 5x5 stride-2 layer fed by a bf16 SPACE-TO-DEPTH image of its input (the codec's e2 behind e1, models/compression.py:
    # 217-220): forward and weight gradient run as the equivalent 3x3 stride-1 layer (ops.s2d_conv_weights) on the bf16-input
    # MFMA kernels instead of the float32 strided ones; the input gradient (backward_input, mask_s2d) already is that layer's.
    def s2d_chain_ok(self, in_hw):
        return self.stride == 2 and self.ks == 5 and self.cin2 == 0 and self.cin % 8 == 0 and self.cout % 8 == 0 and \
            ops.COMPUTE == 'bf16' and ops.STORE_BF16 and ops.D2S_EPILOGUE and \
            ops.s2d_conv_ok(self.ks, self.stride, in_hw[0], in_hw[1], self.cin)

    def forward_s2d(self, store, xs, bf16_copy=False, copy_lrelu=False):
        w3 = ops.s2d_conv_weights(store.p[self.name + '/kernel'])
        fused = self.activation in FUSED_ACTIVATIONS
        y = ops.conv2d(xs, w3, store.p[self.name + '/bias'], act=self.activation if fused else None, bf16_copy=bf16_copy,
                       copy_lrelu=copy_lrelu)
        return y if fused else act_forward(y, self.activation)

    def backward_params_s2d(self, store, xs, dz):
        with ops.side_stream(xs, dz, key=store.g[self.name + '/kernel'].data_ptr()):
            dw3 = ops.conv2d_wgrad(xs, dz, 3, db=store.g[self.name + '/bias'])
            ops.s2d_conv_weights_bwd(dw3, store.g[self.name + '/kernel'])

    def backward_input(self, store, dz, in_hw, act_mask=None, out=None, out2=None, out_bf16=False, residual=None,
                       bf16_copy=False, s2d_out=False, mask_s2d=False, mask_activation=None):
        """residual: the gradient arriving over the block's skip connection, added in the same pass (stride-1 layers);
        bf16_copy: returns (gradient, bf16 copy of it or None); s2d_out: returns tf.nn.space_to_depth(gradient, 2) - the
        gradient at the input of the depth_to_space layer that fed this one (stride-1 layers), see ops.conv2d.
        mask_activation: overrides the layer's (the codec's first block is fed tf.nn.leaky_relu whatever `activation` says)."""
        kind = mask_activation or self.mask_activation
        if act_mask is not None and kind not in FUSED_ACTIVATIONS:
            # an activation the epilogues do not know: the plain gradient, its derivative taken from the stored output, then
            # what the caller would have had fused behind it - float32 tensors (the models keep float32 storage in this case)
            if out2 is not None or mask_s2d:
                raise NotImplementedError('split / space-to-depth-stored masks with activation {}'.format(kind))
            d = self.backward_input(store, dz, in_hw, out=out)
            ops.activation_bwd(d, act_mask, kind, out=d)         # the mask = this layer's input activation, in d's layout
            if residual is not None:
                ops.add(residual, d, out=d)
            if s2d_out:
                d = ops.d2s_clip_bwd(d, 1.0)
            return (d, None) if bf16_copy else d
        if self.stride == 2:
            d = ops.conv2d_dgrad_strided2(dz, store.p[self.name + '/kernel'], in_hw, act_mask=act_mask, out_bf16=out_bf16,
                                          mask_s2d=mask_s2d)
            d = d if residual is None else ops.add(residual, d, out=d)
            return (d, None) if bf16_copy else d
        return ops.conv2d_dgrad(dz, store.p[self.name + '/kernel'], in_hw, stride=self.stride, act_mask=act_mask,
                                out=out, out2=out2, out_bf16=out_bf16, residual=residual, bf16_copy=bf16_copy,
                                s2d_out=s2d_out)


class Conv5x5Stride2Image(Conv2D):
    """The codec's first layer, Conv2D(64, 5, strides=2, 'SAME') on the RGB image (models/compression.py:217): same (5,5,3,64)
    kernel + bias parameters.  In throughput mode it runs as a 3x3 stride-1 convolution over the bf16 space-to-depth image of
    a x + b (12 of 16 block channels, ops.s2d2_affine / s2d_conv_weights): forward, weight gradient (gathered back into the 5x5
    layout) and input gradient all take the stride-1 MFMA kernels instead of the float32 small-channel ones (0.46 + 1.53 ms per
    step at B = 50, and 3.2 ms for the zero-stuffed input gradient at B = 80).  Parity mode: the plain strided layer."""

    def s2d_ok(self, x):
        return ops.s2d_conv_ok(self.ks, self.stride, x.shape[1], x.shape[2], self.cin) and self.cin2 == 0 and 4 * self.cin <= 16

    def forward_image(self, store, x, a, b, s2d_out=False):
        """-> (y, ctx): the layer applied to a x + b; ctx is what the backward passes need (the block image or a x + b).
        s2d_out (only where s2d_ok): y is returned as its bf16 space-to-depth image - the form the next strided layer reads."""
        if not self.s2d_ok(x):
            if s2d_out:
                raise RuntimeError('s2d_out needs the space-to-depth form of the layer')
            x0 = ops.affine(x, a, b)
            return self.forward(store, x0), x0
        xs = ops.s2d2_affine(x, a, b, cp=16)
        w3 = ops.s2d_conv_weights(store.p[self.name + '/kernel'], cp=16)
        fused = self.activation in FUSED_ACTIVATIONS
        y = ops.conv2d(xs, w3, store.p[self.name + '/bias'], act=self.activation if fused else None, s2d_out=s2d_out,
                       out_bf16=s2d_out)
        return (y if fused else act_forward(y, self.activation)), xs

    def backward_params_image(self, store, ctx, dz):
        if ctx.dtype != torch.bfloat16:
            return self.backward_params(store, ctx, dz)
        with ops.side_stream(ctx, dz, key=store.g[self.name + '/kernel'].data_ptr()):
            dw3 = ops.conv2d_wgrad(ctx, dz, 3, db=store.g[self.name + '/bias'])
            ops.s2d_conv_weights_bwd(dw3, store.g[self.name + '/kernel'])

    def backward_input_image(self, store, dz, in_hw, a):
        """d loss / d x of y = layer(a x + b)."""
        return ops.conv2d_dgrad_strided2(dz, store.p[self.name + '/kernel'], in_hw, scale=a)


class Conv2DTranspose2x2(object):
    """Conv2DTranspose(cout, [2,2], [2,2], 'SAME'), kernel (2,2,Cout,Cin) + bias, no activation (pipelines.py:205)."""

    def __init__(self, name, cin, cout):
        self.name, self.cin, self.cout = name, cin, cout

    def specs(self):
        return [(self.name + '/kernel', (2, 2, self.cout, self.cin)), (self.name + '/bias', (self.cout,))]

    def init(self, store, gen):
        from .tfmodel import glorot_uniform_
        k = torch.empty((2, 2, self.cout, self.cin), dtype=torch.float32)
        glorot_uniform_(k, self.cin * 4, self.cout * 4, gen)
        store.p[self.name + '/kernel'].copy_(k)
        store.p[self.name + '/bias'].zero_()

    def forward(self, store, x, out_bf16=False):
        return ops.convt2x2(x, store.p[self.name + '/kernel'], store.p[self.name + '/bias'], out_bf16=out_bf16)

    def backward_params(self, store, x, dy):
        ops.convt2x2_wgrad(x, dy, dw=store.g[self.name + '/kernel'], side=True)
        ops.bias_grad(dy, db=store.g[self.name + '/bias'], side=True)

    def backward_input(self, store, dy, act_mask=None, out_bf16=False, mask_activation='leaky_relu'):
        if act_mask is not None and mask_activation not in FUSED_ACTIVATIONS:
            d = ops.convt2x2_dgrad(dy, store.p[self.name + '/kernel'])
            return ops.activation_bwd(d, act_mask, mask_activation, out=d)
        return ops.convt2x2_dgrad(dy, store.p[self.name + '/kernel'], act_mask=act_mask, out_bf16=out_bf16)


class ConstrainedConv2D(object):
    """Trainable constrained residual filter (models/layers.py:12-57): the (5,5,3,3) kernel is re-normalised on every
    call (centre taps fixed to -strength, every output channel sums to 0), input padded SYMMETRIC, VALID conv."""

    def __init__(self, name='constrained', filter_strength=100.0):
        self.name, self.strength = name, float(filter_strength)

    def specs(self):
        return [(self.name + '/kernel', (5, 5, 3, 3))]

    def init(self, store, gen=None):
        store.p[self.name + '/kernel'].copy_(torch.from_numpy(hk.residual_init_filter().astype(np.float32)))

    def forward(self, store, x, c4_only=False):
        """-> (y, normalised filter): y float32 (N,H,W,3), or - c4_only, the throughput-mode front end - the same image as
        bf16 {y0,y1,y2,1} pixels (N,H,W,4), the only form the row-band conv1 kernels read."""
        nf = ops.constrained_kernel(store.p[self.name + '/kernel'], self.strength)
        y, c4 = ops.cconv3(x, nf, pad_mode=1, want_f32=not c4_only, want_c4=c4_only)
        return (c4 if c4_only else y), nf

    def backward_params(self, store, x, dy):
        # on the side stream like every other parameter gradient (it only needs x and dy; the input gradient runs beside it)
        with ops.side_stream(x, dy, key=store.g[self.name + '/kernel'].data_ptr()):
            dnf = ops.conv2d_wgrad(x, dy, 5, pads=(2, 2), pad_mode=1)
            ops.constrained_kernel_bwd(store.p[self.name + '/kernel'], dnf, store.g[self.name + '/kernel'], self.strength)

    def backward_input(self, nf, dy):
        # correlation with the flipped filter + the terms the SYMMETRIC pad folds back onto the image border
        n, h, w, _ = dy.shape
        if h < 4 or w < 4:
            dpad = ops.conv2d(dy, ops.flip_weights(nf), None, pads=(4, 4), out_hw=(h + 4, w + 4))
            return ops.fold_pad(dpad, 2, 1)
        return ops.cconv3_dgrad(dy, nf)
