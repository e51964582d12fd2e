"""
Neural imaging pipelines (camera ISPs) on the HIP kernels.  Mirrors the reference's models/pipelines.py:
NIPModel (:27-166) - loss selection, training_step, process - UNet (:169-230), INet (:233-295), DNet (:298-349); ONet
(:353-362) is the identity ISP.  ClassicISP is SURVEY 8(f) "next".

UNet graph (pipelines.py:190-223): encoder n=1..5: 2 x [Conv3x3 SAME, 32*2^(n-1), LeakyReLU(0.2)] + MaxPool2 (not after
n=5); decoder n=1..4: ConvT 2x2 s2 -> concat(up, skip) [never materialised: the conv reads two tensors] -> 2 x
Conv3x3+LReLU; Conv3x3 -> 12 (linear) -> depth_to_space(2) -> straight-through clip.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from .. import ops, parallel
from ..device import DeviceArray, to_device
from ..helpers import kernels as hk
from ..helpers import paramspec, utils
from .layers import Conv2D, Conv2DTranspose2x2
from .tfmodel import ParamStore, TFModel


class NIPModel(TFModel):
    """Abstract neural imaging pipeline: RAW (N,h,w,4) -> RGB (N,2h,2w,3)."""

    def __init__(self, loss_metric='L2', patch_size=None, in_channels=4, device=None, seed=1234, **kwargs):
        super().__init__(device=device)
        self.in_channels = in_channels
        self.patch_size = patch_size
        self.x = _Placeholder((None, patch_size, patch_size, in_channels))
        self._seed = seed
        self.construct_model(**kwargs)
        self._has_attributes(['y', '_model'])
        self.loss_metric = loss_metric
        self.construct_loss(loss_metric)

    def construct_loss(self, loss_metric):
        """L2 | L1 | SSIM | MS-SSIM on 255-scaled images (pipelines.py:53-63 -> helpers/tf_helpers.py:31-44)."""
        if loss_metric in ops.IMAGE_LOSSES:
            self._loss_fn = ops.IMAGE_LOSSES[loss_metric]
            self.loss = lambda a, b: DeviceArray(self._loss_fn(to_device(a, self.device), to_device(b, self.device))[0])
        else:
            raise ValueError('Unsupported loss metric!')

    def loss_and_grad(self, y, target, grad_scale=1.0, grad_out=None, accumulate=False):
        """(loss[1], grad_scale * d loss / d y) of the configured metric; grad_out (+)= when accumulate."""
        return self._loss_fn(y, target, grad_scale=grad_scale, grad_out=grad_out, accumulate=accumulate)

    def construct_model(self):
        raise NotImplementedError()

    # forward/backward used by the workflow ----------------------------------------------------------------------
    def forward(self, x, training=False):
        raise NotImplementedError()

    def backward(self, ctx, dy):
        raise NotImplementedError()

    def training_step(self, batch_x, batch_y, learning_rate=None):
        """One step on the configured loss of (255 Y, 255 y) (pipelines.py:77-90). Returns the loss."""
        x = to_device(batch_x, self.device)
        t = to_device(batch_y, self.device)
        y, ctx = self.forward(x, training=True)
        loss, dy = self.loss_and_grad(y, t)
        self.backward(ctx, dy)
        world = parallel.sync_gradients(self._model.flat_grad)       # data parallel: the loss is a mean over the batch
        if learning_rate is not None:
            self.learning_rate = learning_rate
        self._model.adam(self.learning_rate, grad_scale=1.0 / world)
        return DeviceArray(loss)

    learning_rate = 1e-3      # tf.keras.optimizers.Adam() default

    def process(self, batch_x, training=False):
        x = to_device(batch_x, self.device)
        if x.dim() == 3:
            x = x.unsqueeze(0).contiguous()
        return DeviceArray(self.forward(x, training=False)[0])

    def reset_performance_stats(self):
        self.performance = {'loss': {'training': [], 'validation': []}, 'psnr': {'validation': []},
                            'ssim': {'validation': []}}

    def get_hyperparameters(self):
        p = {'in_channels': self.in_channels}
        if hasattr(self, '_h'):
            p.update(self._h.to_json())
        return p

    @property
    def patch_size_raw(self):
        return self.x.shape[1:]

    @property
    def patch_size_rgb(self):
        return self.y.shape[1:]

    @property
    def _input_description(self):
        return utils.format_patch_shape(self.patch_size_raw)

    @property
    def _output_description(self):
        return utils.format_patch_shape(self.patch_size_rgb)

    def summary(self):
        return '{:s} : {} -> {}'.format(super().summary(), self._input_description, self._output_description)

    def load_model(self, dirname, quiet=False):
        if '/' not in dirname:                           # a bare camera / model name lives under the NIP snapshot root (:133-136)
            dirname = os.path.join('data/models/nip', dirname)
        super().load_model(dirname, quiet=quiet)

    def save_model(self, dirname, epoch=0, save_args=False, quiet=False):
        if '/' not in dirname:
            dirname = os.path.join('data/models/nip', dirname)
        super().save_model(dirname, epoch=epoch, save_args=save_args, quiet=quiet)


class _Placeholder(object):
    """Stands in for the tf.keras.Input / output tensors callers only query for .shape."""

    def __init__(self, shape):
        self.shape = tuple(shape)


class UNet(NIPModel):

    def construct_model(self, **kwargs):
        self._h = paramspec.ParamSpec({
            'n_steps': (5, int, (2, 6)),
            'activation': ('leaky_relu', str, set(ops.ACTIVATIONS)),          # helpers/tf_helpers.py:22-28 (pipelines.py:179)
        })
        self._h.update(**kwargs)
        ns = self._h.n_steps
        act = self._h.activation
        self._layers = OrderedDict()
        cin = self.in_channels
        for n in range(1, ns + 1):
            c = 32 * 2 ** (n - 1)
            self._layers['ec{}1'.format(n)] = Conv2D('ec{}1'.format(n), 3, cin, c, act, mask_activation=act)
            self._layers['ec{}2'.format(n)] = Conv2D('ec{}2'.format(n), 3, c, c, act, mask_activation=act)
            cin = c
        for n in range(1, ns):
            c = 32 * 2 ** (ns - n - 1)
            self._layers['dct{}'.format(n)] = Conv2DTranspose2x2('dct{}'.format(n), cin, c)
            self._layers['dc{}1'.format(n)] = Conv2D('dc{}1'.format(n), 3, c, c, act, cin2=c, mask_activation=act)
            self._layers['dc{}2'.format(n)] = Conv2D('dc{}2'.format(n), 3, c, c, act, mask_activation=act)
            cin = c
        self._layers['dc{}'.format(ns)] = Conv2D('dc{}'.format(ns), 3, cin, 12, None, mask_activation=act)
        specs = [s for l in self._layers.values() for s in l.specs()]
        self._model = ParamStore(specs, self.device)
        gen = torch.Generator().manual_seed(self._seed)
        for l in self._layers.values():
            l.init(self._model, gen)
        ps = self.patch_size
        self.y = _Placeholder((None, None if ps is None else 2 * ps, None if ps is None else 2 * ps, 3))

    @property
    def model_code(self):
        return '{}_{}'.format(self.class_name, self._h.n_steps)

    def _store_bf16(self, x):
        """Throughput mode keeps the UNet's internal activations and gradients in HBM as bf16 (like the FAN's): every consumer
        is a kernel that rounds them to bf16 MFMA operands, takes their sign (LeakyReLU') or their maximum (rounding is
        monotonic), so the FORWARD pass is bit-neutral and the level-1 / level-2 layers - HBM-bound at float32 - move half the
        bytes.  The BACKWARD pass is not bit-neutral: the skip gradients are rounded to bf16 before they are accumulated, and
        maxpool2_bwd_bf16 routes to the first maximum of the ROUNDED activations - rounding creates ties that float32 does not
        have, so a few gradients take another window position than in the float32-stored path (and than tf's max-pool gradient);
        per-parameter cosine > 0.995 against the float32-stored run (tests/test_gpu_models.py::test_unet_bf16_storage).  The RAW input, the 12-channel output of the last convolution and the gradient that feeds the 4-channel
        weight-gradient kernel stay float32.  (forward() also asks for even sizes at every pooled level.)"""
        return ops.COMPUTE == 'bf16' and ops.STORE_BF16 and x.is_cuda and self._h.activation == 'leaky_relu'
        # (another activation runs as a float32 element-wise pass behind each convolution: float32 storage)

    writes_into = True            # forward(..., out=) develops straight into a caller's buffer (the workflow's class batch)

    def forward(self, x, training=False, out=None):
        self._model.refresh_images()
        L, P, ns = self._layers, self._model, self._h.n_steps
        sb = self._store_bf16(x) and x.shape[1] % (1 << (ns - 1)) == 0 and x.shape[2] % (1 << (ns - 1)) == 0
        t = OrderedDict()
        t['ep0'] = x
        for n in range(1, ns + 1):
            t['ec{}1'.format(n)] = L['ec{}1'.format(n)].forward(P, t['ep{}'.format(n - 1)], out_bf16=sb)
            if n < ns:          # the skip tensor and the next level's input (bf16 storage: both from the same epilogue)
                t['ec{}2'.format(n)], t['ep{}'.format(n)] = L['ec{}2'.format(n)].forward_and_pool(P, t['ec{}1'.format(n)],
                                                                                                   out_bf16=sb)
            else:
                t['ec{}2'.format(n)] = L['ec{}2'.format(n)].forward(P, t['ec{}1'.format(n)], out_bf16=sb)
        t['dc02'] = t['ec{}2'.format(ns)]
        for n in range(1, ns):
            t['dct{}'.format(n)] = L['dct{}'.format(n)].forward(P, t['dc{}2'.format(n - 1)], out_bf16=sb)
            t['dc{}1'.format(n)] = L['dc{}1'.format(n)].forward(P, t['dct{}'.format(n)], t['ec{}2'.format(ns - n)],
                                                              out_bf16=sb)
            t['dc{}2'.format(n)] = L['dc{}2'.format(n)].forward(P, t['dc{}1'.format(n)], out_bf16=sb)
        last, xin = L['dc{}'.format(ns)], t['dc{}2'.format(ns - 1)]
        if ops.rows_d2s_ok(xin, P.p[last.name + '/kernel']):
            # throughput mode at 128-pixel rows: the 12-channel tensor is written as its clipped depth_to_space image by the
            # convolution itself (csrc/conv3_rows.hip); the backward pass never reads it (the clip is straight-through)
            t['dc{}'.format(ns)] = None
            y = ops.conv3_rows_d2s(xin, P.p[last.name + '/kernel'], P.p[last.name + '/bias'], out=out)
        else:
            t['dc{}'.format(ns)] = last.forward(P, xin)
            y = ops.d2s_clip(t['dc{}'.format(ns)], 1.0, 0.0, True, out=out)
        return y, (t if training else None)

    def decoder_grads(self):
        """The decoder's slice of the flat gradient buffer (complete when the decoder backward is done), the encoder's."""
        return self._model.grad_range(first='dct1/kernel'), self._model.grad_range(before='dct1/kernel')

    def head_gradient(self, parts, y, target, grad_scale):
        """The workflow's hand-over in one pass (where it applies, else None): -> (L2 loss[1], dz_head) with dz_head = the gradient
        behind depth_to_space + clip of sum(parts) + grad_scale * d L2 / d y, to be passed to backward(dz_head=)."""
        if not ops.FUSED_HEAD_GRAD or self.loss_metric != 'L2' or not 1 <= len(parts) <= 6 or y.dtype != torch.float32 or \
                any(p.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() % 8 for p in parts):
            return None
        return ops.mse255_sum_s2d3(parts, y, target, grad_scale)

    def backward(self, t, dy, on_decoder_done=None, dz_head=None):
        """dy = d loss / d y (N,2h,2w,3).  Fills the gradient buffer; the RAW input needs no gradient.
        on_decoder_done(): called once every decoder gradient has been queued (data parallelism all-reduces that slice
        while the encoder backward runs).  dz_head: head_gradient()'s result instead of dy."""
        L, P, ns = self._layers, self._model, self._h.n_steps
        hw = lambda a: (a.shape[1], a.shape[2])
        sb = t['ec12'].dtype == torch.bfloat16              # the forward pass stored its activations as bf16: so are the gradients
        # head: d2s + clip are straight-through
        dz = ops.d2s_clip_bwd(dy, 1.0) if dz_head is None else dz_head
        last = 'dc{}2'.format(ns - 1)
        # the weight gradients of a level are issued on the side streams behind ONE fork, after that level's input gradients
        # are queued (ops.ParamGroup: a fork is a marker packet between two kernels of the launch stream, 6 - 8 us each)
        grp = ops.ParamGroup()
        grp.add(lambda dz=dz: L['dc{}'.format(ns)].backward_params(P, t[last], dz))
        dz = L['dc{}'.format(ns)].backward_input(P, dz, hw(t[last]), act_mask=t[last], out_bf16=sb)   # dZ of dc{ns-1}2
        d_skip = {}
        for n in range(ns - 1, 0, -1):
            a1, up, skip = t['dc{}1'.format(n)], t['dct{}'.format(n)], t['ec{}2'.format(ns - n)]
            grp.add(lambda n=n, a1=a1, dz=dz: L['dc{}2'.format(n)].backward_params(P, a1, dz))
            dz1 = L['dc{}2'.format(n)].backward_input(P, dz, hw(a1), act_mask=a1, out_bf16=sb)      # dZ of dc{n}1
            grp.add(lambda n=n, up=up, dz1=dz1, skip=skip: L['dc{}1'.format(n)].backward_params(P, up, dz1, x2=skip))
            d_up = torch.empty_like(up)
            d_sk = torch.empty_like(skip)
            L['dc{}1'.format(n)].backward_input(P, dz1, hw(up), out=d_up, out2=d_sk)
            d_skip[ns - n] = d_sk
            prev = t['dc{}2'.format(n - 1)]
            grp.add(lambda n=n, prev=prev, d_up=d_up: L['dct{}'.format(n)].backward_params(P, prev, d_up))
            grp.flush()                    # (in front of the level's last input gradient: everything the group reads is queued)
            dz = L['dct{}'.format(n)].backward_input(P, d_up, act_mask=prev, out_bf16=sb,
                                                     mask_activation=self._h.activation)          # dZ of dc{n-1}2 / ec{ns}2
        grp.flush()
        if on_decoder_done is not None:
            on_decoder_done()
        for n in range(ns, 0, -1):
            a1, inp = t['ec{}1'.format(n)], t['ep{}'.format(n - 1)]
            grp.add(lambda n=n, a1=a1, dz=dz: L['ec{}2'.format(n)].backward_params(P, a1, dz))
            # the first layer's weight gradient (4 input channels: the (tap, ci)-packed kernel) stages float32
            dz1 = L['ec{}2'.format(n)].backward_input(P, dz, hw(a1), act_mask=a1, out_bf16=sb and n > 1)
            grp.add(lambda n=n, inp=inp, dz1=dz1: L['ec{}1'.format(n)].backward_params(P, inp, dz1))
            grp.flush()
            if n > 1:
                prev = t['ec{}2'.format(n - 1)]
                w1 = P.p['ec{}1/kernel'.format(n)]
                if sb and self._h.activation == 'leaky_relu' and ops.conv2d_dgrad_unpool_out_ok(dz1, w1, prev, d_skip[n - 1]):
                    # the input gradient written through the max-pool: route + skip sum + LeakyReLU' in the epilogue (bf16 storage)
                    dz = ops.conv2d_dgrad_unpool_out(dz1, w1, prev, skip=d_skip[n - 1], apply_mask=True, out=d_skip[n - 1])
                    continue
                d_pool = L['ec{}1'.format(n)].backward_input(P, dz1, hw(inp), out_bf16=sb)
                if self._h.activation == 'leaky_relu':
                    dz = ops.maxpool2_bwd(d_pool, prev, add=d_skip[n - 1], apply_mask=True, out=d_skip[n - 1])
                else:           # route + skip sum, then the activation's derivative from its stored output
                    dz = ops.maxpool2_bwd(d_pool, prev, add=d_skip[n - 1], apply_mask=False, out=d_skip[n - 1])
                    ops.activation_bwd(dz, prev, self._h.activation, out=dz)
        ops.join_side_stream()
        return None


class INet(NIPModel):
    """The standard-ISP-shaped pipeline (pipelines.py:233-295): 1x1 CFA up-sampling (4 -> 12, frozen unless
    trainable_upsampling) -> depth_to_space(2) -> REFLECT-padded k x k demosaicing (3 -> 3, no bias) -> 1x1 colour
    conversion (3 -> 3, no bias) -> gamma MLP 1x1 (3 -> 12, tanh) + 1x1 (12 -> 3) -> straight-through clip.
    Layer names: up, demosaic, srgb, gamma1, gamma2 (Keras auto-names conv2d_<k> depend on creation order in the
    process and are not reproducible)."""

    def construct_model(self, random_init=False, kernel=5, trainable_upsampling=False, cfa_pattern='gbrg'):
        self._h = paramspec.ParamSpec({
            'random_init': (False, bool, None),
            'kernel': (5, int, (3, 11)),
            'trainable_upsampling': (False, bool, None),
            'cfa_pattern': ('gbrg', str, {'gbrg', 'rggb', 'bggr'}),
        })
        self._h.update(random_init=random_init, kernel=kernel, trainable_upsampling=trainable_upsampling,
                       cfa_pattern=cfa_pattern)
        if self._h.kernel % 2 == 0:
            raise NotImplementedError('even demosaicing kernel {}: the reference pads (k - 1) // 2 and convolves VALID - its output '
                                      'is one pixel short of the target; 3, 5, 7, 9, 11 are built'.format(self._h.kernel))
        if self.in_channels != 4:
            raise ValueError('INet develops 4-plane RAW input')
        k = self._h.kernel
        specs = [('up/kernel', (1, 1, 4, 12)), ('demosaic/kernel', (k, k, 3, 3)), ('srgb/kernel', (1, 1, 3, 3)),
                 ('gamma1/kernel', (1, 1, 3, 12)), ('gamma1/bias', (12,)), ('gamma2/kernel', (1, 1, 12, 3)),
                 ('gamma2/bias', (3,))]
        self._model = ParamStore(specs, self.device)
        if self._h.random_init:
            rng = np.random.RandomState(self._seed)
            dmf = rng.normal(0, 0.1, (k, k, 3, 3))
            g1k, g1b, g2k, g2b = rng.normal(0, 0.1, (3, 12)), np.zeros(12), rng.normal(0, 0.1, (12, 3)), np.zeros(3)
            srgbk = np.eye(3)
        else:
            dmf = hk.bilin_kernel(k)
            g1k, g1b, g2k, g2b = hk.gamma_kernels()
            srgbk = hk.SRGB_EXAMPLE
        init = {'up/kernel': hk.upsampling_kernel(self._h.cfa_pattern), 'demosaic/kernel': dmf, 'srgb/kernel': srgbk,
                'gamma1/kernel': g1k, 'gamma1/bias': g1b, 'gamma2/kernel': g2k, 'gamma2/bias': g2b}
        for name, v in init.items():
            p = self._model.p[name]
            p.copy_(torch.from_numpy(np.asarray(v, np.float32).reshape(tuple(p.shape))))
        ps = self.patch_size
        self.y = _Placeholder((None, None if ps is None else 2 * ps, None if ps is None else 2 * ps, 3))

    @property
    def model_code(self):
        return '{c}_{cfa}{tu}{r}_{k}x{k}'.format(c=self.class_name, cfa=self._h.cfa_pattern, k=self._h.kernel,
                                                 tu='T' if self._h.trainable_upsampling else '',
                                                 r='R' if self._h.random_init else '')

    def forward(self, x, training=False):
        self._model.refresh_images()
        P = self._model.p
        t = OrderedDict()
        h12 = ops.conv2d(x, P['up/kernel'])
        t['x'] = x
        t['bayer'] = ops.d2s_clip(h12, 1.0, 0.0, False)
        t['rgb'] = ops.conv2d(t['bayer'], P['demosaic/kernel'], pad_mode=ops.PAD_MODES['REFLECT'])
        t['srgb'] = ops.conv2d(t['rgb'], P['srgb/kernel'])
        t['g0'] = ops.tanh(ops.conv2d(t['srgb'], P['gamma1/kernel'], P['gamma1/bias']))
        y0 = ops.conv2d(t['g0'], P['gamma2/kernel'], P['gamma2/bias'])
        return ops.clip01(y0, out=y0), (t if training else None)

    def backward(self, t, dy):
        """dy = d loss / d y (the clip is straight-through). The up-sampling kernel keeps a zero gradient unless
        trainable_upsampling is set."""
        P, G = self._model.p, self._model.g
        hw = (dy.shape[1], dy.shape[2])
        k = self._h.kernel
        ops.conv2d_wgrad(t['g0'], dy, 1, dw=G['gamma2/kernel'], db=G['gamma2/bias'])
        dz1 = ops.tanh_bwd(ops.conv2d_dgrad(dy, P['gamma2/kernel'], hw), t['g0'])
        ops.conv2d_wgrad(t['srgb'], dz1, 1, dw=G['gamma1/kernel'], db=G['gamma1/bias'])
        d_srgb = ops.conv2d_dgrad(dz1, P['gamma1/kernel'], hw)
        ops.conv2d_wgrad(t['rgb'], d_srgb, 1, dw=G['srgb/kernel'])
        d_rgb = ops.conv2d_dgrad(d_srgb, P['srgb/kernel'], hw)
        ops.conv2d_wgrad(t['bayer'], d_rgb, k, pad_mode=ops.PAD_MODES['REFLECT'], dw=G['demosaic/kernel'])
        if self._h.trainable_upsampling:
            # through the REFLECT-padded demosaicing: full correlation with the flipped filter on the padded domain, the pad
            # folded back, depth_to_space undone, then the 1x1 filter gradient
            dpad = ops.conv2d(d_rgb, ops.flip_weights(P['demosaic/kernel']), None, pads=(k - 1, k - 1),
                              out_hw=(hw[0] + k - 1, hw[1] + k - 1))
            d_bayer = ops.fold_pad(dpad, (k - 1) // 2, ops.PAD_MODES['REFLECT'])
            ops.conv2d_wgrad(t['x'], ops.d2s_clip_bwd(d_bayer, 1.0), 1, dw=G['up/kernel'])
        ops.join_side_stream()
        return None


class DNet(NIPModel):
    """Joint demosaicing & denoising pipeline after Gharbi et al. 2016 (pipelines.py:298-349): n_layers x [VALID k x k
    conv + ReLU -> REFLECT re-pad] on the RAW planes (last layer 12 channels) -> depth_to_space(2) -> concat with the
    up-sampled Bayer image (never materialised: the projection reads two tensors) -> VALID k x k conv + ReLU (6 -> nf) ->
    REFLECT re-pad -> 1x1 (nf -> 3, ones, no bias) -> straight-through clip.  Layer names: conv0 .. conv{n-1}, up
    (frozen), proj, out."""

    def construct_model(self, n_layers=15, kernel=3, n_features=64):
        self._h = paramspec.ParamSpec({
            'n_layers': (15, int, (1, 32)),
            'kernel': (3, int, (3, 11)),
            'n_features': (64, int, (4, 128)),
        })
        self._h.update(n_layers=n_layers, kernel=kernel, n_features=n_features)
        if self._h.kernel % 2 == 0 or self.in_channels != 4:
            # an even kernel has no working behaviour in the reference: VALID k x k followed by a REFLECT re-pad of (k - 1) // 2
            # (pipelines.py:319-322) returns images one pixel short per layer of what the loss compares them with
            raise NotImplementedError('DNet is built for odd kernels 3 .. 11 on 4-plane RAW input')
        k, nf, nl = self._h.kernel, self._h.n_features, self._h.n_layers
        self._convs = []
        cin = 4
        for r in range(nl):
            cout = 12 if r == nl - 1 else nf
            self._convs.append(Conv2D('conv{}'.format(r), k, cin, cout, None))
            cin = cout
        self._proj = Conv2D('proj', k, 3, nf, None, cin2=3)
        specs = [sp for c in self._convs for sp in c.specs()]
        specs += [('up/kernel', (1, 1, 4, 12))] + self._proj.specs() + [('out/kernel', (1, 1, nf, 3))]
        self._model = ParamStore(specs, self.device)
        gen = torch.Generator().manual_seed(self._seed)
        for c in self._convs + [self._proj]:         # VarianceScaling(1, fan_in, truncated normal), zero bias
            w = self._model.p[c.name + '/kernel']
            fan_in = c.ks * c.ks * (c.cin + c.cin2)
            t = torch.empty(tuple(w.shape), dtype=torch.float32)
            torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
            w.copy_(t * (np.sqrt(1.0 / fan_in) / 0.87962566103423978))
            self._model.p[c.name + '/bias'].zero_()
        self._model.p['up/kernel'].copy_(torch.from_numpy(hk.upsampling_kernel().astype(np.float32).reshape(1, 1, 4, 12)))
        self._model.p['out/kernel'].fill_(1.0)
        ps = self.patch_size
        self.y = _Placeholder((None, None if ps is None else 2 * ps, None if ps is None else 2 * ps, 3))

    @property
    def model_code(self):
        return '{c}_{k}x{k}_{l}x{f}f'.format(c=self.class_name, k=self._h.kernel, f=self._h.n_features,
                                             l=self._h.n_layers)

    def forward(self, x, training=False):
        self._model.refresh_images()
        P = self._model.p
        pad = (self._h.kernel - 1) // 2
        t = OrderedDict()
        t['in0'] = x
        deep = x
        for r, c in enumerate(self._convs):
            z = ops.conv2d(deep, P[c.name + '/kernel'], P[c.name + '/bias'], padding='VALID', act='relu')
            t['z{}'.format(r)] = z
            deep = ops.pad2d(z, pad, 'REFLECT')
            t['in{}'.format(r + 1)] = deep
        bayer = ops.d2s_clip(ops.conv2d(x, P['up/kernel']), 1.0, 0.0, False)
        feat = ops.d2s_clip(deep, 1.0, 0.0, False)
        t['feat'], t['bayer'] = feat, bayer
        pu = ops.conv2d(feat, P['proj/kernel'], P['proj/bias'], x2=bayer, padding='VALID', act='relu')
        t['pu'] = pu
        t['pup'] = ops.pad2d(pu, pad, 'REFLECT')
        y0 = ops.conv2d(t['pup'], P['out/kernel'])
        return ops.clip01(y0, out=y0), (t if training else None)

    def backward(self, t, dy):
        P, G = self._model.p, self._model.g
        k = self._h.kernel
        pad = (k - 1) // 2
        refl = ops.PAD_MODES['REFLECT']
        hw = lambda a: (a.shape[1], a.shape[2])
        ops.conv2d_wgrad(t['pup'], dy, 1, dw=G['out/kernel'])
        d_pu = ops.fold_pad(ops.conv2d_dgrad(dy, P['out/kernel'], hw(t['pup'])), pad, refl)
        dz = ops.lrelu_bwd(d_pu, t['pu'], alpha=0.0)
        ops.conv2d_wgrad(t['feat'], dz, k, x2=t['bayer'], padding='VALID', dw=G['proj/kernel'], db=G['proj/bias'])
        d_feat = torch.empty_like(t['feat'])
        d_bayer = torch.empty_like(t['bayer'])               # gradient of the frozen up-sampling branch: discarded
        ops.conv2d_dgrad(dz, P['proj/kernel'], hw(t['feat']), padding='VALID', out=d_feat, out2=d_bayer)
        d_deep = ops.d2s_clip_bwd(d_feat, 1.0)
        for r in range(len(self._convs) - 1, -1, -1):
            c = self._convs[r]
            z, inp = t['z{}'.format(r)], t['in{}'.format(r)]
            dz = ops.lrelu_bwd(ops.fold_pad(d_deep, pad, refl), z, alpha=0.0)
            ops.conv2d_wgrad(inp, dz, k, padding='VALID', dw=G[c.name + '/kernel'], db=G[c.name + '/bias'])
            if r > 0:
                d_deep = ops.conv2d_dgrad(dz, P[c.name + '/kernel'], hw(inp), padding='VALID')
        ops.join_side_stream()
        return None


class ClassicISP(NIPModel):
    """The classic camera ISP as a differentiable model (pipelines.py:416-514 `_ClassicISP` / `ClassicISP`,
    models/layers.py:206-258 `DemosaicingLayer`): 1x1 CFA up-sampling (4 -> 12) -> depth_to_space(2) -> demosaicing ->
    1x1 colour conversion (camera sRGB matrix) -> pow(clip_ste(., 1/255, 1), 1/2.2).

    Demosaicing, residual=True: REFLECT-padded k x k bilinear interpolation x (frozen) minus alpha (trainable scalar,
    0.1) times a CNN f of the Bayer image: len(c_filters) x [k x k SAME conv + LeakyReLU] -> 1x1 (-> 3) + tanh; with
    c_filters=() the CNN is skipped (f = 0) and its 1x1 layer is never built.  residual=False: the CNN alone, sigmoid on
    the 1x1 head.  Both end in a straight-through clip to [0, 1].

    Parameter names: demosaicing/alpha, demosaicing/conv{i}/(kernel|bias), demosaicing/out/(kernel|bias); the constant
    tensors of the reference (up-sampling, bilinear, sRGB) are held as frozen entries up/kernel, bilinear/kernel,
    srgb/kernel whose gradients stay zero.  `brightness` never reaches `_ClassicISP` in the reference (it is not a key
    of the ParamSpec that is forwarded, pipelines.py:472-479), so there is no brightness stage here either."""

    def construct_model(self, srgb_mat=None, kernel=5, c_filters=(), cfa_pattern='gbrg', residual=True, brightness=None):
        self._h = paramspec.ParamSpec({
            'kernel': (5, int, (3, 11)),
            'c_filters': ((), tuple, paramspec.numbers_in_range(int, 1, 1024)),
            'cfa_pattern': ('gbrg', str, {'gbrg', 'rggb', 'bggr'}),
            'residual': (True, bool, None),
        })
        self._h.update(kernel=kernel, c_filters=tuple(c_filters), cfa_pattern=cfa_pattern, residual=residual)
        if self._h.kernel % 2 == 0 and self._h.residual:
            # (residual=False has no bilinear branch: its k x k layers are Keras 'same' convolutions, models/layers.py:229 - built)
            raise NotImplementedError('even demosaicing kernel {} with the bilinear residual: the reference pads (k - 1) // 2 and '
                                      'convolves VALID - its output is one pixel short of the target'.format(self._h.kernel))
        if self.in_channels != 4:
            raise ValueError('ClassicISP develops 4-plane RAW input')
        k, res = self._h.kernel, self._h.residual
        self._convs = []
        cin = 3
        for i, nf in enumerate(self._h.c_filters):
            self._convs.append(Conv2D('demosaicing/conv{}'.format(i), k, cin, int(nf), 'leaky_relu'))
            cin = int(nf)
        self._head = Conv2D('demosaicing/out', 1, cin, 3, None) if (self._convs or not res) else None
        specs = [('demosaicing/alpha', (1,))] if res else []
        for c in self._convs + ([self._head] if self._head else []):
            specs += c.specs()
        specs += [('up/kernel', (1, 1, 4, 12)), ('srgb/kernel', (1, 1, 3, 3))]
        if res:
            specs += [('bilinear/kernel', (k, k, 3, 3))]
        self._model = ParamStore(specs, self.device)
        gen = torch.Generator().manual_seed(self._seed)
        for c in self._convs + ([self._head] if self._head else []):
            c.init(self._model, gen)
        if res:
            self._model.p['demosaicing/alpha'].fill_(0.1)
            self._model.p['bilinear/kernel'].copy_(torch.from_numpy(
                np.asarray(hk.bilin_kernel(k), np.float32).reshape(k, k, 3, 3)))
        self._frozen = ('up/kernel', 'srgb/kernel', 'bilinear/kernel')
        self._h5_skip = ('up/kernel', 'srgb/kernel')     # tf constants in the reference, not Keras variables
        self.set_cfa_pattern(self._h.cfa_pattern)
        self.set_srgb_conversion(np.eye(3) if srgb_mat is None else srgb_mat)
        ps = self.patch_size
        self.y = _Placeholder((None, None if ps is None else 2 * ps, None if ps is None else 2 * ps, 3))

    @property
    def trainable_names(self):
        return [k for k in self._model.p if k not in self._frozen]

    def count_parameters(self):
        """Trainable parameters only - the constants are plain tensors in the reference, not Keras variables."""
        return int(sum(int(self._model.p[k].numel()) for k in self.trainable_names))

    def set_cfa_pattern(self, cfa_pattern):
        if cfa_pattern is not None:
            cfa_pattern = cfa_pattern.lower()
            up = hk.upsampling_kernel(cfa_pattern).reshape((1, 1, 4, 12)).astype(np.float32)
            self._model.p['up/kernel'].copy_(torch.from_numpy(up))
            self._h.update(cfa_pattern=cfa_pattern)

    def set_srgb_conversion(self, srgb_mat):
        if srgb_mat is not None:
            srgb = np.ascontiguousarray(np.asarray(srgb_mat, np.float32).T).reshape((1, 1, 3, 3))
            self._model.p['srgb/kernel'].copy_(torch.from_numpy(srgb))

    def set_camera(self, camera, cameras_json='config/cameras.json'):
        """Sets both CFA and sRGB from the camera presets (the reference reads config/cameras.json, pipelines.py:499-504)."""
        import json
        with open(cameras_json) as f:
            cameras = json.load(f)
        self.set_cfa_pattern(cameras[camera]['cfa'])
        self.set_srgb_conversion(np.array(cameras[camera]['srgb']))

    def process(self, batch_x, training=False, cfa_pattern=None, srgb_mat=None):
        self.set_cfa_pattern(cfa_pattern)
        self.set_srgb_conversion(srgb_mat)
        return super().process(batch_x, training)

    def keras_layers(self):
        """`_ClassicISP` is a subclassed tf.keras.Model whose only layer is the DemosaicingLayer (pipelines.py:416-432): its weight
        file has ONE top-level group listing the layer's variables - alpha, the convolutions, then the frozen bilinear kernel
        (trainable weights first, models/layers.py:206-233) - which is the parameter order here; so does the file written here."""
        return [('demosaicing_layer', [w for _, ws in super().keras_layers() for w in ws])]

    @property
    def model_code(self):
        return 'ClassicISP_{cfa}_{k}x{k}_{fs}-{of}{r}'.format(
            fs='-'.join(['{:d}'.format(x) for x in self._h.c_filters]), of=3, k=self._h.kernel,
            cfa=self._h.cfa_pattern, r='R' if self._h.residual else '')

    def summary(self):                                   # pipelines.py:529-533
        nf = len(self._h.c_filters)
        fs = self._h.c_filters[0] if len(set(self._h.c_filters)) == 1 else '*'
        k = self._h.kernel
        return '{}[{}] + CNN demosaicing [{}+1 layers : {k}x{k}x{} -> 1x1x3]'.format(self.class_name, self._h.cfa_pattern, nf, fs, k=k)

    def summary_compact(self):                           # pipelines.py:535-539
        nf = len(self._h.c_filters)
        fs = self._h.c_filters[0] if len(set(self._h.c_filters)) == 1 else '*'
        k = self._h.kernel
        return '{}[{}, {}+1 conv2D {k}x{k}x{} > 1x1x3]'.format(self.class_name, self._h.cfa_pattern, nf, fs, k=k)

    @classmethod
    def restore(cls, dir_name='data/models/isp/ClassicISP_auto_3x3_32-32-32-32-3R/', *, camera=None, cfa=None, srgb=None,
                patch_size=128, **kwargs):
        isp = super().restore(dir_name, patch_size=patch_size, **kwargs)
        if camera is not None:
            isp.set_camera(camera)
        isp.set_cfa_pattern(cfa)
        isp.set_srgb_conversion(srgb)
        return isp

    def forward(self, x, training=False):
        self._model.refresh_images()
        P, M = self._model.p, self._model
        t = OrderedDict()
        bayer = ops.d2s_clip(ops.conv2d(x, P['up/kernel']), 1.0, 0.0, False)
        t['a0'] = bayer
        f = None
        if self._head is not None:
            f = bayer
            for i, c in enumerate(self._convs):
                f = c.forward(M, f)
                t['a{}'.format(i + 1)] = f
            z = self._head.forward(M, f)
            f = ops.tanh(z, out=z) if self._h.residual else ops.sigmoid(z, out=z)
            t['f'] = f
        if self._h.residual:
            xb = ops.conv2d(bayer, P['bilinear/kernel'], pad_mode=ops.PAD_MODES['REFLECT'])
            rgb = ops.isp_residual(xb, f, P['demosaicing/alpha'], clip=True, out=xb)
        else:
            rgb = ops.clip01(f)
        t['srgb'] = ops.conv2d(rgb, P['srgb/kernel'])
        return ops.gamma_ste(t['srgb']), (t if training else None)

    def backward(self, t, dy):
        """dy = d loss / d y.  Only the demosaicing CNN and alpha receive gradients."""
        P, G, M = self._model.p, self._model.g, self._model
        hw = (dy.shape[1], dy.shape[2])
        if self._head is None:
            return None                                       # f = 0: d y / d alpha = 0, nothing else is trainable
        d_rgb = ops.conv2d_dgrad(ops.gamma_ste_bwd(t['srgb'], dy), P['srgb/kernel'], hw)
        if self._h.residual:
            df = ops.isp_residual_bwd(d_rgb, t['f'], P['demosaicing/alpha'], G['demosaicing/alpha'])
            dz = ops.tanh_bwd(df, t['f'], out=df)
        else:
            dz = ops.sigmoid_bwd(d_rgb, t['f'], out=d_rgb)
        n = len(self._convs)
        self._head.backward_params(M, t['a{}'.format(n)], dz)
        layer = self._head
        for i in range(n, 0, -1):
            dz = layer.backward_input(M, dz, hw, act_mask=t['a{}'.format(i)])
            layer = self._convs[i - 1]
            layer.backward_params(M, t['a{}'.format(i - 1)], dz)
        ops.join_side_stream()
        return None


class ONet(NIPModel):
    """Dummy pipeline for RGB training (pipelines.py:353-362): identity, no parameters."""

    def __init__(self, loss_metric='L2', patch_size=None, in_channels=3, device=None, **kwargs):
        super().__init__(loss_metric=loss_metric, patch_size=patch_size, in_channels=in_channels, device=device)

    def construct_model(self, **kwargs):
        self._model = ParamStore([], self.device)
        ps = self.patch_size
        self.y = _Placeholder((None, ps, ps, 3))

    def forward(self, x, training=False):
        return x, ({} if training else None)

    def backward(self, ctx, dy):
        ops.join_side_stream()
        return None

    @property
    def model_code(self):
        return self.class_name


supported_models = ['UNet', 'INet', 'DNet', 'ONet', 'ClassicISP']
