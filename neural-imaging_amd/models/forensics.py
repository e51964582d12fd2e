"""
FAN - forensic analysis network on the HIP kernels.  Mirrors the reference's models/forensics.py:29-133.

Graph as built by the workflow's defaults (forensics.py:62-90; ctor values override the ParamSpec defaults, SURVEY 8a
quirk 6): ConstrainedConv2D -> 4 x [Conv5x5 SAME (32,64,128,256) + LeakyReLU(0.2) -> MaxPool2] -> Conv1x1 256 + LReLU
-> GAP -> Dense(n_classes, softmax); loss = SparseCategoricalCrossentropy on probabilities (forensics.py:94).
Also built: use_gap=False (Flatten) and n_dense > 0 hidden Dense + LeakyReLU layers (forensics.py:79-87), as 1x1
convolutions on (N,1,1,F) tensors, with Keras Dropout after each of them at training time (masks from a device generator).
"""
import os
from collections import OrderedDict

import numpy as np

import torch

from .. import ops, parallel
from ..device import DeviceArray, to_device
from ..helpers import paramspec
from .layers import ConstrainedConv2D, Conv2D
from .tfmodel import ParamStore, TFModel, glorot_uniform_


# The FAN's weight gradients issued behind its input-gradient chain when the caller keeps launching (the workflow's codec /
# manipulation / UNet backward): NIMG_NO_LATE_PARAMS=1 = beside the input gradients, as rounds 1 - 3 (A/B runs).
LATE_PARAMS = os.environ.get('NIMG_NO_LATE_PARAMS') is None
LATE_MIN_IMAGES = int(os.environ.get('NIMG_LATE_MIN_IMAGES', '160'))      # FAN batch from which the late order pays
DEFER_MASK = int(os.environ.get('NIMG_PIPELINE_FAN_MASK', '255'))     # pipelined FAN update: the layers whose weight gradients wait for the next step
LATE_MASK = int(os.environ.get('NIMG_LATE_MASK', '255'))      # bit 0: constrained filter, bit i: convolution i (A/B runs)


class FAN(TFModel):

    def __init__(self, n_classes, patch_size=None, n_filters=32, n_fscale=2, n_convolutions=4, kernel=5, dropout=0.0,
                 use_gap=True, n_dense=0, activation='leaky_relu', device=None, seed=4321):
        super().__init__(device=device)
        self._h = paramspec.ParamSpec({
            'n_classes': (7, int, (2, 256)),
            'n_filters': (32, int, (4, 128)),
            'n_fscale': (2, float, (0.25, 4)),
            'n_convolutions': (4, int, (1, 32)),
            'kernel': (5, int, (3, 11)),
            'dropout': (0, float, (0, 1)),
            'use_gap': (False, bool, None),
            'n_dense': (2, int, (0, 16)),
            'activation': ('leaky_relu', str, set(ops.ACTIVATIONS)),          # helpers/tf_helpers.py:22-28 (forensics.py:55)
        })
        params = locals()
        self._h.update(**{k: params[k] for k in self._h.keys()})
        if not 0 <= float(dropout) < 1:
            raise ValueError('dropout rate must be in [0, 1)')
        # Dropout (forensics.py:88) follows every hidden Dense layer at training time.  The masks come from a device
        # generator (seeded per model; TensorFlow's own random stream is not reproducible elsewhere) unless a test injects
        # them through `dropout_masks` (list of uint8 tensors, one per hidden layer, consumed by the next training forward).
        self._dropout_gen, self._dropout_seed = None, int(seed) + 17
        self.dropout_masks = None
        if not use_gap and patch_size is None:
            raise ValueError('the Flatten head (use_gap=False) needs a fixed patch_size')
        # the reference's range is any integer 3 .. 11 (forensics.py:51): 3 and 5 have the throughput-mode kernels, the others run on
        # the generic float32 ones; an even kernel's SAME padding is TF's (the extra row / column after the image)
        self.patch_size = patch_size
        self.x = _Shape((None, patch_size, patch_size, 3))
        self.y = _Shape((None, n_classes))

        self._constrained = ConstrainedConv2D('constrained')
        self._convs = []
        cin, nf = 3, n_filters
        for i in range(self._h.n_convolutions):
            self._convs.append(Conv2D('conv{}'.format(i + 1), self._h.kernel, cin, nf, self._h.activation,
                                      mask_activation=self._h.activation))
            cin, nf = nf, int(nf * self._h.n_fscale)
        nf = int(nf // self._h.n_fscale)
        self._conv1x1 = Conv2D('conv1x1', 1, cin, nf, self._h.activation, mask_activation=self._h.activation)
        # an activation the kernels do not fuse (relu, tanh, sigmoid, softsign): every layer runs unfused on float32 tensors -
        # convolution, element-wise activation, MaxPool2D - and the backward pass takes the derivative from the stored outputs
        self._generic_act = self._h.activation != 'leaky_relu'
        self._use_gap = bool(use_gap)
        feat = nf if use_gap else nf * (patch_size // 2 ** self._h.n_convolutions) ** 2
        # hidden Dense + LeakyReLU layers (Keras names dense, dense_1, ...; the classifier is the last Dense)
        self._hidden = []
        for i in range(self._h.n_dense):
            nf = int(nf // self._h.n_fscale)
            self._hidden.append(Conv2D('dense' if i == 0 else 'dense_{}'.format(i), 1, feat, nf, self._h.activation))
            feat = nf
        self._cls = 'dense' if self._h.n_dense == 0 else 'dense_{}'.format(self._h.n_dense)
        self._n_features = feat
        specs = self._constrained.specs()
        for c in self._convs:
            specs += c.specs()
        specs += self._conv1x1.specs()
        for d in self._hidden:                 # Dense kernel (in, out) == 1x1 HWIO kernel in memory
            specs += [(d.name + '/kernel', (d.cin, d.cout)), (d.name + '/bias', (d.cout,))]
        specs += [(self._cls + '/kernel', (feat, n_classes)), (self._cls + '/bias', (n_classes,))]
        self._model = ParamStore(specs, self.device)
        gen = torch.Generator().manual_seed(seed)
        self._constrained.init(self._model)
        for c in self._convs:
            c.init(self._model, gen)
        self._conv1x1.init(self._model, gen)
        for d in self._hidden + [None]:
            name, fi, fo = (d.name, d.cin, d.cout) if d is not None else (self._cls, feat, n_classes)
            k = torch.empty((fi, fo), dtype=torch.float32)
            glorot_uniform_(k, fi, fo, gen)
            self._model.p[name + '/kernel'].copy_(k)
            self._model.p[name + '/bias'].zero_()
        self.learning_rate = 1e-3
        self.loss = self._loss

    def reset_performance_stats(self):
        self.performance = {'loss': {'training': [], 'validation': []}, 'accuracy': {'validation': []},
                            'confusion': []}

    # ------------------------------------------------------------------------------------------------------------
    def forward(self, x, labels=None, training=False, loss_scale=None):
        """x (N,H,W,3) device tensor. labels: int32 device tensor or None. Returns (probs, ctx)."""
        self._model.refresh_images()
        P = self._model
        t = OrderedDict()
        t['x'] = x
        c1 = self._convs[0]
        front = ops.front_end_ok(c1.cin, c1.cout, c1.ks, x.shape[1], x.shape[2], n=x.shape[0] if training else None) and \
            c1.activation == 'leaky_relu'
        # throughput mode: the filtered image leaves as 8-byte bf16 {c, 1} pixels, the form the row-band conv1 kernels read
        net, nf = self._constrained.forward(P, x, c4_only=front)
        t['constrained'], t['nf'] = net, nf
        for i, c in enumerate(self._convs):
            if i == 0 and front:
                nxt = self._convs[1] if len(self._convs) > 1 else self._conv1x1
                as_bf16 = ops.STORE_BF16 and nxt.cout >= 8
                net, idx = ops.conv1_pool_c4(net, P.p[c.name + '/kernel'], P.p[c.name + '/bias'], want_idx=training,
                                             out_bf16=as_bf16)
                t['idx1'], t['front'] = idx, True
            elif c.can_pool(net):        # conv + LeakyReLU + pool in one pass; the full-resolution tensor is not stored
                # throughput mode: pooled activations live in HBM as bf16 - every consumer (next convolution, its weight
                # gradient, the LeakyReLU' sign test) rounds to bf16 / reads the sign anyway, so no result bit changes
                nxt = self._convs[i + 1] if i + 1 < len(self._convs) else self._conv1x1
                as_bf16 = ops.COMPUTE == 'bf16' and ops.STORE_BF16 and c.cout % 8 == 0 and nxt.cout >= 8
                net, idx = c.forward_pool(P, net, want_idx=training, out_bf16=as_bf16)
                t['idx{}'.format(i + 1)] = idx
            else:
                a = c.forward(P, net)
                t['conv{}'.format(i + 1)] = a
                net = ops.maxpool2(a)
            t['pool{}'.format(i + 1)] = net
        n = x.shape[0]
        ls = (1.0 / n) if loss_scale is None else loss_scale
        c5 = self._conv1x1
        if self._use_gap and not self._hidden and not self._generic_act and c5.cin == c5.cout and ops.head_fused_ok(net, c5.cout):
            # throughput mode: 1x1 conv + LeakyReLU + global average pooling in ONE pass (csrc/head.hip); the activation leaves as
            # one sign bit per value, all the backward pass needs of it
            gap, mask, mask_p = ops.head_fwd(net, P.p[c5.name + '/kernel'], P.p[c5.name + '/bias'], want_mask=training)
            t['head_mask_p'] = mask_p
            probs, loss_per, dlogits = ops.fan_dense_fwd(gap, P.p[self._cls + '/kernel'], P.p[self._cls + '/bias'], labels, ls)
            t['head_mask'], t['gap'], t['probs'], t['loss_per'], t['dlogits'], t['loss_scale'] = mask, gap, probs, loss_per, dlogits, ls
            return probs, (t if training else None)
        a = self._conv1x1.forward(P, net)
        t['conv1x1'] = a
        head_in = a
        if self._hidden or not self._use_gap:
            # general head: features as an (N,1,1,F) tensor, hidden Dense layers as 1x1 convolutions
            if self._use_gap:
                if a.shape[1] != a.shape[2]:
                    raise ValueError('the hidden-Dense head pools square feature maps')
                head_in = ops.avgpool(a, a.shape[1])
            else:
                head_in = a.reshape(n, 1, 1, -1)
            t['feat'] = head_in
            rate = float(self._h.dropout)
            for li, d in enumerate(self._hidden):
                w4 = P.p[d.name + '/kernel'].view(1, 1, d.cin, d.cout)
                if self._generic_act:
                    head_in = ops.conv2d(head_in, w4, P.p[d.name + '/bias'])
                    ops.activation(head_in, self._h.activation, out=head_in)
                else:
                    head_in = ops.conv2d(head_in, w4, P.p[d.name + '/bias'], act='leaky_relu')
                t[d.name] = head_in                              # the activation (its sign gates LeakyReLU' backwards)
                if training and rate > 0:
                    if self.dropout_masks is not None:
                        keep = self.dropout_masks[li].to(device=x.device, dtype=torch.uint8).reshape(head_in.shape).contiguous()
                    else:
                        if self._dropout_gen is None:            # per rank: the shards of a global batch get different masks
                            self._dropout_gen = parallel.rank_generator(self._dropout_seed, x.device)
                        keep = (torch.rand(head_in.shape, device=x.device, generator=self._dropout_gen) >= rate).to(torch.uint8)
                    head_in = ops.mask_scale(head_in, keep, 1.0 / (1.0 - rate))
                    t[d.name + '/keep'], t[d.name + '/dropped'] = keep, head_in
            if training:
                self.dropout_masks = None
        gap, probs, loss_per, dlogits = ops.fan_head_fwd(head_in, P.p[self._cls + '/kernel'], P.p[self._cls + '/bias'],
                                                         labels, ls)
        t['head_in'] = head_in
        t['gap'], t['probs'], t['loss_per'], t['dlogits'], t['loss_scale'] = gap, probs, loss_per, dlogits, ls
        return probs, (t if training else None)

    def backward(self, t, need_input_grad=False, join=True, defer=None):
        """Fills the gradient buffer; returns (loss[1], d loss / d x or None).
        join=False: the caller joins the side streams itself (ops.join_side_stream / ops.nan_flag / the Adam step) - with
        LATE_PARAMS the weight gradients of the fused conv + pool layers are ISSUED behind the whole input-gradient chain, so
        they run beside whatever the caller launches next instead of beside the (equally chip-filling) input-gradient kernels.
        defer (a list, with join=False): those launches are not issued at all but appended to the list as closures - the caller
        runs them later (the workflow's pipelined FAN update issues them beside the NEXT step's UNet forward)."""
        late = [] if ((LATE_PARAMS or defer is not None) and not join) else None
        self._defer = defer if late is not None else None
        def params(fn, layer=0):           # a parameter-gradient launch: now, or behind the input-gradient chain
            if late is None or not (LATE_MASK >> layer) & 1:
                fn()
            elif self._defer is not None and (DEFER_MASK >> layer) & 1:
                self._defer.append(fn)     # ... or not in this step at all (the workflow's pipelined FAN update)
            else:
                late.append(fn)
        P = self._model
        hw = lambda a: (a.shape[1], a.shape[2])
        nconv = len(self._convs)
        fused = lambda i: i >= 1 and t.get('idx{}'.format(i)) is not None
        pooled_path = lambda i: fused(i) and not fused(i - 1) and ops.pooled_backward_ok(
            self._convs[i - 1].cin, self._convs[i - 1].cout, self._convs[i - 1].ks)
        # a pooled gradient that is only un-pooled into a bf16 tensor can itself be stored as bf16 (throughput mode)
        g_bf16 = lambda i: (ops.COMPUTE == 'bf16' and ops.STORE_BF16 and fused(i) and self._convs[i - 1].cout % 8 == 0 and
                            (pooled_path(i) or self._convs[i - 1].cin % 8 == 0))
        if 'head_mask' in t:
            d_pool, loss = self._fused_head_backward(t)
        else:
            d_pool, loss = self._head_backward(t, fused, g_bf16)
        return self._conv_backward(t, d_pool, loss, fused, g_bf16, params, late, need_input_grad, join)

    def _fused_head_backward(self, t):
        """Backward of the fused head (forward above): classifier gradients, then the gradient at the INPUT of the 1x1 layer straight
        from dlogits and the activation's sign bits; the 1x1 layer's weight gradient on the side stream from the same bits."""
        P, c5 = self._model, self._conv1x1
        pool = t['pool{}'.format(len(self._convs))]
        wd = P.p[self._cls + '/kernel']
        loss = ops.fan_dense_bwd(t['gap'], t['dlogits'], t['loss_per'], t['loss_scale'], P.g[self._cls + '/kernel'],
                                 P.g[self._cls + '/bias'])
        dw = P.g[c5.name + '/kernel']
        with ops.side_stream(pool, t['head_mask_p'], t['dlogits'], wd, key=dw.data_ptr()):
            ops.head_wgrad(pool, t['head_mask_p'], t['dlogits'], wd, dw.view(dw.shape[2], dw.shape[3]), P.g[c5.name + '/bias'])
        d_pool = ops.head_dgrad(t['head_mask'], t['dlogits'], wd, P.p[c5.name + '/kernel'],
                                pool if t.get('idx{}'.format(len(self._convs))) is not None else None, pool.shape)
        return d_pool, loss

    def _head_backward(self, t, fused, g_bf16):
        P = self._model
        hw = lambda a: (a.shape[1], a.shape[2])
        a = t['conv1x1']
        # classifier backward; dz = gradient w.r.t. the PRE-activation of whatever fed the head (its LeakyReLU' applied)
        gen = self._generic_act
        act_bwd = (lambda g, y: ops.activation_bwd(g, y, self._h.activation, out=g)) if gen else ops.lrelu_bwd
        dz, loss = ops.fan_head_bwd(t['head_in'], t['gap'], P.p[self._cls + '/kernel'], t['dlogits'], t['loss_per'],
                                    t['loss_scale'], P.g[self._cls + '/kernel'], P.g[self._cls + '/bias'],
                                    alpha=1.0 if gen else None)       # slope 1 = no derivative applied: done explicitly below
        if gen and not ('feat' in t and self._hidden):
            # the head was fed an activation (conv1x1's, directly or through Flatten): its derivative from the stored output
            dz = act_bwd(dz.reshape(a.shape), a)
        if 'feat' in t:
            keep_scale = 1.0 / (1.0 - float(self._h.dropout))
            for i in range(len(self._hidden) - 1, -1, -1):
                d = self._hidden[i]
                if d.name + '/keep' in t:          # dz arrived w.r.t. the dropped tensor (x LeakyReLU' of its sign = the
                    dz = ops.mask_scale(dz.reshape(t[d.name].shape), t[d.name + '/keep'], keep_scale)   # activation's)
                if gen:                            # derivative of this hidden layer's activation (the head / the layer above
                    dz = act_bwd(dz.reshape(t[d.name].shape), t[d.name])                      # handed the plain gradient)
                prev = self._hidden[i - 1].name if i > 0 else None
                inp = t.get(prev + '/dropped', t[prev]) if i > 0 else t['feat']
                ops.conv2d_wgrad(inp, dz, 1, dw=P.g[d.name + '/kernel'].view(1, 1, d.cin, d.cout),
                                 db=P.g[d.name + '/bias'])
                w4 = P.p[d.name + '/kernel'].view(1, 1, d.cin, d.cout)
                # hidden activations carry a LeakyReLU; the feature vector itself (GAP / Flatten output) does not
                dz = ops.conv2d_dgrad(dz, w4, (1, 1), act_mask=t[prev] if (i > 0 and not gen) else None)
            if self._hidden:
                # dz is now d loss / d feat (no activation applied yet): route it back into the 1x1-conv activation
                if self._use_gap:
                    dz = act_bwd(ops.avgpool_bwd(dz, a.shape[1]), a)
                else:
                    dz = act_bwd(dz.reshape(a.shape).contiguous(), a)
            else:
                dz = dz.reshape(a.shape)          # Flatten straight into the classifier: the head applied LReLU'(a)
        nconv = len(self._convs)
        pool = t['pool{}'.format(nconv)]
        self._conv1x1.backward_params(P, pool, dz)
        # fused layers: the producer of d_pool applies LeakyReLU'(pooled) in its epilogue (sign(window max) = sign(pooled))
        d_pool = self._conv1x1.backward_input(P, dz, hw(pool), act_mask=pool if fused(nconv) else None,
                                              out_bf16=g_bf16(nconv))
        return d_pool, loss

    def _conv_backward(self, t, d_pool, loss, fused, g_bf16, params, late, need_input_grad, join):
        P = self._model
        hw = lambda a: (a.shape[1], a.shape[2])
        gen = self._generic_act
        act_bwd = (lambda g, y: ops.activation_bwd(g, y, self._h.activation, out=g)) if gen else ops.lrelu_bwd
        nconv = len(self._convs)
        for i in range(nconv, 0, -1):
            conv = self._convs[i - 1]
            inp = t['pool{}'.format(i - 1)] if i > 1 else t['constrained']
            prev_mask = inp if fused(i - 1) else None
            if i == 1 and t.get('front'):
                # row-band front end: `inp` is the filtered image as 8-byte bf16 pixels, idx1 the 2-bit arg-max codes only these
                # two kernels read (ops.conv1_pool_c4)
                params(lambda inp=inp, g=d_pool, i=i, conv=conv: ops.conv1_wgrad_c4(
                    inp, g, t['idx{}'.format(i)], dw=P.g[conv.name + '/kernel'], db=P.g[conv.name + '/bias'], side=True), i)
                d_pool = ops.conv1_dgrad_pooled(d_pool, t['idx{}'.format(i)], P.p[conv.name + '/kernel'])
                continue
            if fused(i) and ops.pooled_backward_ok(conv.cin, conv.cout, conv.ks) and prev_mask is None:
                # the pooled gradient feeds the weight / input gradient kernels directly (un-pooled while staging)
                params(lambda inp=inp, g=d_pool, i=i, conv=conv: ops.conv2d_wgrad_pooled(
                    inp, g, t['idx{}'.format(i)], conv.ks, dw=P.g[conv.name + '/kernel'], db=P.g[conv.name + '/bias'],
                    side=True), i)
                d_pool = ops.conv2d_dgrad_pooled(d_pool, t['idx{}'.format(i)], P.p[conv.name + '/kernel'])
                continue
            if fused(i) and ops.unpool_fold_ok(inp, d_pool, conv.cin, conv.cout, conv.ks):
                # throughput mode, 5x5 layers: both gradient kernels read (pooled gradient, arg-max bytes) and route while
                # staging - the full-resolution gradient (4x the bytes, 3/4 zeros) is never written nor re-read
                params(lambda inp=inp, g=d_pool, i=i, conv=conv: ops.conv2d_wgrad_unpool(
                    inp, g, t['idx{}'.format(i)], conv.ks, dw=P.g[conv.name + '/kernel'], db=P.g[conv.name + '/bias'], side=True), i)
                d_pool = ops.conv2d_dgrad_unpool(d_pool, t['idx{}'.format(i)], P.p[conv.name + '/kernel'], act_mask=prev_mask,
                                                 out_bf16=g_bf16(i - 1))
                continue
            if fused(i):
                # throughput mode: the un-pooled gradient only feeds bf16 MFMA kernels - store it as bf16 (same bits)
                as_bf16 = ops.COMPUTE == 'bf16' and ops.STORE_BF16 and conv.cout % 8 == 0 and conv.cin % 8 == 0
                dz = ops.maxpool2_unpool(d_pool, t['idx{}'.format(i)], None, apply_mask=False, out_bf16=as_bf16)
            else:
                dz = ops.maxpool2_bwd(d_pool, t['conv{}'.format(i)], None, apply_mask=not gen)
                if gen:
                    dz = act_bwd(dz, t['conv{}'.format(i)])
            params(lambda conv=conv, inp=inp, dz=dz: conv.backward_params(P, inp, dz), i)
            d_pool = conv.backward_input(P, dz, hw(inp), act_mask=prev_mask, out_bf16=g_bf16(i - 1))
        params(lambda g=d_pool: self._constrained.backward_params(P, t['x'], g))
        dx = self._constrained.backward_input(t['nf'], d_pool) if need_input_grad else None
        self._defer = None
        with ops.one_fork():               # the late launches depend on nothing queued after this point: one marker for all
            for fn in late or ():
                fn()
        if join:
            ops.join_side_stream()
            P.grads_pending = False
        else:
            # the weight gradients are still in flight on the side streams: whoever reads the gradient buffer next joins first.
            # ParamStore.adam and parallel.GradientBucket.launch call ops.join_side_stream() unconditionally (a no-op once the
            # side streams are clean: ops keeps the set of dirty streams); a test or tool that reads P.flat_grad directly after
            # backward(join=False) must do the same.  The attribute is informational (cleared by adam and backward(join=True)).
            P.grads_pending = True
        return loss, dx

    # -- reference surface ---------------------------------------------------------------------------------------
    def _loss(self, labels, probs):
        """SparseCategoricalCrossentropy()(labels, probabilities) (forensics.py:94), host-side convenience."""
        p = np.clip(np.asarray(probs, np.float64), 1e-7, 1 - 1e-7)
        lab = np.asarray(labels).astype(np.int64)
        return float(np.mean(np.log(p.sum(axis=1)) - np.log(p[np.arange(len(lab)), lab])))

    def process(self, batch_x, training=False):
        """Class probabilities for an image batch (NHWC rgb)."""
        return DeviceArray(self.forward(to_device(batch_x, self.device))[0])

    def process_and_decide(self, batch_x, with_confidence=False):
        probs = self.process(batch_x).numpy()
        if with_confidence:
            return probs.argmax(axis=1), probs.max(axis=1)
        return probs.argmax(axis=1)

    def training_step(self, batch_x, target_labels, learning_rate=None):
        """One optimisation step (forensics.py:116-125); returns the loss."""
        x = to_device(batch_x, self.device)
        labels = torch.as_tensor(np.asarray(target_labels), dtype=torch.int32).to(self.device)
        _, ctx = self.forward(x, labels, training=True)
        loss, _ = self.backward(ctx, need_input_grad=False)
        world = parallel.sync_gradients(self._model.flat_grad)       # data parallel: CE is a mean over the batch
        if learning_rate is not None:
            self.learning_rate = learning_rate
        self._model.adam(self.learning_rate, grad_scale=1.0 / world)
        return DeviceArray(loss)

    def summary(self):
        return '{kernel}x{kernel} CNN: 1+{conv}+1 conv layers {gap}+ {fc} fc layers [{params:,} parameters]'.format(
            kernel=self._h.kernel, conv=self._h.n_convolutions, fc=self._h.n_dense,
            gap='+ (GAP) ' if self._h.use_gap else '', params=self.count_parameters())


class _Shape(object):
    def __init__(self, shape):
        self.shape = tuple(shape)
