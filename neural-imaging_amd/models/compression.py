"""
Learned image compression on the HIP kernels.  Mirrors the reference's models/compression.py: DCN (:28-184) - the
abstract codec with a soft-codebook latent, entropy-regularised L2 loss and Adam - and TwitterDCN (:187-291).

TwitterDCN graph (compression.py:219-271): 2(x-0.5) -> Conv5x5 s2 64 +LReLU -> Conv5x5 s2 128 -> 3 residual blocks
(block 1 is fed LeakyReLU(net) but its skip adds the PRE-activation net, :224-227) -> Conv5x5 s2 n_features ->
DiscreteLatent -> Conv3x3 512 -> d2s -> 3 residual blocks -> Conv3x3 256 +LReLU -> d2s -> Conv3x3 12 -> d2s -> (x+1)/2
-> straight-through clip.  Strided convolutions use TF's asymmetric SAME padding.

As in the reference the latent scale is always trainable and the codebook never, whatever
scale_latent / train_codebook say; they are recorded hyper-parameters only.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from .. import ops, parallel
from ..device import DeviceArray, to_device
from ..helpers import paramspec
from .layers import Conv2D, Conv5x5Stride2Image
from .tfmodel import ParamStore, TFModel


# A/B switches, read once at import (not per call in the hot path)
_NO_BF16_COPY = bool(os.environ.get('NIMG_NO_BF16_COPY'))
_NO_S2D_CHAIN = bool(os.environ.get('NIMG_NO_S2D_CHAIN'))

CODEC_FORK_GROUPS = os.environ.get('NIMG_CODEC_FORK_GROUPS') is not None

class _Shape(object):
    def __init__(self, shape):
        self.shape = tuple(shape)


class _DiscreteLatent(object):
    """Surface callers touch: discrete_latent.scaling_factor, .quantization.codebook (layers.py:183-203)."""

    def __init__(self, model):
        self._model = model

    @property
    def scaling_factor(self):
        return DeviceArray(self._model._model.p['latent_scaling'])


class DCN(TFModel):

    def __init__(self, patch_size=128, latent_bpf=5, rounding='soft-codebook', train_codebook=False,
                 entropy_weight=250, scale_latent=True, use_batchnorm=False, loss_metric='L2', device=None,
                 seed=777, **kwargs):
        super().__init__(device=device)
        self._h = paramspec.ParamSpec({
            'latent_bpf': (5, int, (1, 8)),
            'train_codebook': (False, bool, None),
            'entropy_weight': (250, float, (0, 1e6)),
            'scale_latent': (True, bool, None),
            'use_batchnorm': (False, bool, None),
            'loss_metric': ('L2', str, {'L2'}),
            'rounding': ('soft', str, {'identity', 'soft', 'soft-codebook', 'sin'}),
        })
        params = locals()
        self._h.update(**{k: params[k] for k in self._h.keys()})
        self.patch_size = patch_size
        self._seed = seed
        self.x = _Shape((None, patch_size, patch_size, 3))
        qmin, qmax = -2 ** (self._h.latent_bpf - 1) + 1, 2 ** (self._h.latent_bpf - 1)      # layers.py:110-116
        self._codebook = torch.arange(qmin, qmax + 1, dtype=torch.float32).to(self.device)
        self.discrete_latent = _DiscreteLatent(self)
        self.construct_model(**kwargs)
        self._has_attributes(['y', '_model'])
        if loss_metric != 'L2':
            raise NotImplementedError('Loss metric {} not supported yet.'.format(loss_metric))
        self.loss = self._loss
        self.learning_rate = 1e-3
        self._lws = None

    def _loss(self, image_target, image_compressed, entropy):
        """tf.nn.l2_loss(target - compressed) + entropy_weight * entropy  (compression.py:92-93)"""
        a, b = to_device(image_target, self.device), to_device(image_compressed, self.device)
        return float(DeviceArray(ops.l2_loss(a, b)[0])) + self._h.entropy_weight * float(entropy)

    def construct_model(self, **kwargs):
        raise NotImplementedError('Not implemented!')

    def reset_performance_stats(self):
        self.performance = self._reset_performance(['loss', 'entropy', 'ssim', 'psnr'])

    def get_codebook(self):
        return self._codebook.cpu().numpy().reshape((-1,))

    # -- reference surface ---------------------------------------------------------------------------------------
    def compress(self, batch_x):
        x = to_device(batch_x, self.device)
        x = x.unsqueeze(0).contiguous() if x.dim() == 3 else x
        return DeviceArray(self.encode(x)[0])

    def decompress(self, batch_z):
        z = to_device(batch_z, self.device)
        z = z.unsqueeze(0).contiguous() if z.dim() == 3 else z
        return DeviceArray(self.decode(z)[0])

    def process(self, batch_x, return_entropy=False):
        y, ent, _ = self.forward(to_device(batch_x, self.device))
        return (DeviceArray(y), DeviceArray(ent)) if return_entropy else DeviceArray(y)

    def training_step(self, batch_x, learning_rate=None, sync=True):
        """One optimisation step on l2_loss(x - y) + entropy_weight * H (compression.py:123-138).  sync=False keeps the step
        asynchronous: 'loss' is then a lazy value that reads the device scalars only when converted (float / numpy)."""
        x = to_device(batch_x, self.device)
        y, ent, ctx = self.forward(x, training=True)
        l2, dy = ops.l2_loss(x, y, grad_scale=1.0)
        self.backward(ctx, dy, entropy_coef=self._h.entropy_weight)
        parallel.sync_gradients(self._model.flat_grad)
        if learning_rate is not None:
            self.learning_rate = learning_rate
        self._model.adam(self.learning_rate)          # l2_loss is a SUM over the batch: summed gradients, no 1/world
        h, w = x.shape[1], x.shape[2]
        ssim = DeviceArray(ops.ssim(x, y, mode='tf').mean()) if min(h, w) >= 11 else np.nan     # compression.py:89,129
        if not sync:
            return {'loss': _LazyDcnLoss(l2, ent, self._h.entropy_weight), 'ssim': ssim, 'entropy': DeviceArray(ent)}
        loss = float(DeviceArray(l2)) + self._h.entropy_weight * float(DeviceArray(ent))
        return {'loss': np.sqrt(2 * loss), 'ssim': ssim, 'entropy': DeviceArray(ent)}

    def compression_stats(self, patch_size=None, n_latent_bytes=None):
        n_latent_bytes = n_latent_bytes or self._h.latent_bpf / 8
        ps = patch_size or self.patch_size
        if ps is None:
            raise ValueError('Patch size not specified!')
        n_latent = int(np.prod((ps // 8, ps // 8, self._h.n_features)))
        bitmap_size = ps * ps * 3
        return {'rate': bitmap_size / (n_latent_bytes * n_latent), 'bpp': 8 * n_latent * n_latent_bytes / (ps * ps),
                'bpf': 8 * n_latent_bytes, 'bytes': n_latent * n_latent_bytes}

    def summary(self):
        l_shape = 'x'.join(str(x) for x in self.latent_shape if x is not None)
        return '{} : {}-D latent space @ {}-bpf [{:,.0f} params]'.format(self.class_name, l_shape, self._h.latent_bpf,
                                                                         self.count_parameters())

    def summary_compact(self):
        return '{} {}-D'.format(self.class_name, self.latent_shape[-1])

    def keras_layers(self):
        """The reference saves `codec` = Model(x -> [decoder(encoder(x)), entropy]) (compression.py:274-279): its weight file has
        TWO top-level layers, the nested Models 'encoder' (its convolutions + the DiscreteLatent scaling factor) and 'decoder',
        each listing its variables as '<layer>/<variable>:0' - so does the file written here (Keras loads by layer order and
        checks the layer count; a flat list of 19 layers would be refused)."""
        groups = [('encoder', []), ('decoder', [])]
        which = 0
        for lname, ws in super().keras_layers():
            groups[which][1].extend(ws)
            if lname == 'latent_scaling':
                which = 1
        return groups

    @property
    def model_code(self):
        return '{}-{}C'.format(type(self).__name__, self._h.n_features)


class _LazyDcnLoss(DeviceArray):
    """sqrt(2 (l2 + w H)) (compression.py:135), evaluated on the host only when somebody reads it."""
    __slots__ = ('l2', 'ent', 'w')

    def __init__(self, l2, ent, w):
        self.t = l2
        self.l2, self.ent, self.w = l2, ent, float(w)

    def numpy(self):
        v = float(self.l2.detach().cpu().reshape(())) + self.w * float(self.ent.detach().cpu().reshape(()))
        return np.asarray(np.sqrt(2 * v), dtype=np.float64)

    def __float__(self):
        return float(self.numpy())


class TwitterDCN(DCN):

    def construct_model(self, n_features=32, activation='leaky_relu'):
        self._h.add({'n_features': (32, int, (4, 128)),
                     'activation': ('leaky_relu', str, set(ops.ACTIVATIONS))})        # helpers/tf_helpers.py:22-28 (compression.py:202)
        self._h.update(n_features=n_features, activation=activation)
        nf = self._h.n_features
        act = self._h.activation
        if self.patch_size is None:
            self.latent_shape, self.n_latent = (None, None, nf), None
        else:
            self.latent_shape = (self.patch_size // 8, self.patch_size // 8, nf)
            self.n_latent = int(np.prod(self.latent_shape))
        L = OrderedDict()
        L['e1'] = Conv5x5Stride2Image('e1', 5, 3, 64, act, stride=2)
        L['e2'] = Conv2D('e2', 5, 64, 128, None, stride=2, mask_activation=act)
        for b in (1, 2, 3):
            L['er{}a'.format(b)] = Conv2D('er{}a'.format(b), 3, 128, 128, act)
            L['er{}b'.format(b)] = Conv2D('er{}b'.format(b), 3, 128, 128, None, mask_activation=act)
        L['elat'] = Conv2D('elat', 5, 128, nf, None, stride=2)
        L['d512'] = Conv2D('d512', 3, nf, 512, None)
        for b in (1, 2, 3):
            L['dr{}a'.format(b)] = Conv2D('dr{}a'.format(b), 3, 128, 128, act)
            L['dr{}b'.format(b)] = Conv2D('dr{}b'.format(b), 3, 128, 128, None, mask_activation=act)
        L['d256'] = Conv2D('d256', 3, 128, 256, act)
        L['d12'] = Conv2D('d12', 3, 64, 12, None, mask_activation=act)
        self._layers = L
        specs = []
        for name in ('e1', 'e2', 'er1a', 'er1b', 'er2a', 'er2b', 'er3a', 'er3b', 'elat'):
            specs += L[name].specs()
        specs += [('latent_scaling', ())]
        for name in ('d512', 'dr1a', 'dr1b', 'dr2a', 'dr2b', 'dr3a', 'dr3b', 'd256', 'd12'):
            specs += L[name].specs()
        self._model = ParamStore(specs, self.device)
        gen = torch.Generator().manual_seed(self._seed)
        for l in L.values():
            l.init(self._model, gen)
        self._model.p['latent_scaling'].fill_(1.0)
        ps = self.patch_size
        self.y = _Shape((None, ps, ps, 3))

    @property
    def model_code(self):
        s = [self._h.rounding, 'Q+{}bpf'.format(self._h.latent_bpf) if self._h.train_codebook else
             'Q-{}bpf'.format(self._h.latent_bpf), 'S+' if self._h.scale_latent else 'S-']
        if self._h.entropy_weight is not None:
            s.append('H+{:.2f}'.format(self._h.entropy_weight))
        return '{}/{}'.format(super().model_code, '_'.join(s))

    # ------------------------------------------------------------------------------------------------------------
    def _bf16_inner(self):
        """Throughput mode: the tensors INSIDE a residual block - the activation between its two convolutions and that
        activation's gradient - live in HBM as bf16.  Their only consumers are convolution / weight-gradient operands (rounded to
        bf16 on the way to the matrix core anyway) and the LeakyReLU' sign test: the forward pass is bit-identical, the weight
        gradients agree to summation order; the fused BIAS gradient of a block's first layer sums the stored (now bf16-rounded)
        gradient - an unbiased 2^-9 relative rounding per element (test_dcn_bf16_storage_inside_residual_blocks_is_bit_neutral).
        The residual stream itself (a running float32 sum) stays float32; the kernels that write it also write a bf16 COPY
        (ops.conv2d bf16_copy) - what its consumers (the next block's first convolution, the weight gradients, the input
        gradient of the block's second layer) would round it to on the way to the matrix core - so that all 3x3 work of the
        blocks reads bf16 operands through the bf16-input kernels (weight gradient: the all-taps kernel) while the skip sums
        read and write the exact float32 tensor."""
        # the A/B switches that send the strided layers / depth-to-space epilogues down the unfused float32-output paths
        # (NIMG_NO_S2D_CONV, NIMG_NO_D2S_OUT) also switch the bf16 storage off: those paths hand back float32 tensors
        return ops.COMPUTE == 'bf16' and ops.STORE_BF16 and ops.S2D_CONV and ops.D2S_EPILOGUE and self._h.activation == 'leaky_relu'

    @staticmethod
    def _operand(t, tb):
        """The tensor a convolution / weight gradient reads: the bf16 copy of a residual-stream tensor where one exists
        (NIMG_NO_BF16_COPY=1 keeps the float32 tensor: the A/B switch of tools/r03_ah.sh)."""
        return t if tb is None or _NO_BF16_COPY else tb

    def encode(self, x, training=False):
        L, P = self._layers, self._model
        P.refresh_images()
        t = OrderedDict()
        self._in_hw = (x.shape[1], x.shape[2])
        bf = self._bf16_inner()
        # throughput mode: e1's activation goes to HBM once, as the bf16 space-to-depth image e2 reads through the 3x3 kernels
        chain = bf and not _NO_S2D_CHAIN and x.shape[1] % 4 == 0 and x.shape[2] % 4 == 0 and \
            L['e1'].s2d_ok(x) and \
            L['e2'].s2d_chain_ok((x.shape[1] // 2, x.shape[2] // 2))
        chain_lat = chain and x.shape[1] % 8 == 0 and x.shape[2] % 8 == 0 and \
            L['elat'].s2d_chain_ok((x.shape[1] // 4, x.shape[2] // 4))
        if chain:
            t['e1s'], t['x0'] = L['e1'].forward_image(P, x, 2.0, -1.0, s2d_out=True)
        else:
            t['e1'], t['x0'] = L['e1'].forward_image(P, x, 2.0, -1.0)   # x0 = 2 x - 1 (or its bf16 space-to-depth image)
        # block 1 reads LeakyReLU(e2) while its skip adds e2 itself (:224-227): the layer writes both, the activation as bf16
        if chain:
            t['e2'], act0 = L['e2'].forward_s2d(P, t['e1s'], bf16_copy=True, copy_lrelu=True)
        elif bf:
            t['e2'], act0 = L['e2'].forward(P, t['e1'], bf16_copy=True, copy_lrelu=True)
        else:
            t['e2'], act0 = L['e2'].forward(P, t['e1']), None
        net, net_b = t['e2'], None
        t['n0'] = net
        for b in (1, 2, 3):
            inp = (ops.lrelu(net) if act0 is None else act0) if b == 1 else self._operand(net, net_b)
            t['er{}in'.format(b)] = inp
            a = L['er{}a'.format(b)].forward(P, inp, out_bf16=bf)
            t['er{}a'.format(b)] = a
            if b == 3 and chain_lat:
                # the last block's sum only feeds the stride-2 latent layer: written once, as its bf16 space-to-depth image
                t['n3s'] = L['er3b'].forward(P, a, residual=net, s2d_out=True, out_bf16=True)
                break
            want = bf and b < 3
            net = L['er{}b'.format(b)].forward(P, a, residual=net, bf16_copy=want)                  # net + conv(a), one pass
            net, net_b = net if want else (net, None)
            t['n{}'.format(b)] = net
        t['zl'] = L['elat'].forward_s2d(P, t['n3s']) if chain_lat else L['elat'].forward(P, net)
        if self._lws is None or self._lws.buf.device != x.device:
            self._lws = ops.LatentWorkspace(self._codebook.numel(), x.device)
        world, dp = parallel.world_size(), parallel.is_distributed()
        count = t['zl'].numel()
        soft = self._h.rounding == 'soft-codebook'
        # (self._codebook is torch.arange(qmin, qmax + 1): consecutive integers - the unit_codebook promise of ops.latent_fwd)
        rnd = 'identity' if soft else self._h.rounding            # identity | soft | sin (models/layers.py:118-134)
        lat, ent = ops.latent_fwd(t['zl'], P.p['latent_scaling'], self._codebook, self._lws, soft_codebook=soft, unit_codebook=True,
                                  rounding=rnd,
                                  count_global=count * world, finalize=not dp)
        if dp:              # batch-global soft histogram: 2^bpf float64 sums are all-reduced (SURVEY 8e caveat 1)
            torch.distributed.all_reduce(self._lws.hist_sums())
            ops.latent_entropy_finalize(self._lws, count * world, ent)
        t['latent'] = lat
        return lat, ent, (t if training else None)

    def decode(self, lat, training=False):
        L, P = self._layers, self._model
        P.refresh_images()
        t = OrderedDict()
        t['latent'] = lat
        net_b, bf = None, self._bf16_inner()
        net = L['d512'].forward(P, lat, d2s_out=True, bf16_copy=bf)     # depth_to_space written by the convolution itself
        net, net_b = net if bf else (net, None)
        t['i0'] = net
        for b in (1, 2, 3):
            t['dr{}in'.format(b)] = self._operand(net, net_b)
            a = L['dr{}a'.format(b)].forward(P, t['dr{}in'.format(b)], out_bf16=bf)
            t['dr{}a'.format(b)] = a
            net = L['dr{}b'.format(b)].forward(P, a, residual=net, bf16_copy=bf)
            net, net_b = net if bf else (net, None)
            t['i{}'.format(b)] = net
        t['d256in'] = self._operand(net, net_b)
        # conv + LeakyReLU + depth_to_space; not a residual stream: stored as bf16 like every tensor that only feeds operands
        t['i4'] = L['d256'].forward(P, t['d256in'], d2s_out=True, out_bf16=bf)
        t['d12'] = L['d12'].forward(P, t['i4'])
        y = ops.d2s_clip(t['d12'], 0.5, 0.5, True)               # (x + 1) / 2 then straight-through clip
        return y, (t if training else None)

    def forward(self, x, training=False):
        lat, ent, et = self.encode(x, training)
        y, dt = self.decode(lat, training)
        return y, ent, ((et, dt) if training else None)

    def backward(self, ctx, dy, entropy_coef=0.0, need_input_grad=False):
        """dy = d loss / d y; entropy_coef = d loss / d entropy.  Fills the gradient buffer."""
        et, dt = ctx
        L, P = self._layers, self._model
        hw = lambda a: (a.shape[1], a.shape[2])
        # ops.ParamGroup: the weight gradients of a block behind ONE fork of the launch stream.  OFF here (every launch forks for
        # itself, as before): config 3 loses 3 % with the groups (9425 -> 9150 patches/s, profiles/r06_fork_markers.txt) - the codec's
        # step ends with this chain, and weight gradients that start a kernel later end a kernel later.  NIMG_CODEC_FORK_GROUPS=1.
        grp = ops.ParamGroup(defer=CODEC_FORK_GROUPS)
        # ---- decoder
        dz = ops.d2s_clip_bwd(dy, 0.5)
        grp.add(lambda dz=dz: L['d12'].backward_params(P, dt['i4'], dz))
        # LeakyReLU' of the d256 layer is taken on its depth-to-space image i4 (same signs, permuted), in the epilogue of
        # this input gradient; the space_to_depth of the result is then the gradient at d256's output
        bf = self._bf16_inner()
        dz = L['d12'].backward_input(P, dz, hw(dt['i4']), act_mask=dt['i4'], s2d_out=True, out_bf16=bf)
        grp.add(lambda dz=dz: L['d256'].backward_params(P, dt['d256in'], dz))
        d_net = L['d256'].backward_input(P, dz, hw(dt['i3']), bf16_copy=bf)
        d_net, d_net_b = d_net if bf else (d_net, None)
        for b in (3, 2, 1):
            a, inp = dt['dr{}a'.format(b)], dt['dr{}in'.format(b)]
            dzs = self._operand(d_net, d_net_b)         # the matrix-core operand form of the stream's gradient
            grp.add(lambda b=b, a=a, dzs=dzs: L['dr{}b'.format(b)].backward_params(P, a, dzs))
            dza = L['dr{}b'.format(b)].backward_input(P, dzs, hw(a), act_mask=a, out_bf16=bf)
            grp.add(lambda b=b, inp=inp, dza=dza: L['dr{}a'.format(b)].backward_params(P, inp, dza))
            grp.flush()
            if b > 1:
                d_net = L['dr{}a'.format(b)].backward_input(P, dza, hw(inp), residual=d_net, bf16_copy=bf)
                d_net, d_net_b = d_net if bf else (d_net, None)
            else:       # the gradient leaves the blocks through the depth_to_space behind d512: written as its space_to_depth
                dz = L['dr1a'].backward_input(P, dza, hw(inp), residual=d_net, s2d_out=True, out_bf16=bf)
        grp.add(lambda dz=dz: L['d512'].backward_params(P, dt['latent'], dz))
        grp.flush()
        d_lat = L['d512'].backward_input(P, dz, hw(dt['latent']))
        # ---- latent
        soft = self._h.rounding == 'soft-codebook'
        dzl = ops.latent_bwd(et['zl'], P.p['latent_scaling'], et['latent'], d_lat, entropy_coef, self._codebook,
                             self._lws, dscale=P.g['latent_scaling'].view(1), soft_codebook=soft, unit_codebook=True,
                             rounding='identity' if soft else self._h.rounding)
        # ---- encoder
        if 'n3s' in et:
            grp.add(lambda dzl=dzl: L['elat'].backward_params_s2d(P, et['n3s'], dzl))
            d_net, d_net_b = L['elat'].backward_input(P, dzl, (2 * et['n3s'].shape[1], 2 * et['n3s'].shape[2])), None
        else:
            grp.add(lambda dzl=dzl: L['elat'].backward_params(P, et['n3'], dzl))
            d_net, d_net_b = L['elat'].backward_input(P, dzl, hw(et['n3'])), None
        for b in (3, 2, 1):
            a, inp = et['er{}a'.format(b)], et['er{}in'.format(b)]
            dzs = self._operand(d_net, d_net_b)
            grp.add(lambda b=b, a=a, dzs=dzs: L['er{}b'.format(b)].backward_params(P, a, dzs))
            dza = L['er{}b'.format(b)].backward_input(P, dzs, hw(a), act_mask=a, out_bf16=bf)
            grp.add(lambda b=b, inp=inp, dza=dza: L['er{}a'.format(b)].backward_params(P, inp, dza))
            grp.flush()
            # block 1 was fed LeakyReLU(e2): its input gradient goes through that activation (mask by sign of e2)
            d_net = L['er{}a'.format(b)].backward_input(P, dza, hw(inp), act_mask=et['e2'] if b == 1 else None,
                                                        residual=d_net, bf16_copy=bf, mask_activation='leaky_relu')
            d_net, d_net_b = d_net if bf else (d_net, None)
        # e1's gradient only feeds matrix-core operands (e1's weight / input gradient): stored as bf16
        if 'e1s' in et:
            e1s = et['e1s']
            grp.add(lambda e1s=e1s, g=self._operand(d_net, d_net_b): L['e2'].backward_params_s2d(P, e1s, g))
            dz1 = L['e2'].backward_input(P, self._operand(d_net, d_net_b), (2 * e1s.shape[1], 2 * e1s.shape[2]), act_mask=e1s,
                                         out_bf16=bf, mask_s2d=True)
        else:
            grp.add(lambda d_net=d_net: L['e2'].backward_params(P, et['e1'], d_net))
            dz1 = L['e2'].backward_input(P, self._operand(d_net, d_net_b), hw(et['e1']), act_mask=et['e1'], out_bf16=bf)
        grp.add(lambda dz1=dz1: L['e1'].backward_params_image(P, et['x0'], dz1))
        grp.flush()
        dx = L['e1'].backward_input_image(P, dz1, self._in_hw, 2.0) if need_input_grad else None
        ops.join_side_stream()
        return dx
