"""
ManipulationClassification workflow restated (workflows/manipulation_classification.py:162-285).
TEST INFRASTRUCTURE - see oracle/__init__.py.

  run_workflow  : nip -> manipulations (concat on batch axis, [native, op1..opk]) -> downsampling -> codec -> fan
  training_step : loss = CE [+ lambda_nip * mse255(by, Y)] [+ lambda_dcn * dcn_loss]; one shared Keras-Adam over
                  fan (+nip)(+dcn) parameters.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import djpeg, manip, nets
from . import tfops as T

DEFAULT_STRENGTHS = {'sharpen': 1, 'resample': 50, 'gaussian': 0.83, 'jpeg': 80, 'awgn': 5.1, 'gamma': 3, 'median': 3}
OP_ORDER = ['sharpen', 'resample', 'gaussian', 'jpeg', 'awgn', 'gamma', 'median']      # workflows/...:106-132


def apply_manipulation(name, y, s, awgn_noise=None):
    if name == 'sharpen':
        return manip.manipulation_sharpen(y, s, hsv=True)
    if name == 'resample':
        return manip.manipulation_resample(y, s)
    if name == 'gaussian':
        return manip.manipulation_gaussian(y, 5, s)
    if name == 'jpeg':
        return djpeg.djpeg_torch(y, djpeg.resolve_quality(s), 'soft')[0]   # jpeg.differentiable_jpeg
    if name == 'awgn':
        return manip.manipulation_awgn(y, s / 255, awgn_noise)
    if name == 'gamma':
        return manip.manipulation_gamma(y, s)
    if name == 'median':
        return manip.manipulation_median(y, s)
    raise ValueError(name)


class Workflow(object):
    def __init__(self, manipulations=('sharpen', 'resample', 'gaussian', 'jpeg'), downsampling='none',
                 codec='jpeg', jpeg_quality=80, jpeg_codec='soft', trainable=('nip',), dtype=torch.float64,
                 unet_seed=1234, fan_seed=4321, dcn_seed=777, nip='UNet', strengths=None, jpeg_trainable=False):
        self.dtype = dtype
        self.strengths = dict(DEFAULT_STRENGTHS)
        names = set()
        for m in manipulations:
            spec = m.split(':')
            names.add(spec[0])
            if len(spec) > 1:
                self.strengths[spec[0]] = float(spec[-1])
        if strengths:
            self.strengths.update(strengths)
        self.operations = [n for n in OP_ORDER if n in names]
        self.n_classes = len(self.operations) + 1
        self.downsampling = downsampling
        self.codec = codec
        self.jpeg_quality = jpeg_quality
        self.jpeg_codec = jpeg_codec
        self.trainable = set(trainable) | {'fan'}
        self.nip_kind = nip
        self.nip = nets.unet_init(unet_seed, dtype=dtype) if nip == 'UNet' else OrderedDict()
        self.fan = nets.fan_init(self.n_classes, fan_seed, dtype=dtype)
        self.dcn = nets.dcn_init(dcn_seed, dtype=dtype) if codec == 'dcn' else OrderedDict()
        # JPEG(trainable=True): the quantisation tables are the codec's two weights (models/jpeg.py:57-62), IJG-initialised
        self.jpeg_tables = OrderedDict()
        if codec == 'jpeg' and jpeg_trainable:
            q = djpeg.qtables_torch(jpeg_quality, dtype)
            self.jpeg_tables = OrderedDict([('Q_mtx_luma', q[0].clone()), ('Q_mtx_chroma', q[1].clone())])
        self._m = self._v = None
        self._t = 0

    # -- parameter plumbing -------------------------------------------------------------------
    def trainable_params(self):
        ps = list(self.fan.values())
        if 'nip' in self.trainable:
            ps += list(self.nip.values())
        if 'dcn' in self.trainable:
            ps += list(self.dcn.values()) + list(self.jpeg_tables.values())
        return ps

    @property
    def downsampling_factor(self):
        if self.downsampling == 'none':
            return 1
        if ':' in self.downsampling:
            return int(self.downsampling.split(':')[-1])
        return 2

    # -- forward ------------------------------------------------------------------------------
    def run_nip(self, bx):
        return nets.unet_forward(self.nip, bx) if self.nip_kind == 'UNet' else bx     # ONet = identity

    def run_manipulations(self, by, strengths=None):
        s = strengths or self.strengths
        ys = [by] + [apply_manipulation(n, by, s[n]) for n in self.operations]
        return torch.cat(ys, dim=0)

    def run_downsampling(self, bm):
        f = self.downsampling_factor
        if self.downsampling.startswith('pool'):
            return T.avg_pool(bm, f)
        if self.downsampling == 'bilinear':
            return T.resize_bilinear(bm, bm.shape[1] // f, bm.shape[1] // f)
        return bm

    def run_compression(self, bc):
        if self.codec == 'jpeg' and self.jpeg_tables:
            ql, qc = self.jpeg_tables['Q_mtx_luma'], self.jpeg_tables['Q_mtx_chroma']
            return djpeg.djpeg_torch(bc, None, self.jpeg_codec, q=torch.stack([ql, qc, qc]))[0], float('nan')
        if self.codec == 'jpeg':
            return djpeg.djpeg_torch(bc, self.jpeg_quality, self.jpeg_codec)[0], float('nan')
        if self.codec == 'dcn':
            y, ent, _ = nets.dcn_forward(self.dcn, bc)
            return y, ent
        return bc, float('nan')

    def run_workflow(self, bx, strengths=None):
        Y = self.run_nip(bx)
        m = self.run_manipulations(Y, strengths)
        c = self.run_downsampling(m)
        C, ent = self.run_compression(c)
        probs = nets.fan_forward(self.fan, C)
        return Y, c, C, ent, probs

    def batch_labels(self, b):
        return np.concatenate([k * np.ones((b,), dtype=np.int64) for k in range(self.n_classes)])

    # -- training -----------------------------------------------------------------------------
    def loss_and_grads(self, bx, by, lambda_nip=0.0, lambda_dcn=0.0, strengths=None):
        params = self.trainable_params()
        for p in params:
            p.requires_grad_(True)
            p.grad = None
        Y, c, C, ent, probs = self.run_workflow(bx, strengths)
        loss_ce = T.sparse_ce_from_probs(probs, self.batch_labels(bx.shape[0]))
        loss_nip = T.mse255(by, Y)
        loss = loss_ce
        if 'nip' in self.trainable:
            loss = loss + lambda_nip * loss_nip
        loss_dcn = None
        if self.codec == 'dcn':
            loss_dcn = nets.dcn_loss(c, C, ent)
            if 'dcn' in self.trainable:
                loss = loss + lambda_dcn * loss_dcn
        if self.jpeg_tables:               # JPEG.loss = Keras MeanSquaredError(c, C) (models/jpeg.py:197; NaN sample weight ignored)
            loss_dcn = ((c - C) ** 2).mean()
            if 'dcn' in self.trainable:
                loss = loss + lambda_dcn * loss_dcn
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for p, g in zip(params, grads)]
        for p in params:
            p.requires_grad_(False)
        parts = {'ce': float(loss_ce.detach()), 'nip': float(loss_nip.detach()),
                 'dcn': float(loss_dcn.detach()) if loss_dcn is not None else float('nan')}
        return loss.detach(), parts, params, grads, dict(Y=Y.detach(), C=C.detach(), probs=probs.detach())

    def training_step(self, bx, by, lambda_nip=0.0, lambda_dcn=0.0, learning_rate=1e-4, strengths=None):
        loss, parts, params, grads, aux = self.loss_and_grads(bx, by, lambda_nip, lambda_dcn, strengths)
        if any(bool(torch.isnan(g).any()) for g in grads):
            raise RuntimeError('gradient NaNs')                     # workflows/...:281-282
        if self._m is None:
            self._m = [torch.zeros_like(p) for p in params]
            self._v = [torch.zeros_like(p) for p in params]
        self._t += 1
        with torch.no_grad():
            T.adam_step(params, grads, self._m, self._v, self._t, learning_rate)
        return float(loss), parts
