"""
JPEG compression models on the fused HIP kernel.  Mirrors the reference's models/jpeg.py:

  DifferentiableJPEG (:45-159)  - the codec itself; here one fused kernel per direction (nimg_djpeg_fwd/_bwd)
  JPEG (:162-286)               - framework wrapper: quality resolution/randomisation, codec switch
  differentiable_jpeg (:38-42)  - lazily created shared instance used by the 'jpeg' manipulation

Differences by design: Q tables are passed to the kernel per call, so the temporary table swap of JPEG.process
(:235-243, not re-entrant in the reference) has no shared state here.  The 'libjpeg' codec (:227-233) is a CPU
validation path through imageio and is out of scope - it raises NotImplementedError.
"""
import numpy as np
import torch

from .. import ops
from ..device import DeviceArray, default_device, to_device
from ..helpers.utils import is_number
from ..compression.jpeg_helpers import jpeg_qf_estimation
from .tfmodel import ParamStore, TFModel

_common_codec = None


def is_valid_quality(quality):
    if is_number(quality) and 1 <= quality <= 100:
        return True
    elif hasattr(quality, '__getitem__') and len(quality) > 1 and all((1 <= x <= 100) for x in quality):
        return True
    return False


def differentiable_jpeg(x, quality):
    global _common_codec
    if _common_codec is None:
        _common_codec = JPEG(None, 'soft')
    return _common_codec.process(x, quality)


class DifferentiableJPEG(object):
    """Holds the codec settings; __call__ returns (y, X_dequantised) like the Keras model's call()."""

    def __init__(self, quality=None, rounding_approximation='sin', rounding_approximation_steps=5, trainable=False,
                 device=None):
        if quality is not None and not is_valid_quality(quality):
            raise ValueError('Invalid JPEG quality: requires int in [1,100] or an iterable with least 2 such numbers')
        if rounding_approximation is not None and rounding_approximation not in ['sin', 'harmonic', 'soft']:
            raise ValueError('Unsupported rounding approximation: {}'.format(rounding_approximation))
        self.quality = quality
        self.trainable = trainable
        self.rounding_approximation = rounding_approximation
        self.rounding_approximation_steps = rounding_approximation_steps
        self.device = device
        self._q_cache = {}
        # trainable=True (models/jpeg.py:57-62): two (8,8) weights 'Q_mtx_luma' / 'Q_mtx_chroma', initialised from the IJG tables
        # of a numeric quality or ones; the chroma table serves Cb and Cr (:125-128).  Both sit back to back in ONE flat buffer
        # (64 floats each = the ParamStore's alignment), which is also the (2,8,8) layout nimg_djpeg_bwd_dq writes.
        self.params = None
        if trainable:
            self.params = ParamStore([('Q_mtx_luma', (8, 8)), ('Q_mtx_chroma', (8, 8))],
                                     device if device is not None else default_device())
            init = ops.qtables_device(quality if is_number(quality) else None, 'cpu')
            self.params.p['Q_mtx_luma'].copy_(init[0])
            self.params.p['Q_mtx_chroma'].copy_(init[1])

    def qtables(self, quality, device):
        key = (quality if is_number(quality) else None, str(device))
        if key not in self._q_cache:
            self._q_cache[key] = ops.qtables_device(key[0], device)
        return self._q_cache[key]

    def trainable_tables(self):
        """(3,8,8) [Y, Cb, Cr] view of the trainable weights for the kernel (one small gather per forward pass)."""
        return self.params.flat.view(2, 8, 8)[[0, 1, 1]].contiguous()

    def __call__(self, x, quality=None):
        q = self.trainable_tables() if (self.trainable and (quality is None or quality == self.quality)) else \
            self.qtables(self.quality if quality is None else quality, x.device)
        y, _, _, xdq = ops.djpeg_fwd(x, q, self.rounding_approximation, want_mask=False, want_xdq=True)
        return y, xdq


class JPEG(TFModel):

    def __init__(self, quality=None, codec='soft', trainable=False, device=None):
        super().__init__(device=device)
        if codec is not None and codec not in ['libjpeg', 'soft', 'sin', 'harmonic']:
            raise ValueError('Unsupported codec version: {}'.format(codec))
        self._codec_model = None if codec == 'libjpeg' else DifferentiableJPEG(quality, codec, trainable=trainable,
                                                                               device=self.device)
        # no trainable parameters unless trainable=True: then the two quantisation tables (models/jpeg.py:57-62)
        self._model = self._codec_model.params if (trainable and self._codec_model is not None) else ParamStore([], self.device)
        self.trainable = bool(trainable)
        self.codec = codec
        self.quality = quality
        self.loss = self._mse

    def _mse(self, y_true, y_pred, sample_weight=None):
        """tf.keras.losses.MeanSquaredError (models/jpeg.py:197). The workflow passes entropy = NaN as sample_weight
        (workflows/...:269); it is ignored here so the NaN does not propagate."""
        a, b = to_device(y_true, self.device), to_device(y_pred, self.device)
        return float(DeviceArray(ops.mse255(a, b)[0])) / (255.0 * 255.0)

    def reset_performance_stats(self):
        self.performance = self._reset_performance(['entropy', 'ssim', 'psnr'])

    @staticmethod
    def resolve_quality(quality):
        """Quality resolution of JPEG.process (models/jpeg.py:210-225)."""
        if not is_valid_quality(quality):
            raise ValueError('Invalid or unspecified JPEG quality!')
        if hasattr(quality, '__getitem__') and len(quality) > 2:
            return int(np.random.choice(quality))
        elif hasattr(quality, '__getitem__') and len(quality) == 2:
            return int(np.random.randint(quality[0], quality[1]))
        elif is_number(quality) and 1 <= quality <= 100:
            return int(quality)
        raise ValueError('Invalid quality! {}'.format(quality))

    # forward/backward used by the workflow ----------------------------------------------------------------------
    def forward(self, x, quality=None, training=False, out=None):
        if self._codec_model is None:
            raise NotImplementedError('the libjpeg codec is CPU validation tooling (out of scope, SURVEY 2 row 10)')
        # the quality is resolved FIRST (an invalid / unspecified one raises even with trainable tables); the model's own tables -
        # the learned ones - serve iff the resolved quality is the constructor's, any other quality swaps in its IJG tables for
        # this call (models/jpeg.py:210-243)
        resolved = self.resolve_quality(self.quality if quality is None else quality)
        learned = bool(self.trainable) and is_number(self.quality) and resolved == self.quality
        if learned:
            q = self._codec_model.trainable_tables()
        else:
            q = self._codec_model.qtables(resolved, x.device)
        y, mask, _, _ = ops.djpeg_fwd(x, q, self.codec, want_mask=training, out=out)
        return y, ({'x': x, 'mask': mask, 'q': q, 'learned': learned} if training else None)

    def backward(self, ctx, dy, accumulate=False):
        """d loss / d x; with trainable tables also fills (accumulate: adds to) their gradients in the model's gradient buffer."""
        if ctx.get('learned'):
            return ops.djpeg_bwd(ctx['x'], dy, ctx['mask'], ctx['q'], self.codec, dq=self._model.flat_grad, accumulate=accumulate)
        return ops.djpeg_bwd(ctx['x'], dy, ctx['mask'], ctx['q'], self.codec)

    def process(self, batch_x, quality=None, return_entropy=False):
        """Compress a batch (NHW3 rgb): number -> that quality; 2 numbers -> random integer in [lo, hi);
        more -> random choice (models/jpeg.py:202-251).  Entropy is NaN, as in the reference (:245-249)."""
        y, _ = self.forward(to_device(batch_x, self.device), quality)
        y = DeviceArray(y)
        return (y, np.nan) if return_entropy else y

    def __repr__(self):
        if self._codec_model is not None:              # models/jpeg.py:253-257
            return 'JPEG(quality={},codec="{}",trainable={})'.format(self.quality, self.codec, bool(self.trainable))
        return 'JPEG(quality={},codec="{}")'.format(self.quality, self.codec)

    def summary(self, quality=None):
        return 'JPEG ({}) {}'.format(self.codec, self._quality_mode(quality))

    def summary_compact(self, quality=None):
        return 'JPEG ({}) {}'.format(self.codec, self._quality_mode(quality))

    def _model_tables(self):
        """The codec model's own (luma, chroma) tables on the host: the learned weights with trainable=True, else the IJG tables of
        a numeric constructor quality, else ones (models/jpeg.py:57-66)."""
        if self.trainable:
            t = self._codec_model.params.flat.detach().cpu().view(2, 8, 8).numpy()
            return t[0], t[1]
        q = self._codec_model.qtables(self.quality, torch.device('cpu')).numpy()
        return q[0], q[1]

    def estimate_qf(self, channel=0):
        """Closest IJG quality of the current tables.  Like the reference (models/jpeg.py:265-269) the LUMA table is compared
        whatever `channel` says - `channel` only picks the IJG family it is compared with."""
        return jpeg_qf_estimation(self._model_tables()[0], channel)

    def _quality_mode(self, quality=None):
        quality = quality or self.quality
        if self.trainable and self._codec_model is not None:           # models/jpeg.py:274-278
            luma, chroma = self._model_tables()
            return 'trainable QF~{}/{}'.format(jpeg_qf_estimation(luma, 0), jpeg_qf_estimation(chroma, 1))
        elif is_number(quality):
            return 'QF={}'.format(quality)
        elif hasattr(quality, '__getitem__') and len(quality) == 2:
            return 'QF~[{},{}]'.format(*quality)
        elif hasattr(quality, '__getitem__') and len(quality) > 2:
            return 'QF~{{{}}}'.format(','.join(str(x) for x in quality))
        return 'QF=?'
