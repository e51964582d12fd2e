"""
ManipulationClassification - the end-to-end imaging channel, on the HIP kernels:

    raw -> (nip) -> rgb -> (N manipulations) -> [(downsample) ->] (compression) -> (forensics) -> class probabilities

Mirrors the reference's workflows/manipulation_classification.py:14-327: same constructor arguments, the same
run_workflow / run_manipulations / run_downsampling / run_compression / training_step surface, batch layout
[native, op1 .. opk] along axis 0 with positional labels (:199-208, :257-258), joint loss
CE [+ lambda_nip * nip.loss] [+ lambda_dcn * codec.loss] (:267-277) and one Adam step over fan (+nip)(+dcn) (:279-283).

Build-side additions: explicit backward pass (no tape), device-side NaN guard, data-parallel gradient all-reduce
(neural-imaging_amd/parallel.py).
"""
from collections import OrderedDict

import os

import numpy as np
import torch

from .. import ops, parallel
from ..device import DeviceArray, to_device
from ..helpers import tf_helpers
from ..models import forensics, jpeg, pipelines


class ManipulationClassification(object):

    def __init__(self, nip_model, manipulations=None, distribution=None, fan_args=None, trainable=None,
                 raw_patch_size=128, loss_metric='L2', device=None, nan_check='eager', pipeline_fan_update=None):
        if raw_patch_size < 16 or raw_patch_size > 512:
            raise ValueError('The patch size ({}) looks incorrect, typical values should be >= 16 and <= 512'.format(
                raw_patch_size))
        self._trainable = set() if trainable is None else set(trainable)
        self._trainable.add('fan')
        if distribution is None:
            self._distribution = {'downsampling': 'pool:2', 'compression': 'jpeg',
                                  'compression_params': {'quality': 50, 'codec': 'soft'}}
        else:
            self._distribution = dict(distribution)
        if ':' in nip_model:
            nip_model, nip_pretrained_dirname = nip_model.split(':')
        else:
            nip_pretrained_dirname = None
        if not hasattr(pipelines, nip_model) or not isinstance(getattr(pipelines, nip_model), type) or \
                not issubclass(getattr(pipelines, nip_model), pipelines.NIPModel):
            raise ValueError('Invalid NIP model ({})! Available NIPs: ({})'.format(nip_model, pipelines.supported_models))
        if loss_metric not in ['L2', 'L1', 'SSIM']:
            raise ValueError('Invalid loss metric ({})!'.format(loss_metric))

        self.nip = getattr(pipelines, nip_model)(loss_metric=loss_metric, patch_size=raw_patch_size, device=device)
        self.device = self.nip.device
        if nip_pretrained_dirname is not None:
            self.nip.load_model(nip_pretrained_dirname)

        manipulations = manipulations or ['sharpen', 'resample', 'gaussian', 'jpeg']
        self._strengths = {'sharpen': 1, 'resample': 50, 'gaussian': 0.83, 'jpeg': 80, 'awgn': 5.1, 'gamma': 3,
                           'median': 3}
        self._strengths_range = {'sharpen': (0.25, 1.5), 'resample': (40, 90), 'gaussian': (0.5, 7),
                                 'jpeg': (50, 90), 'awgn': (1, 5), 'gamma': (1, 5), 'median': (3, 9)}
        manipulations_set = set()
        for m in manipulations:
            spec = m.split(':')
            manipulations_set.add(spec[0])
            if len(spec) > 1:
                self._strengths[spec[0]] = float(spec[-1])
        if any(x not in self._strengths.keys() for x in manipulations_set):
            raise ValueError('Unsupported manipulation requested! Available: {}'.format(self._strengths.keys()))

        self._jpeg_manip = jpeg.JPEG(None, 'soft', device=self.device)       # jpeg.differentiable_jpeg's shared codec
        self._operations = OrderedDict()
        self._forensics_classes = ['native']
        builders = OrderedDict([('sharpen', tf_helpers.Sharpen), ('resample', tf_helpers.Resample),
                                ('gaussian', tf_helpers.Gaussian), ('jpeg', None), ('awgn', tf_helpers.Awgn),
                                ('gamma', tf_helpers.Gamma), ('median', tf_helpers.Median)])
        for name, cls in builders.items():
            if name not in manipulations_set:
                continue
            if name == 'jpeg':
                self._operations[name] = _JpegManipulation(self._jpeg_manip)
            elif cls is None:
                raise NotImplementedError('manipulation {} is not built yet'.format(name))
            else:
                self._operations[name] = cls()
            self._forensics_classes.append('{}:{}'.format(name, self._strengths[name]))
        assert len(self._forensics_classes) == self.n_classes

        if self._distribution['compression'] == 'jpeg':
            self.codec = jpeg.JPEG(device=self.device, **self._distribution['compression_params'])
        elif self._distribution['compression'] == 'dcn':
            from ..models import compression
            params = self._distribution['compression_params']
            self.codec = params['model'] if 'model' in params else compression.TwitterDCN.restore(
                params['dirname'], device=self.device)
            self._distribution = dict(self._distribution, compression_params={
                k: v for k, v in params.items() if k != 'model'})
        elif self._distribution['compression'] == 'none':
            self.codec = None
        else:
            raise ValueError('Unsupported channel compression {}'.format(self._distribution['compression']))
        if 'dcn' in self._trainable and (self.codec is None or len(self.codec.parameters) == 0):
            raise ValueError('The current codec does not appear to be trainable: {}!'.format(
                None if self.codec is None else self.codec.class_name))

        self._is_dcn = self._distribution['compression'] == 'dcn'
        fan_input_patch = 2 * raw_patch_size // self.downsampling_factor
        self.fan = forensics.FAN(n_classes=self.n_classes, patch_size=fan_input_patch, device=self.device,
                                 **(fan_args or {}))
        self._parameters = list(self.fan.parameters)
        if 'nip' in self._trainable:
            self._parameters.extend(self.nip.parameters)
        if 'dcn' in self._trainable:
            self._parameters.extend(self.codec.parameters)
        self._step = 0
        self._nan_check = nan_check
        # device-side NaN guard: _nan_flag is this step's flag (it gates the step's Adam updates), _nan_seen keeps every
        # flag raised since the last check_nan() (nan_check='deferred' reads it once per epoch / bench run, not per step)
        self._nan_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._nan_seen = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._labels_cache = {}
        self._lr_t_dev = None
        self._bucket = parallel.GradientBucket()
        # pipeline_fan_update (single process, gradients flowing upstream, nan_check='deferred'): the FAN's chip-filling weight
        # gradients and its Adam update are not issued inside the step that produced them but at the START of the next one,
        # beside that step's UNet forward / manipulations / codec (latency- and byte-bound kernels on an otherwise idle chip),
        # and are joined in front of its FAN forward.  Same kernels on the same operands in the same update order: the weights
        # after finish_pending() are bit-identical to the unpipelined run (tests/test_gpu_models.py).  What changes: the FAN's
        # weights / gradient buffer are one update behind between steps until finish_pending() is called (run_* / check_nan /
        # the summaries call it), and a NaN in the FAN's weight gradients gates the FAN's update only (it is seen after the
        # upstream models' updates of that step were decided).
        if pipeline_fan_update is None:           # NIMG_PIPELINE_FAN=1: default for workflows built without the argument (A/B runs)
            pipeline_fan_update = os.environ.get('NIMG_PIPELINE_FAN', '0') == '1'
        self._pipeline_fan = bool(pipeline_fan_update)
        self._pending_fan = None

    # ------------------------------------------------------------------------------------------------------------
    @property
    def n_classes(self):
        return len(self._operations) + 1

    @property
    def downsampling_factor(self):
        if self._distribution['downsampling'] == 'none':
            return 1
        elif ':' in self._distribution['downsampling']:
            return int(self._distribution['downsampling'].split(':')[-1])
        return 2

    def _batch_labels(self, batch_size):
        return np.concatenate([x * np.ones((batch_size,), dtype=np.int32) for x in range(self.n_classes)])

    def _device_labels(self, batch_size):
        if batch_size not in self._labels_cache:
            self._labels_cache[batch_size] = torch.from_numpy(self._batch_labels(batch_size)).to(self.device)
        return self._labels_cache[batch_size]

    # -- forward pieces (device tensors in/out) ------------------------------------------------------------------
    def _manipulations(self, Y, randomize=False, override=None, training=False, m=None):
        """m: optional pre-allocated class batch whose first slice already IS Y (the NIP developed straight into it)."""
        override = override if override is not None else self._strengths
        b = Y.shape[0]
        if m is None:
            m = torch.empty((self.n_classes * b,) + tuple(Y.shape[1:]), dtype=torch.float32, device=Y.device)
            m[:b].copy_(Y)
        elif Y.data_ptr() != m.data_ptr() or m.shape[0] != self.n_classes * b:
            raise ValueError('the class batch must start with the developed images')
        ctxs = []
        names = list(self._operations.keys())
        if randomize:         # one draw per operation and step from the numpy global stream (workflows/...:205); under data
            # parallelism every rank draws (the streams stay in step) and rank 0's values are used everywhere
            drawn = parallel.broadcast_floats([np.random.uniform(*self._strengths_range[n]) for n in names], Y.device)
            strengths = dict(zip(names, drawn))
        else:
            strengths = override
        for k, (name, op) in enumerate(self._operations.items()):
            s = strengths[name]
            _, ctx = op.forward(Y, s, out=m[(k + 1) * b:(k + 2) * b], training=training)
            ctxs.append(ctx)
        return m, ctxs

    def _downsampling(self, m):
        factor = self.downsampling_factor
        mode = self._distribution['downsampling']
        if mode.startswith('pool'):
            return ops.avgpool(m, factor)
        elif mode == 'bilinear':
            # tf.image.resize(bilinear) to shape[1] // factor on BOTH axes (workflows/...:237-238): one banded operator
            op = self._bilinear_operator(m.shape[1], factor, m.device)
            tmp = ops.sparse_axis_apply(m, op.fwd, 0, op.out_size)
            return ops.sparse_axis_apply(tmp, op.fwd, 1, op.out_size)
        elif mode == 'none':
            return m
        raise ValueError('Unsupported channel down-sampling {}'.format(mode))

    def _bilinear_operator(self, size, factor, device):
        from ..helpers import kernels as hk
        key = (size, factor, str(device))
        cache = self.__dict__.setdefault('_bilinear_cache', {})
        if key not in cache:
            cache[key] = ops.AxisOperator(hk.bilinear_axis_matrix(size, size // factor), device)
        return cache[key]

    def _downsampling_bwd(self, dc):
        mode = self._distribution['downsampling']
        if mode.startswith('pool'):
            return ops.avgpool_bwd(dc, self.downsampling_factor)
        if mode == 'bilinear':
            op = self._bilinear_operator(dc.shape[1] * self.downsampling_factor, self.downsampling_factor, dc.device)
            tmp = ops.sparse_axis_apply(dc, op.bwd, 0, op.in_size)
            return ops.sparse_axis_apply(tmp, op.bwd, 1, op.in_size)
        return dc

    # -- reference surface ---------------------------------------------------------------------------------------
    def run_workflow(self, batch_x, augment=False, training=False):
        """Returns batch_Y, batch_c, batch_C, entropy, probabilities (workflows/...:162-176)."""
        self.finish_pending()
        x = to_device(batch_x, self.device)
        Y = self.nip.forward(x)[0]
        m, _ = self._manipulations(Y, augment)
        c = self._downsampling(m)
        C, ent, _ = self._codec_forward(c)
        entropy = np.nan if ent is None else DeviceArray(ent)
        probs = self.fan.forward(C)[0]
        return DeviceArray(Y), DeviceArray(c), DeviceArray(C), entropy, DeviceArray(probs)

    def run_workflow_to_decisions(self, batch_x):
        return self.run_workflow(batch_x)[-1].numpy().argmax(axis=1)

    def run_manipulations(self, batch_y, randomize=False, override=None):
        return DeviceArray(self._manipulations(to_device(batch_y, self.device), randomize, override)[0])

    def manipulations_timing(self, batch_y):
        """{operation: seconds} of one application of every manipulation at its default strength (workflows/...:210-221).  The
        launches are asynchronous here: the device is drained before and after each one, so the figures are kernel time, not
        the time to queue a launch."""
        from datetime import datetime
        Y = to_device(batch_y, self.device)
        times = {}
        for name, op in self._operations.items():
            torch.cuda.synchronize(self.device)
            d1 = datetime.now()
            op.forward(Y, self._strengths[name])
            torch.cuda.synchronize(self.device)
            times[name] = (datetime.now() - d1).total_seconds()
        return times

    def run_downsampling(self, batch_y):
        return DeviceArray(self._downsampling(to_device(batch_y, self.device)))

    def run_compression(self, batch_y, return_entropy=False):
        y = to_device(batch_y, self.device)
        C, ent, _ = self._codec_forward(y)
        return (DeviceArray(C), np.nan if ent is None else DeviceArray(ent)) if return_entropy else DeviceArray(C)

    def run_rgb_to_fan(self, batch_Y):
        m = self._manipulations(to_device(batch_Y, self.device))[0]
        c = self._downsampling(m)
        return self._codec_forward(c)[0].cpu().numpy()

    def run_rgb_to_probabilities(self, batch_Y):
        self.finish_pending()
        C = torch.from_numpy(self.run_rgb_to_fan(batch_Y)).to(self.device)
        return self.fan.forward(C)[0].cpu().numpy()

    # -- training ------------------------------------------------------------------------------------------------
    def _codec_forward(self, c, training=False):
        """-> (C, entropy or None, ctx); the learned codec returns its entropy, JPEG has none (jpeg.py:245-249)."""
        if self.codec is None:
            return c, None, None
        if self._is_dcn:
            return self.codec.forward(c, training=training)
        C, ctx = self.codec.forward(c, training=training)
        return C, None, ctx

    def training_step(self, batch_x, batch_y, lambda_nip=0, lambda_dcn=0, augment=False, learning_rate=1e-4):
        """One joint optimisation step (workflows/...:260-285).  Returns (loss, {'ce','nip','dcn'})."""
        x = to_device(batch_x, self.device)
        target = to_device(batch_y, self.device)
        b = x.shape[0]
        train_nip = 'nip' in self._trainable and self.nip.count_parameters() > 0
        train_dcn = 'dcn' in self._trainable
        need_upstream = train_nip or train_dcn
        world = parallel.world_size()

        # ---- the previous step's FAN weight gradients (pipelined update): on the side streams, beside this step's forward
        self._launch_pending_fan()

        # ---- forward
        m = None
        if getattr(self.nip, 'writes_into', False):       # develop straight into the first slice of the class batch (saves a copy)
            b = x.shape[0]
            m = torch.empty((self.n_classes * b, 2 * x.shape[1], 2 * x.shape[2], 3), dtype=torch.float32, device=x.device)
            Y, nctx = self.nip.forward(x, training=train_nip, out=m[:b])
        else:
            Y, nctx = self.nip.forward(x, training=train_nip)
        m, mctxs = self._manipulations(Y, augment, training=train_nip, m=m)
        c = self._downsampling(m)
        C, entropy, cctx = self._codec_forward(c, training=need_upstream)
        self._finish_pending_fan()               # ... joined, checked and applied before the FAN is used again
        _, fctx = self.fan.forward(C, self._device_labels(b), training=True)

        # ---- backward
        ops.int_fill(self._nan_flag, 0)
        # one process, gradients flowing on upstream: the FAN's weight gradients are issued BEHIND its input-gradient chain and
        # run beside the codec / manipulation / UNet backward (HBM- and latency-bound kernels that leave the matrix cores idle)
        # instead of beside the FAN's own input gradients (which fill the chip themselves); their NaN flag is taken at the end.
        # With ranks to talk to, the FAN bucket has to leave early instead: the old order.
        # (measured: C4, 320 FAN images, 8.33 -> 8.19 ms; config 5, 80 images, 11.2 -> 11.3 ms - there the 5x5 input gradients leave
        # room on the chip and the old order wins: forensics.LATE_MIN_IMAGES.  The codec's weight gradients moved the same way cost
        # config 5 another 1 % and stay where they were.)
        late_fan = forensics.LATE_PARAMS and need_upstream and not parallel.is_distributed() and \
            C.shape[0] >= forensics.LATE_MIN_IMAGES
        pipelined = self._pipeline_fan and need_upstream and not parallel.is_distributed() and self._nan_check == 'deferred' and \
            self._lr_t_dev is None              # (a captured step replays fixed buffers: no work may cross its boundary)
        deferred = [] if pipelined else None
        if pipelined:
            late_fan = True
        loss_ce, dC = self.fan.backward(fctx, need_input_grad=need_upstream, join=not late_fan, defer=deferred)
        if not late_fan:
            ops.nan_flag(self.fan._model.flat_grad, self._nan_flag)
            self._bucket.launch(self.fan._model.flat_grad)          # overlaps with the rest of the backward pass
        loss_dcn = None
        dc = None
        if need_upstream:
            if self.codec is None:
                dc = dC
            elif self._is_dcn:
                # codec.loss(batch_c, batch_C, entropy) = l2_loss(c - C) + w_H * H (compression.py:92-93) joins the
                # objective only when the codec is trained (workflows/...:275-277).  l2_loss is a SUM over the batch, so
                # under data parallelism its gradient must not be averaged: pre-multiply by the world size.
                lam = float(lambda_dcn) * world if train_dcn else 0.0
                dc_direct = None
                if train_dcn:
                    l2, _ = ops.l2_loss(c, C, grad_scale=lam, grad_out=dC, accumulate=True)       # d/dC: lam * (C - c)
                    if train_nip:
                        dc_direct = torch.empty_like(c)
                        ops.l2_loss(C, c, grad_scale=lam, grad_out=dc_direct, accumulate=False)   # d/dc: lam * (c - C)
                    loss_dcn = (l2, entropy)
                dc = self.codec.backward(cctx, dC, entropy_coef=lam * self.codec._h.entropy_weight,
                                         need_input_grad=train_nip)
                if train_nip and dc_direct is not None:
                    ops.add(dc, dc_direct, out=dc)
                if train_dcn:
                    ops.nan_flag(self.codec._model.flat_grad, self._nan_flag)
                    self._bucket.launch(self.codec._model.flat_grad)
            else:
                dc_direct = None
                if train_dcn:
                    # trainable quantisation tables (JPEG(trainable=True), models/jpeg.py:57-62): the codec's loss is Keras'
                    # MeanSquaredError(c, C) on [0,1] images (models/jpeg.py:197; its NaN "entropy" sample weight is ignored,
                    # SURVEY 8a quirk 11) = mse255 / 255^2, a MEAN - averaged over ranks like the other terms
                    lam = float(lambda_dcn) / (255.0 * 255.0)
                    mse, _ = ops.mse255(C, c, grad_scale=lam, grad_out=dC, accumulate=True)          # d/dC
                    if train_nip:
                        dc_direct = torch.empty_like(c)
                        ops.mse255(c, C, grad_scale=lam, grad_out=dc_direct, accumulate=False)       # d/dc
                    loss_dcn = (mse, None)
                dc = self.codec.backward(cctx, dC)           # fills the table gradients when the tables are trainable
                if dc_direct is not None:
                    ops.add(dc, dc_direct, out=dc)
                if train_dcn:
                    ops.nan_flag(self.codec._model.flat_grad, self._nan_flag)
                    self._bucket.launch(self.codec._model.flat_grad)
        if train_nip:
            dm = self._downsampling_bwd(dc)
            # d loss / d Y = the native branch's gradient + every manipulation's input gradient, summed in one pass (in place
            # in dm[:b]: dm is this step's own scratch)
            parts_dY = [dm[:b]] + [op.backward(mctxs[k], dm[(k + 1) * b:(k + 2) * b])
                                   for k, (name, op) in enumerate(self._operations.items())]
            dY = dm[:b]
            head = self.nip.head_gradient(parts_dY, Y, target, float(lambda_nip)) if hasattr(self.nip, 'head_gradient') else None
            nip_kw = {}
            if head is not None:        # sum + L2 term + the gradient of depth_to_space / clip in one pass (UNet, L2 loss)
                loss_nip, nip_kw['dz_head'] = head
            else:
                for i in range(0, len(parts_dY) - 1, 5):            # nimg_add_n takes up to 6 tensors per launch
                    chunk = ([dY] if i else [parts_dY[0]]) + parts_dY[i + 1:i + 6]
                    ops.add_n(chunk, out=dY)
                loss_nip, _ = self.nip.loss_and_grad(Y, target, grad_scale=float(lambda_nip), grad_out=dY, accumulate=True)
            if parallel.is_distributed() and hasattr(self.nip, 'decoder_grads'):
                # two buckets: the decoder's gradients (its backward runs first) travel while the encoder backward
                # computes; only the encoder's slice is exposed at the end of the step
                dec, enc = self.nip.decoder_grads()
                def decoder_done():
                    ops.nan_flag(dec, self._nan_flag)
                    self._bucket.launch(dec)
                self.nip.backward(nctx, dY, on_decoder_done=decoder_done, **nip_kw)
                ops.nan_flag(enc, self._nan_flag)
                self._bucket.launch(enc)
            else:
                self.nip.backward(nctx, dY, **nip_kw)
                ops.nan_flag(self.nip._model.flat_grad, self._nan_flag)
                self._bucket.launch(self.nip._model.flat_grad)
        else:
            loss_nip, _ = self.nip._loss_fn(Y, target)
        if late_fan and not pipelined:
            ops.nan_flag(self.fan._model.flat_grad, self._nan_flag)         # joins the side streams
            self._bucket.launch(self.fan._model.flat_grad)
        parallel.all_reduce_flag(self._nan_flag)
        ops.int_max_(self._nan_seen, self._nan_flag)
        if self._nan_check == 'eager' and int(self._nan_flag.item()) != 0:       # host sync, like the reference
            self._nan_seen.zero_()
            self._bucket.wait()
            raise RuntimeError('gradient NaNs in the workflow step')
        self._bucket.wait()

        # ---- one shared Adam step (lr assigned every step, workflows/...:279)
        self._step += 1
        gscale = 1.0 / world
        rate = self._lr_t_dev              # None, or the device-resident rate of a captured step (graphs.CapturedStep)
        if pipelined:
            # the FAN's part of this step - the deferred weight-gradient launches, its NaN check (on top of this step's flag, which
            # stays in _nan_flag until the next backward pass resets it) and its Adam update - waits for the next step
            self._pending_fan = (deferred, learning_rate, self._step, gscale)
        else:
            self.fan._model.adam(learning_rate, self._step, gscale, skip_flag=self._nan_flag, lr_t_dev=rate)
        if train_nip:
            self.nip._model.adam(learning_rate, self._step, gscale, skip_flag=self._nan_flag, lr_t_dev=rate)
        if train_dcn:
            self.codec._model.adam(learning_rate, self._step, gscale, skip_flag=self._nan_flag, lr_t_dev=rate)

        dcn_value = np.nan
        if loss_dcn is not None and loss_dcn[1] is None:       # JPEG with trainable tables: MeanSquaredError on [0,1] images
            dcn_value = _LazySum(torch.zeros_like(loss_dcn[0]), loss_dcn[0], 1.0 / (255.0 * 255.0))
        elif loss_dcn is not None:         # codec.loss = l2 + w H, read from the device only when somebody looks at it
            dcn_value = _LazySum(loss_dcn[0], loss_dcn[1], self.codec._h.entropy_weight)
        loss = _LazyLoss(loss_ce, loss_nip, float(lambda_nip) if 'nip' in self._trainable else 0.0,
                         dcn_value if loss_dcn is not None else None, float(lambda_dcn))
        return loss, {'ce': DeviceArray(loss_ce), 'nip': DeviceArray(loss_nip), 'dcn': dcn_value}

    def _launch_pending_fan(self):
        if self._pending_fan is not None and self._pending_fan[0] is not None:
            with ops.one_fork():
                for fn in self._pending_fan[0]:
                    fn()
            self._pending_fan = (None,) + self._pending_fan[1:]

    def _finish_pending_fan(self):
        if self._pending_fan is None:
            return
        self._launch_pending_fan()
        _, lr, step, gscale = self._pending_fan
        self._pending_fan = None
        ops.nan_flag(self.fan._model.flat_grad, self._nan_flag)             # joins the side streams
        ops.int_max_(self._nan_seen, self._nan_flag)
        self.fan._model.adam(lr, step, gscale, skip_flag=self._nan_flag)

    def finish_pending(self):
        """Pipelined FAN update (constructor): issue and apply what the last training step left for the next one.  Call before
        reading the FAN's weights or gradients outside training_step; the run_* methods, check_nan and the summaries do."""
        self._finish_pending_fan()

    def check_nan(self):
        self.finish_pending()
        """Deferred NaN guard (nan_check='deferred'): raises if any step since the last check produced NaN grads (those
        steps' Adam updates were skipped on the device)."""
        seen = int(self._nan_seen.item())
        self._nan_seen.zero_()
        if seen != 0:
            raise RuntimeError('gradient NaNs in the workflow step')

    # -- strings -------------------------------------------------------------------------------------------------
    def summary_compact(self):
        return '{class_name}[{trainables}]: {nip} -> [{manips}] {pool}{codec}-> {fan}'.format(
            class_name=type(self).__name__, nip=self.nip.class_name,
            manips=''.join([x[0] for x in self._forensics_classes]),
            trainables=''.join([x[0] for x in self.trainable_models]),
            pool='' if self._distribution['downsampling'] == 'none' else '-> {} '.format(
                self._distribution['downsampling']),
            codec='' if self.codec is None else '-> {} '.format(self.codec.summary_compact()), fan='FAN')

    def summary(self):
        return '{class_name}[opt={trainables}]: {input} -> {nip} -> {n_ops} manipulations [{manips}] {pool}{codec}-> {fan}'.format(
            class_name=type(self).__name__, input='(rgb)' if self.nip.x.shape[-1] == 3 else '(raw)',
            nip=self.nip.class_name, n_ops=self.n_classes - 1,
            manips=''.join([x[0] for x in self._forensics_classes]),
            trainables=''.join([x[0] for x in self.trainable_models]),
            pool='' if self._distribution['downsampling'] == 'none' else '-> {} '.format(
                self._distribution['downsampling']),
            codec='' if self.codec is None else '-> {} '.format(self.codec.summary_compact()),
            fan='FAN -> (prob. {} classes)'.format(self.n_classes))

    def details(self):
        out = [self.summary()]
        out.append('Input         : {} {}'.format(self.nip.x.shape, '(rgb)' if self.nip.x.shape[-1] == 3 else '(raw)'))
        out.append('Camera ISP    : {}'.format(self.nip.summary()))
        out.append('Manipulations : {} -> {}'.format(self.n_classes, self._forensics_classes))
        out.append('Downsampling  : {}'.format(self._distribution['downsampling']))
        out.append('Codec         : {}'.format('' if self.codec is None else self.codec.summary()))
        out.append('Forensics     : {}'.format(self.fan.summary()))
        out.append('Output        : {}'.format(self.fan.y.shape))
        return '\n'.join(out)

    def is_trainable(self, model):
        return model in self._trainable

    @property
    def trainable_models(self):
        return tuple(x for x in self._trainable)


class _JpegManipulation(object):
    """The 'jpeg' manipulation: jpeg.differentiable_jpeg(x, quality) (workflows/...:118-120)."""

    def __init__(self, codec):
        self.codec = codec

    def forward(self, x, quality, out=None, training=False):
        return self.codec.forward(x, quality, training=training, out=out)

    def backward(self, ctx, dy):
        return self.codec.backward(ctx, dy)


class _LazySum(DeviceArray):
    """a + w * b of two device scalars, evaluated on the host only when somebody reads it."""
    __slots__ = ('a', 'b', 'w')

    def __init__(self, a, b, w):
        self.t = a
        self.a, self.b, self.w = a, b, float(w)

    def numpy(self):
        return np.asarray(float(self.a.detach().cpu().reshape(())) + self.w * float(self.b.detach().cpu().reshape(())))

    def __float__(self):
        return float(self.numpy())


class _LazyLoss(DeviceArray):
    """loss = ce + lambda_nip * nip [+ lambda_dcn * dcn], evaluated on the host only when somebody reads it (keeps the step
    asynchronous and capturable)."""
    __slots__ = ('ce', 'nip', 'lam', 'dcn', 'lam_dcn')

    def __init__(self, ce, nip, lam, dcn=None, lam_dcn=0.0):
        self.t = ce
        self.ce, self.nip, self.lam, self.dcn, self.lam_dcn = ce, nip, lam, dcn, lam_dcn

    def numpy(self):
        v = self.ce.detach().cpu().numpy().reshape(()) + np.float32(self.lam) * self.nip.detach().cpu().numpy().reshape(())
        extra = 0.0 if self.dcn is None else self.lam_dcn * float(self.dcn)
        return np.asarray(v + np.float32(extra), dtype=np.float32)

    def __float__(self):
        return float(self.numpy())
