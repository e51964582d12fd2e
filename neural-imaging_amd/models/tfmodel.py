"""
TFModel - the operator surface shared by all components, mirrored from the reference's models/tfmodel.py:86-294:
performance log, parameter access/count, save/load, restore, process.  Parameters live in ONE flat float32 device
buffer per model (`flat_params`), each named parameter being a view in the Keras layout; gradients use a twin buffer
(`flat_grads`) - that buffer is also the RCCL all-reduce unit and the fused-Adam unit.

Checkpoints: <dir>/<scoped_name>/<classname>.h5 in the Keras `save_weights` layout (helpers/keras_h5.py over a pure-Python
HDF5 codec - no h5py in this image) + optional <classname>.json {'model', 'args'}: directory layout and JSON contract of
tfmodel.py:150-182.
"""
import json
import os
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

from ..device import DeviceArray, default_device
from ..helpers import utils
from .. import ops


class ParamStore(object):
    """Named views into one flat parameter buffer and one flat gradient buffer."""

    def __init__(self, specs, device):
        self.specs = OrderedDict(specs)              # name -> shape
        self.device = device
        # every parameter starts on a 256-byte boundary of the flat buffer (float4 / uint4 loads in the kernels stay
        # aligned whatever the element counts are - e.g. the 225-element constrained filter); the gaps stay zero
        align = 64
        sizes = [int(np.prod(s)) if len(s) else 1 for s in self.specs.values()]
        total = int(sum(-(-n // align) * align for n in sizes))
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.p, self.g, self.offsets = OrderedDict(), OrderedDict(), OrderedDict()
        off = 0
        for (name, shape), n in zip(self.specs.items(), sizes):
            self.p[name] = self.flat[off:off + n].view(shape)
            self.g[name] = self.flat_grad[off:off + n].view(shape)
            self.offsets[name] = off
            off += -(-n // align) * align
        self.m = self.v = None
        self.step = 0
        self.lr_t_dev = None         # one-float device buffer with Keras Adam's bias-corrected rate: set by a captured step
                                     # (graphs.CapturedModelStep) so that the recorded Adam launch never changes
        # bf16 images of every 4-D (convolution) kernel, rebuilt by one launch per step in throughput mode
        self.images = ops.WeightImages([t for t in self.p.values() if t.dim() == 4 and min(t.shape) > 0], device) \
            if torch.device(device).type == 'cuda' else None

    def grad_range(self, first=None, before=None):
        """Contiguous view of the flat gradient buffer from parameter `first` (default: the start) up to, not including,
        parameter `before` (default: the end) - the unit of a partial all-reduce."""
        lo = 0 if first is None else self.offsets[first]
        hi = self.flat_grad.numel() if before is None else self.offsets[before]
        return self.flat_grad[lo:hi]

    def refresh_images(self):
        """Call at the top of a forward pass: the kernels of this step read the images built here."""
        if self.images is not None:
            self.images.refresh()

    def ensure_adam(self):
        if self.m is None:
            self.m = torch.zeros_like(self.flat)
            self.v = torch.zeros_like(self.flat)

    def adam(self, lr, step=None, grad_scale=1.0, skip_flag=None, lr_t_dev=None):
        """One Keras-Adam update over the whole buffer (beta1 .9, beta2 .999, eps 1e-7)."""
        self.ensure_adam()
        # a backward(join=False) may have left weight gradients running on the side streams.  The pending state lives in ops (the
        # set of side streams with work since the last join): joining is a no-op when it is empty, so every reader of the gradient
        # buffer - this update, parallel.GradientBucket.launch, ops.nan_flag's callers - simply joins
        ops.join_side_stream()
        self.grads_pending = False
        if step is None:
            self.step += 1
            step = self.step
        ops.adam_step(self.flat, self.flat_grad, self.m, self.v, lr, step, grad_scale=grad_scale, skip_flag=skip_flag,
                      lr_t_dev=lr_t_dev if lr_t_dev is not None else self.lr_t_dev)


def glorot_uniform_(t, fan_in, fan_out, gen):
    """Keras default kernel initialiser; generated on the host so runs are reproducible across devices."""
    limit = float(np.sqrt(6.0 / (fan_in + fan_out)))
    vals = (torch.rand(t.shape, generator=gen, dtype=torch.float64) * 2 - 1) * limit
    t.copy_(vals.to(torch.float32))


class TFModel(object):

    def __init__(self, device=None, **kwargs):
        self.device = torch.device(device) if device is not None else default_device()
        self._model = None           # ParamStore once constructed (name kept from the reference)
        self.reset_performance_stats()

    # -- performance log (tfmodel.py:111-131) ------------------------------------------------------------------
    @staticmethod
    def _reset_performance(metrics):
        return {k: {'training': [], 'validation': []} for k in metrics}

    def reset_performance_stats(self):
        self.performance = self._reset_performance(['loss'])

    def log_metric(self, metric, scope, value, raw=False):
        if not raw:
            if isinstance(value, DeviceArray):
                value = float(np.mean(value.numpy()))
            elif utils.is_number(value):
                value = float(value)
            else:
                value = float(np.mean([float(v) for v in value]))
        self.performance[metric][scope].append(value)

    def pop_metric(self, metric, scope):
        return self.performance[metric][scope][-1]

    # -- parameters ----------------------------------------------------------------------------------------------
    @property
    def parameters(self):
        return [] if self._model is None else list(self._model.p.values())

    @property
    def variables(self):
        return self.parameters

    @property
    def parameter_names(self):
        return [] if self._model is None else list(self._model.p.keys())

    def count_parameters(self):
        return int(sum(int(p.numel()) for p in self.parameters))

    def count_parameters_breakdown(self):
        """One row per parameter tensor: name, shape, number of values, share of the total in per cent (tfmodel.py:144-148)."""
        import pandas as pd
        total = max(self.count_parameters(), 1)
        rows = [(k, tuple(v.shape), int(v.numel()), round(100 * int(v.numel()) / total, 1)) for k, v in self._model.p.items()]
        return pd.DataFrame(rows, columns=['name', 'shape', 'parameters', 'total'])

    def state_dict(self):
        return OrderedDict((k, v.detach().cpu().numpy()) for k, v in self._model.p.items())

    def load_state_dict(self, state):
        for k, v in self._model.p.items():
            if k not in state:
                raise KeyError('missing parameter {} in checkpoint'.format(k))
            a = np.asarray(state[k], np.float32)
            if tuple(a.shape) != tuple(v.shape):
                raise ValueError('shape mismatch for {}: {} vs {}'.format(k, a.shape, tuple(v.shape)))
            v.copy_(torch.from_numpy(a))

    # -- checkpoints (tfmodel.py:150-182) ------------------------------------------------------------------------
    _h5_skip = ()            # parameter names that are constants (not Keras variables) in the reference: never stored

    def keras_layers(self):
        """[(layer name, [(weight name, ndarray)])] - the parameters grouped the way Keras lists them: consecutive
        entries that share the prefix before the last '/' form one layer ('/' -> '_' in the layer name)."""
        layers = []
        for k, v in self.state_dict().items():
            if k in self._h5_skip:
                continue
            prefix, _, leaf = k.rpartition('/')
            lname = (prefix or leaf).replace('/', '_')
            if not layers or layers[-1][0] != lname:
                layers.append((lname, []))
            layers[-1][1].append(('{}/{}:0'.format(lname, leaf), v))
        return layers

    def load_keras_weights(self, filename):
        """Keras' default h5 loading rule (topological = file order, layers without weights skipped), with the shapes
        checked one by one; scalars may be stored as () or (1,)."""
        from ..helpers import keras_h5
        flat = [(l, w, a) for l, ws in keras_h5.load_weights(filename) for w, a in ws]
        names = [k for k in self._model.p if k not in self._h5_skip]
        if len(flat) != len(names):
            raise ValueError('{} holds {} weight tensors, {} expects {}'.format(filename, len(flat), self.class_name,
                                                                                len(names)))
        state = {}
        for k, (l, w, a) in zip(names, flat):
            want = tuple(self._model.p[k].shape)
            if tuple(a.shape) != want and int(a.size) == int(np.prod(want)) and \
                    tuple(d for d in a.shape if d != 1) == tuple(d for d in want if d != 1):
                a = a.reshape(want)
            if tuple(a.shape) != want:
                raise ValueError('{}: weight {} of layer {} has shape {}, parameter {} expects {}'.format(
                    filename, w, l, tuple(a.shape), k, want))
            state[k] = a
        for k in self._h5_skip:
            if k in self._model.p:
                state[k] = self._model.p[k].detach().cpu().numpy()
        self.load_state_dict(state)

    def save_model(self, dirname, epoch=0, save_args=False, quiet=False):
        from ..helpers import keras_h5
        if not dirname.endswith(self.scoped_name):
            dirname = os.path.join(dirname, self.scoped_name)
        os.makedirs(dirname, exist_ok=True)
        keras_h5.save_weights(os.path.join(dirname, '{}.h5'.format(self.class_name.lower())), self.keras_layers())
        if save_args:
            with open(os.path.join(dirname, '{}.json'.format(self.class_name.lower())), 'w') as f:
                json.dump({'model': self.class_name, 'args': self.get_hyperparameters()}, f, indent=4)

    def load_model(self, dirname, quiet=False):
        if not dirname.endswith(self.scoped_name):
            dirname = os.path.join(dirname, self.scoped_name)
        filename = os.path.join(dirname, '{}.h5'.format(self.class_name.lower()))
        if os.path.isfile(filename):
            self.load_keras_weights(filename)
        else:
            legacy = os.path.join(dirname, '{}.npz'.format(self.class_name.lower()))      # round-1 snapshots
            if not os.path.isfile(legacy):
                raise FileNotFoundError(filename)
            with np.load(legacy) as data:
                self.load_state_dict({k: data[k] for k in data.files})
        self.reset_performance_stats()

    def migrate_model(self, dirname, mapping=None, verbose=False):
        """The reference reads variables out of a TensorFlow CHECKPOINT by name (tf.train.list_variables / load_variable,
        tfmodel.py:184-230); no TensorFlow here to parse one - Keras .h5 weight files load through load_model."""
        raise NotImplementedError('TensorFlow checkpoints cannot be read here (no TensorFlow); load_model reads the .h5 weights')

    def deploy_model(self, dirname):
        raise NotImplementedError()                      # as in the reference (tfmodel.py:292-294: a TODO there)

    @classmethod
    def restore(cls, dir_name, *, key=None, patch_size=None, **kwargs):
        candidates = list(Path(dir_name).glob('**/*.json'))
        training_log_path = str(candidates[0]) if candidates else None
        if training_log_path is None or not os.path.isfile(training_log_path):
            raise FileNotFoundError('Could not find a training log (JSON file) in {}'.format(dir_name))
        with open(training_log_path) as f:
            training_log = json.load(f)
        if key is not None:
            training_log = training_log[key]
        parameters = training_log['args']
        if patch_size is not None:
            parameters['patch_size'] = patch_size
        for k, value in parameters.items():
            if isinstance(value, str) and value and value[0] == '(' and value[-1] == ')':
                parameters[k] = tuple(int(v) for v in value[1:-1].split(',') if v.strip())
        parameters.update(kwargs)
        instance = cls(**parameters)
        instance.load_model(dir_name)
        return instance

    # -- strings -------------------------------------------------------------------------------------------------
    @property
    def class_name(self):
        return type(self).__name__

    def summary(self):
        return '{} model [{:,.0f} parameters]'.format(self.class_name, self.count_parameters())

    def summary_compact(self):
        return '{}'.format(self.class_name)

    @property
    def model_code(self):
        raise NotImplementedError()

    @property
    def scoped_name(self):
        return '{}'.format(type(self).__name__.lower())

    def get_hyperparameters(self):
        return self._h.to_json() if hasattr(self, '_h') else None

    def __repr__(self):
        try:
            extra_params = utils.join_args(self._h.changed_params())
        except Exception:
            extra_params = ''
        return '{}({})'.format(self.class_name, extra_params)

    def _has_attributes(self, attrs, message='Expected attributes not found: {}'):
        missing = [key for key in attrs if not hasattr(self, key)]
        if missing:
            raise NotImplementedError(message.format(missing))

    def process(self, x, training=False):
        raise NotImplementedError()
