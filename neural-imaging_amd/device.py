"""
Device plumbing: which GPU this process drives, host<->device conversion of caller-owned arrays, and the
host-convertible result wrapper.  One process per GPU (LOCAL_RANK selects it), PyTorch-ROCm only for memory/streams.
"""
import os

import numpy as np
import torch


def default_device():
    if torch.cuda.is_available():
        return torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1))
    return torch.device('cpu')     # host logic (shapes, configs) only - any kernel call raises


class DeviceArray(object):
    """What process()/training_step() hand back: a device tensor that still answers .numpy(), np.asarray(), float()
    like the EagerTensors the reference's callers expect (training/validation.py:35,66-69; compression.py:135)."""
    __slots__ = ('t',)

    def __init__(self, t):
        self.t = t

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __float__(self):
        return float(self.t.detach().reshape(-1)[0].item())

    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.dim()

    @property
    def dtype(self):
        return self.t.dtype

    def __len__(self):
        return self.t.shape[0]

    def __getitem__(self, item):
        return DeviceArray(self.t[item])

    # scalar results take part in host arithmetic like the EagerTensors they stand for (np.sqrt(2 * loss), a - b, ...)
    def __mul__(self, other):
        return float(self) * other

    __rmul__ = __mul__

    def __add__(self, other):
        return float(self) + other

    __radd__ = __add__

    def __sub__(self, other):
        return float(self) - other

    def __rsub__(self, other):
        return other - float(self)

    def __truediv__(self, other):
        return float(self) / other

    def __rtruediv__(self, other):
        return other / float(self)

    def __repr__(self):
        return 'DeviceArray(shape={}, device={})'.format(self.shape, self.t.device)


def unwrap(x):
    return x.t if isinstance(x, DeviceArray) else x


def to_device(x, device, dtype=torch.float32):
    """numpy / DeviceArray / torch -> contiguous device tensor (inputs are caller-owned and never mutated)."""
    x = unwrap(x)
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32 if dtype == torch.float32 else None)).to(
        device=device, dtype=dtype)
