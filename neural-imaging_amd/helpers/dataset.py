"""
Training data for the imaging pipelines (reference helpers/dataset.py).

`Dataset` is the reference's host class: full-resolution training images (RAW uint16 RGGB stacks + rendered uint8 RGB) and
pre-sampled validation patches, `next_training_batch` / `next_validation_batch` answering float32 numpy batches.  It can be
built from a directory of *.npy + *.png pairs like the reference, or straight from arrays (`Dataset.from_arrays`).

`DeviceDataset` keeps the same images in HBM instead (a 120-image, 12-megapixel training set is 7 GB of the 288) and cuts
every batch on the device: candidate corners from the device RNG, patch statistics, the discard policy and the crop +
normalisation are kernels (csrc/datafeed.hip), nothing crosses PCIe per step and nothing synchronises with the host, so the
feed can be queued ahead of the training step on the same stream.  It answers DeviceArrays with the interface of the host
class, and `sampler='host'` reproduces the reference's numpy-RNG stream exactly (host `sample_patch`, device crop).
"""
import os

import numpy as np
import torch

from ..device import DeviceArray, default_device
from .. import ops
from . import loading
from .loading import sample_patch


class Dataset(object):

    def __init__(self, data_directory, *, randomize=2468, load='xy', n_images=120, v_images=30, val_rgb_patch_size=128,
                 val_n_patches=1, val_discard='flat-aggressive'):
        if not any(load == allowed for allowed in ['xy', 'x', 'y']):
            raise ValueError('Invalid X/Y data requested!')
        if not os.path.isdir(data_directory):
            if '/' in data_directory or '\\' in data_directory:
                raise ValueError('Cannot find the data directory: {}'.format(data_directory))
            if os.path.isdir(os.path.join('data/raw/training_data/', data_directory)):
                data_directory = os.path.join('data/raw/training_data/', data_directory)
            elif os.path.isdir(os.path.join('data/rgb/', data_directory)):
                data_directory = os.path.join('data/rgb/', data_directory)
            else:
                raise ValueError('Cannot find the data directory: {}'.format(data_directory))
        self.files = {}
        self._loaded_data = load
        self._data_directory = data_directory
        self._counts = (n_images, v_images, val_n_patches)
        self._val_discard = 'flat-aggressive'           # sic: the reference reports this label whatever was asked for
        self.files['training'], self.files['validation'] = loading.discover_images(
            data_directory, randomize=randomize, n_images=n_images, v_images=v_images)
        self.data = {
            'training': loading.load_images(self.files['training'], data_directory, load=load),
            'validation': loading.load_patches(self.files['validation'], data_directory,
                                               patch_size=val_rgb_patch_size // 2, n_patches=val_n_patches, load=load,
                                               discard=val_discard),
        }
        self._set_resolution()

    @classmethod
    def from_arrays(cls, training, validation, load=None, name='arrays'):
        """training: {'x': (n, H/2, W/2, 4) uint16, 'y': (n, H, W, 3) uint8} full-resolution images, validation: the same
        keys holding patches - what the directory loader would have produced."""
        self = cls.__new__(cls)
        load = load or ''.join(k for k in 'xy' if k in training)
        if load not in ('xy', 'x', 'y'):
            raise ValueError('Invalid X/Y data requested!')
        for split in (training, validation):
            for k in load:
                want = np.uint16 if k == 'x' else np.uint8
                if split[k].dtype != want or split[k].ndim != 4:
                    raise ValueError('{} data must be a 4-D {} array'.format('RAW' if k == 'x' else 'RGB', want.__name__))
        n = len(training[load[0]])
        self.files = {'training': ['{:04d}'.format(i) for i in range(n)],
                      'validation': ['{:04d}'.format(i) for i in range(len(validation[load[0]]))]}
        self._loaded_data, self._data_directory = load, name
        self._counts = (n, len(validation[load[0]]), 1)
        self._val_discard = 'flat-aggressive'
        self.data = {'training': {k: training[k] for k in load}, 'validation': {k: validation[k] for k in load}}
        self._set_resolution()
        return self

    def _set_resolution(self):
        if 'y' in self.data['training']:
            self.H, self.W = self.data['training']['y'].shape[1:3]
        else:
            self.H, self.W = (2 * dim for dim in self.data['training']['x'].shape[1:3])

    def __getitem__(self, key):
        if key in ['training', 'validation']:
            return self.data[key]
        raise KeyError('Key: {} not found!'.format(key))

    def _check_batch(self, batch_id, batch_size, discard):
        if discard is not None and 'y' not in self.data['training']:
            raise ValueError('Cannot discard patches if RGB data is not loaded.')
        if (batch_id + 1) * batch_size > len(self.files['training']):
            raise ValueError('Not enough images for the requested batch_id & batch_size')

    def _pack(self, x, y):
        return (x, y) if self._loaded_data == 'xy' else (y if self._loaded_data == 'y' else x)

    def next_training_batch(self, batch_id, batch_size, rgb_patch_size, discard='flat', max_attempts=25):
        """One random patch of each of the batch's images (dataset.py:89-131) -> (RAW, RGB) float32 arrays in [0, 1]."""
        self._check_batch(batch_id, batch_size, discard)
        ps = rgb_patch_size // 2
        x = np.zeros((batch_size, ps, ps, 4), dtype=np.float32) if 'x' in self._loaded_data else None
        y = np.zeros((batch_size, rgb_patch_size, rgb_patch_size, 3), dtype=np.float32) if 'y' in self._loaded_data else None
        for b in range(batch_size):
            bid = batch_id * batch_size + b
            current_rgb = self.data['training']['y'][bid] if 'y' in self._loaded_data else \
                np.zeros((self.H, self.W, 3), np.uint8)
            xx, yy = sample_patch(current_rgb, rgb_patch_size, discard, max_attempts)
            rx, ry = xx // 2, yy // 2
            if x is not None:
                x[b] = self.data['training']['x'][bid][ry:ry + ps, rx:rx + ps].astype(np.float64) / (2 ** 16 - 1)
            if y is not None:
                y[b] = current_rgb[yy:yy + rgb_patch_size, xx:xx + rgb_patch_size].astype(np.float64) / (2 ** 8 - 1)
        return self._pack(x, y)

    def next_validation_batch(self, batch_id, batch_size):
        sl = slice(batch_id * batch_size, (batch_id + 1) * batch_size)
        x = (self.data['validation']['x'][sl].astype(np.float64) / (2 ** 16 - 1)).astype(np.float32) \
            if 'x' in self._loaded_data else None
        y = (self.data['validation']['y'][sl].astype(np.float64) / (2 ** 8 - 1)).astype(np.float32) \
            if 'y' in self._loaded_data else None
        return self._pack(x, y)

    def is_raw_and_rgb(self):
        return len(self._loaded_data) == 2

    @property
    def rgb_patch_size(self):
        if 'y' in self._loaded_data:
            return self.data['validation']['y'].shape[1]
        return 2 * self.data['validation']['x'].shape[1]

    @property
    def count_training(self):
        return self.data['training'][self._loaded_data[0]].shape[0]

    @property
    def count_validation(self):
        return self.data['validation'][self._loaded_data[0]].shape[0]

    @property
    def loaded_data(self):
        return {'xy': 'raw+rgb', 'y': 'rgb', 'x': 'raw'}[self._loaded_data]

    def shapes(self):
        stats = {'path': self._data_directory}
        for k in self._loaded_data:
            stats['training/{}'.format(k)] = self.data['training'][k].shape
            stats['validation/{}'.format(k)] = self.data['validation'][k].shape
        return stats

    def __repr__(self):
        return 'Dataset("{}", load="{}", n_images={}, v_images={}, val_rgb_patch_size={}, discard="{}")'.format(
            self._data_directory, self._loaded_data, self._counts[0], self._counts[1], self.rgb_patch_size,
            self._val_discard)

    def summary(self):
        valid_label = '' if self._val_discard is None else ', {}'.format(self._val_discard)
        return 'Dataset[{},{}] : {} train. images + {} valid. patches ({} px{})'.format(
            os.path.split(self._data_directory)[-1], self.loaded_data, self.count_training, self.count_validation,
            self.rgb_patch_size, valid_label)

    def details(self):
        label = [self.summary()]
        for k, l in zip('xy', ['RAW', 'RGB']):
            if k in self._loaded_data:
                label.append('{} -> training {} + validation {}'.format(l, self.data['training'][k].shape,
                                                                        self.data['validation'][k].shape))
        return '\n'.join(label)

    def get_training_generator(self, batch_size, rgb_patch_size, discard='flat'):
        for batch_id in range(self.count_training // batch_size):
            yield self.next_training_batch(batch_id, batch_size, rgb_patch_size, discard)

    def get_validation_generator(self, batch_size):
        for batch_id in range(self.count_validation // batch_size):
            yield self.next_validation_batch(batch_id, batch_size)


class DeviceDataset(object):
    """The images of a `Dataset` resident in HBM, batches cut by kernels.  Same calls as the host class; results are
    DeviceArrays (`.numpy()` for the host view, `.t` is the device tensor the models take without a copy)."""

    EXTRA_FLAT_CANDIDATES = 8        # coin flips 'flat' may lose before the candidate list runs out (probability 2^-8)

    def __init__(self, dataset, device=None, seed=2468, sampler='device'):
        if sampler not in ('device', 'host'):
            raise ValueError('sampler: device | host')
        self.host = dataset
        self.device = torch.device(device) if device is not None else default_device()
        if self.device.type != 'cuda':
            raise RuntimeError('DeviceDataset needs a GPU (the host path is helpers.dataset.Dataset)')
        self.sampler = sampler
        self._loaded_data = dataset._loaded_data
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed))
        self._dev = {}
        for split in ('training', 'validation'):
            self._dev[split] = {}
            for k in self._loaded_data:
                a = np.ascontiguousarray(dataset.data[split][k])
                t = torch.from_numpy(a.view(np.int16) if k == 'x' else a)      # uint16 bits travel as int16
                self._dev[split][k] = t.to(self.device)
        self.H, self.W = dataset.H, dataset.W
        self.last_xy = None

    def __getattr__(self, name):                           # count_*, rgb_patch_size, summary, ... : the host object's
        if name in ('host', '_dev'):
            raise AttributeError(name)
        return getattr(self.host, name)

    def _pack(self, x, y):
        x = None if x is None else DeviceArray(x)
        y = None if y is None else DeviceArray(y)
        return (x, y) if self._loaded_data == 'xy' else (y if self._loaded_data == 'y' else x)

    def candidates(self, batch_size, rgb_patch_size, attempts):
        """(B, A, 2) int32 even patch corners + (B, A) float32 uniforms from the device generator."""
        max_x, max_y = self.W - rgb_patch_size, self.H - rgb_patch_size
        shape = (batch_size, attempts)
        draw = lambda hi: (torch.randint(0, hi, shape, device=self.device, generator=self._gen, dtype=torch.int32) // 2) * 2 \
            if hi > 0 else torch.zeros(shape, dtype=torch.int32, device=self.device)
        cand = torch.stack((draw(max_x), draw(max_y)), dim=2).contiguous()
        uni = torch.rand(shape, device=self.device, generator=self._gen, dtype=torch.float32)
        return cand, uni

    def next_training_batch(self, batch_id, batch_size, rgb_patch_size, discard='flat', max_attempts=25, candidates=None,
                            uniforms=None):
        """The device counterpart of dataset.py:89-131.  `candidates` (B, A, 2) / `uniforms` (B, A) override the device
        RNG (parity tests); `last_xy` keeps the chosen corners on the device."""
        self.host._check_batch(batch_id, batch_size, discard)
        if discard not in ops.DISCARD_MODES:
            raise ValueError('Unrecognized discard mode: {}'.format(discard))
        if rgb_patch_size % 2 or rgb_patch_size > min(self.H, self.W):
            raise ValueError('rgb_patch_size must be even and fit the images')
        dt = self._dev['training']
        idx = torch.arange(batch_id * batch_size, (batch_id + 1) * batch_size, dtype=torch.int32, device=self.device)
        if self.sampler == 'host' and candidates is None:
            rgb_host = self.host.data['training'].get('y')
            xy = [sample_patch(rgb_host[int(i)] if rgb_host is not None else np.zeros((self.H, self.W, 3), np.uint8),
                               rgb_patch_size, discard, max_attempts) for i in idx.tolist()]
            xy = torch.tensor(xy, dtype=torch.int32, device=self.device)
        else:
            if candidates is None:
                attempts = 1 if discard is None else max_attempts + (self.EXTRA_FLAT_CANDIDATES if discard == 'flat' else 0)
                candidates, uniforms = self.candidates(batch_size, rgb_patch_size, attempts)
            var = mean = None
            if discard is not None:
                var, mean = ops.patch_stats(dt['y'], idx, candidates, rgb_patch_size)
            xy, _ = ops.patch_select(candidates, uniforms, var, mean, discard, max_attempts)
        self.last_xy = xy
        x, y = ops.patch_gather(dt.get('x'), dt.get('y'), idx, xy, rgb_patch_size)
        return self._pack(x, y)

    def next_validation_batch(self, batch_id, batch_size):
        dv = self._dev['validation']
        n = self.host.count_validation
        if (batch_id + 1) * batch_size > n:
            raise ValueError('Not enough patches for the requested batch_id & batch_size')
        idx = torch.arange(batch_id * batch_size, (batch_id + 1) * batch_size, dtype=torch.int32, device=self.device)
        xy = torch.zeros((batch_size, 2), dtype=torch.int32, device=self.device)
        x, y = ops.patch_gather(dv.get('x'), dv.get('y'), idx, xy, self.host.rgb_patch_size)
        return self._pack(x, y)

    def get_training_generator(self, batch_size, rgb_patch_size, discard='flat'):
        for batch_id in range(self.host.count_training // batch_size):
            yield self.next_training_batch(batch_id, batch_size, rgb_patch_size, discard)

    def get_validation_generator(self, batch_size):
        for batch_id in range(self.host.count_validation // batch_size):
            yield self.next_validation_batch(batch_id, batch_size)
