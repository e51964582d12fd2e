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


_SEARCH_ROOTS = ('data/raw/training_data', 'data/rgb')            # where a bare data-set name is looked up (dataset.py:60-70)
_FULL_SCALE = {'x': float(2 ** 16 - 1), 'y': float(2 ** 8 - 1)}    # uint16 RAW stacks, uint8 RGB
_KINDS = {'xy': 'raw+rgb', 'y': 'rgb', 'x': 'raw'}


def _find_directory(name):
    """A path is taken as it is; a bare name is searched under the repository's data roots.  ValueError if nothing is there."""
    if os.path.isdir(name):
        return name
    if not any(sep in name for sep in '/\\'):
        for root in _SEARCH_ROOTS:
            if os.path.isdir(os.path.join(root, name)):
                return os.path.join(root, name)
    raise ValueError('Cannot find the data directory: {}'.format(name))


def _unit_range(kind, a):
    """Integer samples -> float32 in [0, 1]: the float64 quotient rounded once (bit-exact with the reference's numpy)."""
    return (a.astype(np.float64) / _FULL_SCALE[kind]).astype(np.float32)


class _Pipeline(object):
    """Iterable over the batches a generator factory yields; iterating again starts again."""

    def __init__(self, factory):
        self._factory = factory

    def __iter__(self):
        return iter(self._factory())


class Dataset(object):
    """Host-side training data with the reference's interface (helpers/dataset.py:50-189): `data[split][kind]` holds the
    full-resolution training images and the pre-cut validation patches, kind 'x' = RAW (n, H/2, W/2, 4) uint16, 'y' = RGB
    (n, H, W, 3) uint8.  The container is filled either from a directory (loading.discover_images / load_images /
    load_patches) or from arrays; everything else works on the arrays alone."""

    # the label summary() / repr() print for the validation patches: the reference reports this one whatever policy the caller
    # asked for (dataset.py:57), and scripts parse the summary line
    VAL_DISCARD_LABEL = 'flat-aggressive'

    def __init__(self, data_directory, *, randomize=2468, load='xy', n_images=120, v_images=30, val_rgb_patch_size=128,
                 val_n_patches=1, val_discard='flat-aggressive'):
        if load not in _KINDS:
            raise ValueError('Invalid X/Y data requested!')
        root = _find_directory(data_directory)
        train_files, val_files = loading.discover_images(root, randomize=randomize, n_images=n_images, v_images=v_images)
        self._fill(root, load, {'training': train_files, 'validation': val_files},
                   loading.load_images(train_files, root, load=load),
                   loading.load_patches(val_files, root, patch_size=val_rgb_patch_size // 2, n_patches=val_n_patches,
                                        load=load, discard=val_discard),
                   (n_images, v_images, val_n_patches))

    @classmethod
    def from_arrays(cls, training, validation, load=None, name='arrays'):
        """training: {'x': (n, H/2, W/2, 4) uint16, 'y': (n, H, W, 3) uint8} full-resolution images, validation: the same
        keys holding patches - what the directory loader would have produced."""
        load = load or ''.join(k for k in 'xy' if k in training)
        if load not in _KINDS:
            raise ValueError('Invalid X/Y data requested!')
        for split in (training, validation):
            for k in load:
                want = np.uint16 if k == 'x' else np.uint8
                if split[k].dtype != want or split[k].ndim != 4:
                    raise ValueError('{} data must be a 4-D {} array'.format('RAW' if k == 'x' else 'RGB', want.__name__))
        n, v = len(training[load[0]]), len(validation[load[0]])
        self = cls.__new__(cls)
        self._fill(name, load, {'training': ['{:04d}'.format(i) for i in range(n)],
                                'validation': ['{:04d}'.format(i) for i in range(v)]},
                   {k: training[k] for k in load}, {k: validation[k] for k in load}, (n, v, 1))
        return self

    def _fill(self, name, load, files, training, validation, counts):
        self._data_directory, self._loaded_data, self.files, self._counts = name, load, files, counts
        self._val_discard = self.VAL_DISCARD_LABEL
        self.data = {'training': training, 'validation': validation}
        t = training
        self.H, self.W = t['y'].shape[1:3] if 'y' in t else tuple(2 * d for d in t['x'].shape[1:3])

    def __getitem__(self, key):
        if key not in self.data:
            raise KeyError('Key: {} not found!'.format(key))
        return self.data[key]

    # ---- batches -------------------------------------------------------------------------------------------------------
    def _check_batch(self, batch_id, batch_size, discard):
        if discard is not None and 'y' not in self.data['training']:
            raise ValueError('Cannot discard patches if RGB data is not loaded.')
        if (batch_id + 1) * batch_size > len(self.files['training']):
            raise ValueError('Not enough images for the requested batch_id & batch_size')

    def _pack(self, x, y):
        return (x, y) if self._loaded_data == 'xy' else (y if self._loaded_data == 'y' else x)

    def _crop(self, split, kind, ids, corners, rgb_patch_size):
        """float32 patches of `kind` for images `ids` at RGB corners (xx, yy); RAW patches sit at half the coordinates."""
        if kind not in self._loaded_data:
            return None
        side, div = (rgb_patch_size, 1) if kind == 'y' else (rgb_patch_size // 2, 2)
        src = self.data[split][kind]
        cut = [src[i, yy // div:yy // div + side, xx // div:xx // div + side] for i, (xx, yy) in zip(ids, corners)]
        return _unit_range(kind, np.stack(cut))

    def next_training_batch(self, batch_id, batch_size, rgb_patch_size, discard='flat', max_attempts=25):
        """One random patch of each of the batch's images (dataset.py:89-131) -> (RAW, RGB) float32 arrays in [0, 1].  The
        corners come from loading.sample_patch, image by image in batch order (that fixes the numpy RNG stream the golden
        vectors replay); without RGB data the policy sees a black image, i.e. only discard=None is possible."""
        self._check_batch(batch_id, batch_size, discard)
        ids = range(batch_id * batch_size, (batch_id + 1) * batch_size)
        rgb = self.data['training'].get('y')
        blank = None if rgb is not None else np.zeros((self.H, self.W, 3), np.uint8)
        corners = [sample_patch(rgb[i] if rgb is not None else blank, rgb_patch_size, discard, max_attempts) for i in ids]
        return self._pack(self._crop('training', 'x', ids, corners, rgb_patch_size),
                          self._crop('training', 'y', ids, corners, rgb_patch_size))

    def next_validation_batch(self, batch_id, batch_size):
        sl = slice(batch_id * batch_size, (batch_id + 1) * batch_size)
        part = {k: _unit_range(k, self.data['validation'][k][sl]) for k in self._loaded_data}
        return self._pack(part.get('x'), part.get('y'))

    def get_training_generator(self, batch_size, rgb_patch_size, discard='flat'):
        for batch_id in range(self.count_training // batch_size):
            yield self.next_training_batch(batch_id, batch_size, rgb_patch_size, discard)

    def get_validation_generator(self, batch_size):
        for batch_id in range(self.count_validation // batch_size):
            yield self.next_validation_batch(batch_id, batch_size)

    def get_training_pipeline(self, batch_size, rgb_patch_size, discard='flat'):
        """A re-iterable stream of training batches - the role tf.data.Dataset.from_generator plays in the reference
        (helpers/dataset.py:247-255): every `for batch in pipeline` starts a fresh pass of get_training_generator."""
        return _Pipeline(lambda: self.get_training_generator(batch_size, rgb_patch_size, discard))

    def get_validation_pipeline(self, batch_size):
        return _Pipeline(lambda: self.get_validation_generator(batch_size))

    # ---- descriptions ----------------------------------------------------------------------------------------------------
    def is_raw_and_rgb(self):
        return self._loaded_data == 'xy'

    @property
    def rgb_patch_size(self):
        v = self.data['validation']
        return v['y'].shape[1] if 'y' in v else 2 * v['x'].shape[1]

    @property
    def count_training(self):
        return len(self.data['training'][self._loaded_data[0]])

    @property
    def count_validation(self):
        return len(self.data['validation'][self._loaded_data[0]])

    @property
    def loaded_data(self):
        return _KINDS[self._loaded_data]

    def shapes(self):
        out = {'path': self._data_directory}
        for k in self._loaded_data:
            for split in ('training', 'validation'):
                out['{}/{}'.format(split, k)] = self.data[split][k].shape
        return out

    def __repr__(self):
        return 'Dataset("{}", load="{}", n_images={}, v_images={}, val_rgb_patch_size={}, discard="{}")'.format(
            self._data_directory, self._loaded_data, self._counts[0], self._counts[1], self.rgb_patch_size,
            self._val_discard)

    def summary(self):
        return 'Dataset[{},{}] : {} train. images + {} valid. patches ({} px{})'.format(
            os.path.basename(self._data_directory), self.loaded_data, self.count_training, self.count_validation,
            self.rgb_patch_size, ', {}'.format(self._val_discard) if self._val_discard is not None else '')

    def details(self):
        lines = [self.summary()]
        lines += ['{} -> training {} + validation {}'.format({'x': 'RAW', 'y': 'RGB'}[k], self.data['training'][k].shape,
                                                             self.data['validation'][k].shape) for k in self._loaded_data]
        return '\n'.join(lines)


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

    def get_training_pipeline(self, batch_size, rgb_patch_size, discard='flat'):
        """A re-iterable stream of training batches - the role tf.data.Dataset.from_generator plays in the reference
        (helpers/dataset.py:247-255): every `for batch in pipeline` starts a fresh pass of get_training_generator."""
        return _Pipeline(lambda: self.get_training_generator(batch_size, rgb_patch_size, discard))

    def get_validation_pipeline(self, batch_size):
        return _Pipeline(lambda: self.get_validation_generator(batch_size))
