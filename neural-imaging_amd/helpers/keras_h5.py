"""
Keras weight files (`model.save_weights(path, save_format='h5')`, the reference's checkpoint container:
models/tfmodel.py:150-182) through the pure-Python HDF5 codec in helpers/hdf5.py.

Layout restated from Keras' hdf5 saving code (tf.keras 2.1 `save_weights_to_hdf5_group` / `load_weights_from_hdf5_group`):
  root attrs   layer_names = [b'<layer>' ...] (split into layer_names0, layer_names1, ... when larger than 64 kB),
               backend = b'tensorflow', keras_version = b'2.2.4-tf'
  /<layer>     attrs weight_names = [b'<layer>/kernel:0', b'<layer>/bias:0', ...]  (trainable, then non-trainable)
  /<layer>/<weight name>   one dataset per weight, native shape (Conv2D HWIO, Conv2DTranspose (kh,kw,Cout,Cin),
               Dense (in,out), scalars ()), float32
Loading follows Keras' default (not by_name) rule: layers without weights are dropped and the remaining weights are taken
in file order - the auto-generated layer names (conv2d_17, ...) depend on how many layers the saving process had created
before and carry no meaning.
"""
import numpy as np

from . import hdf5

KERAS_VERSION = b'2.2.4-tf'
_CHUNK = 64512                      # HDF5_OBJECT_HEADER_LIMIT of the Keras writer


def _as_names(values):
    out = []
    for v in np.asarray(values).reshape(-1):
        out.append(v.decode('utf8') if isinstance(v, (bytes, np.bytes_)) else str(v))
    return out


def _load_attr_list(attrs, name):
    """Keras `load_attributes_from_hdf5_group`: the attribute itself, or its chunks name0, name1, ..."""
    if name in attrs:
        a = attrs[name]
        return [] if np.asarray(a).size == 0 else _as_names(a)
    out, k = [], 0
    while '{}{}'.format(name, k) in attrs:
        out += _as_names(attrs['{}{}'.format(name, k)])
        k += 1
    if k == 0:
        raise KeyError('attribute {} not found - not a Keras weight file?'.format(name))
    return out


def load_weights(path):
    """-> [(layer name, [(weight name, ndarray), ...]), ...] in the file's layer / weight order (all layers listed)."""
    root = hdf5.read_hdf5(path)
    if 'layer_names' not in root.attrs and 'layer_names0' not in root.attrs and 'model_weights' in root:
        root = root['model_weights']                 # a full `model.save()` file keeps the same tree one level down
    layers = []
    for lname in _load_attr_list(root.attrs, 'layer_names'):
        g = root[lname]
        weights = [(wname, np.asarray(g[wname].value)) for wname in _load_attr_list(g.attrs, 'weight_names')]
        layers.append((lname, weights))
    return layers


def _set_attr_list(attrs, name, values):
    data = [v.encode('utf8') if isinstance(v, str) else v for v in values]
    width = max([len(v) for v in data] + [1])
    if width * len(data) <= _CHUNK:
        attrs[name] = np.asarray(data, dtype='S{}'.format(width)) if data else np.zeros((0,), np.float64)
        return
    per = max(1, _CHUNK // width)
    for k, s in enumerate(range(0, len(data), per)):
        attrs['{}{}'.format(name, k)] = np.asarray(data[s:s + per], dtype='S{}'.format(width))


def save_weights(path, layers):
    """layers: [(layer name, [(weight name, ndarray), ...]), ...]; weight names as Keras gives them
    ('<layer>/kernel:0'), '/' in them nests groups exactly like h5py's create_dataset does."""
    root = hdf5.Group()
    _set_attr_list(root.attrs, 'layer_names', [l for l, _ in layers])
    root.attrs['backend'] = b'tensorflow'
    root.attrs['keras_version'] = KERAS_VERSION
    for lname, weights in layers:
        if '/' in lname:
            raise ValueError('Keras layer names cannot contain "/": {}'.format(lname))
        g = hdf5.Group()
        _set_attr_list(g.attrs, 'weight_names', [w for w, _ in weights])
        for wname, value in weights:
            node = g
            parts = wname.split('/')
            for part in parts[:-1]:
                if part not in node:
                    node[part] = hdf5.Group()
                node = node[part]
            node[parts[-1]] = hdf5.Dataset(np.asarray(value))
        root[lname] = g
    hdf5.write_hdf5(path, root)
