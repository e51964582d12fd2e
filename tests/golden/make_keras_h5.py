"""
Writes tests/golden/keras_weights_h5py.h5 (+ keras_weights_expected.npz): a weight file in the layout of
`tf.keras.Model.save_weights(path, save_format='h5')` (tf.keras 2.1 hdf5_format.save_weights_to_hdf5_group), produced
with the real HDF5 library through h5py so that the pure-Python reader in neural-imaging_amd/helpers/hdf5.py is pinned
against libhdf5's own bytes.  TensorFlow is not installed anywhere in this image, so the tree is laid out by hand the way
Keras does it: fixed-length `layer_names` / `weight_names` string arrays (Keras passes numpy 'S' arrays), variable-length
`backend` / `keras_version` scalars (h5py stores Python bytes that way), a layer without weights, a scalar weight, a
non-trainable weight listed after the trainable ones, and enough layers for several symbol-table nodes.

Run with the conda interpreter of this image (the system python has no h5py):
    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(20260927)
layers = [('input_1', []),
          ('conv2d', [('conv2d/kernel:0', (3, 3, 4, 8)), ('conv2d/bias:0', (8,))]),
          ('max_pooling2d', []),
          ('conv2d_transpose', [('conv2d_transpose/kernel:0', (2, 2, 4, 8)), ('conv2d_transpose/bias:0', (4,))]),
          ('demosaicing_layer', [('demosaicing_layer/alpha:0', ()), ('demosaicing_layer/conv2d_1/kernel:0', (5, 5, 3, 3))]),
          ('dense', [('dense/kernel:0', (16, 5)), ('dense/bias:0', (5,))])]
layers += [('conv2d_{}'.format(i), [('conv2d_{}/kernel:0'.format(i), (1, 1, i, 2)), ('conv2d_{}/bias:0'.format(i), (2,))])
           for i in range(2, 22)]
expected = {}
with h5py.File(os.path.join(HERE, 'keras_weights_h5py.h5'), 'w') as f:
    f.attrs['layer_names'] = np.asarray([n.encode('utf8') for n, _ in layers])
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['keras_version'] = '2.2.4-tf'.encode('utf8')
    for name, weights in layers:
        g = f.create_group(name)
        g.attrs['weight_names'] = np.asarray([w.encode('utf8') for w, _ in weights])
        for w, shape in weights:
            val = rng.normal(size=shape).astype(np.float32)
            d = g.create_dataset(w, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val
            expected[name + '|' + w.replace('/', '|')] = val
np.savez(os.path.join(HERE, 'keras_weights_expected.npz'), order=np.asarray([k for k in expected]), **expected)
print('wrote', len(expected), 'weights')
