"""Small host-side helpers mirrored from the reference's helpers/utils.py (is_number :53-58, join_args)."""
import numpy as np

_numeric_types = {int, float, bool, np.bool_, np.float16, np.float32, np.float64, np.int8, np.int16, np.int32,
                  np.int64, np.uint8, np.uint16, np.uint32, np.uint64}


def is_number(value):
    return type(value) in _numeric_types


def is_numeric_type(t):
    return t in _numeric_types


def join_args(params):
    return ','.join('{}={}'.format(k, v) for k, v in params.items())


def format_patch_shape(shape):
    if shape is None:
        return '?'
    return '({})'.format(','.join('?' if s is None else str(s) for s in shape))
