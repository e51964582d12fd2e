"""
Hyper-parameter specification / validation with the reference's semantics (helpers/paramspec.py:33-178):
specs are {name: (default, dtype, validator)} with validators range-tuple | enum-set | substring | callable;
update(**kw) casts, validates and raises ValueError on unknown or invalid values; values cannot be set directly.
"""
import types

import numpy as np

from . import utils


def numbers_in_range(dtype, min_value=None, max_value=None):
    """Validator for tuple-valued parameters (helpers/paramspec.py:20-30 in the reference)."""
    def check(items):
        return all(isinstance(i, dtype) and (min_value is None or i >= min_value) and
                   (max_value is None or i <= max_value) for i in items)
    return check


class ParamSpec(object):

    def __init__(self, specs):
        self._validate_specs(specs)
        self.__dict__['_specs'] = dict(specs)
        self.__dict__['_values'] = {}

    @staticmethod
    def _validate_specs(specs):
        for key, spec in specs.items():
            if type(spec) is not tuple or len(spec) != 3:
                raise ValueError('Invalid parameter specification for key {} - expected tuple of length 3'.format(key))
            if spec[2] is None:
                continue
            if spec[1] is str and not any(type(spec[2]) is s for s in [str, set, types.FunctionType]):
                raise ValueError('String data types can be validated by a regex (string), enum (set) or custom function')
            if utils.is_numeric_type(spec[1]) and not any(type(spec[2]) is s for s in [tuple, set]):
                raise ValueError('Numeric data types can be validated by a range (2-elem tuple), or enum (set)')

    def add(self, specs):
        self._validate_specs(specs)
        self._specs.update(specs)

    def __getattr__(self, name):
        if name in self.__dict__['_values']:
            return self.__dict__['_values'][name]
        if name in self.__dict__['_specs']:
            return self.__dict__['_specs'][name][0]
        raise KeyError(name)

    def __setattr__(self, key, value):
        raise ValueError('Values cannot be set directly. Use the `update` method.')

    def get_dtype(self, name):
        return self._specs[name][1]

    def get_default(self, name):
        return self._specs[name][0]

    def get_value(self, name):
        return self.__getattr__(name)

    def __repr__(self):
        return '{}({})'.format(type(self).__name__, self.to_dict())

    def to_dict(self):
        params = {key: spec[0] for key, spec in self._specs.items()}
        params.update(self._values)
        return params

    def to_json(self):
        return {k: v if utils.is_number(v) else str(v) for k, v in self.to_dict().items()}

    def __contains__(self, item):
        return item in self._specs

    def keys(self):
        return list(self._specs.keys())

    def changed_params(self):
        return {key: value for key, value in self._values.items() if self._specs[key][0] != value}

    def update(self, **params):
        for key, value in params.items():
            if key not in self._specs:
                raise ValueError('Unexpected parameter: {}!'.format(key))
            _, dtype, validation = self._specs[key]
            if value is None:
                continue
            if utils.is_number(value) and np.isnan(value):
                raise ValueError('Invalid value {} for attribute {}'.format(value, key))
            candidate = value if dtype is None else dtype(value)
            if validation is not None:
                if type(validation) == tuple and len(validation) == 2:
                    if validation[0] is not None and candidate < validation[0]:
                        raise ValueError('{}: {} fails minimum validation check >= {}!'.format(key, candidate, validation[0]))
                    if validation[1] is not None and candidate > validation[1]:
                        raise ValueError('{}: {} fails maximum validation check (<= {})!'.format(key, candidate, validation[1]))
                if type(validation) == set and candidate not in validation:
                    raise ValueError('{}: {} is not an allowed value ({})!'.format(key, candidate, validation))
                if type(validation) == str and dtype == str and validation not in candidate:
                    raise ValueError('{}: {} does not match regex ({})!'.format(key, candidate, validation))
                if callable(validation) and not validation(candidate):
                    raise ValueError('{}: {} failed custom validation check!'.format(key, candidate))
            self._values[key] = candidate
