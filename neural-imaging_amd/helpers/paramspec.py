"""
Hyper-parameter records of the models: a schema {name: (default, type, rule)} plus the values that differ from it.

Same contract as the reference's helpers/paramspec.py:33-178 (callers read attributes, call `update`, `to_dict`,
`to_json`, `changed_params`, `keys`, `add`, `get_dtype`, `get_default`, `get_value`, `get_min` / `get_max` / `get_enum` /
`get_regex`; assignment is refused; every failure is
a ValueError).  A rule is None, a (low, high) range with open ends as None, a set of allowed values, a substring a string
value must contain, or a predicate.  `update` casts each value to the field's type first; None leaves a field as it is.
"""
import math
import numbers
import types

from . import utils


def numbers_in_range(dtype, min_value=None, max_value=None):
    """Rule for tuple-valued fields: every item is a `dtype` inside [min_value, max_value] (paramspec.py:20-30)."""
    def inside(item):
        if not isinstance(item, dtype):
            return False
        return (min_value is None or item >= min_value) and (max_value is None or item <= max_value)
    return lambda items: all(inside(i) for i in items)


def item_passes(check):
    """Rule for tuple-valued fields: `check` holds for every item (paramspec.py:11-17)."""
    return lambda items: all(check(i) for i in items)


class _Field(object):
    """One schema entry with its rule compiled into a list of (test, message) pairs."""
    __slots__ = ('default', 'dtype', 'rule', 'tests')

    def __init__(self, name, entry):
        if not isinstance(entry, tuple) or len(entry) != 3:
            raise ValueError('Invalid parameter specification for key {} - expected tuple of length 3'.format(name))
        self.default, self.dtype, self.rule = entry
        rule, kind = self.rule, type(self.rule)
        if rule is not None:
            if self.dtype is str and kind not in (str, set, types.FunctionType):
                raise ValueError('String data types can be validated by a regex (string), enum (set) or custom function')
            if utils.is_numeric_type(self.dtype) and kind not in (tuple, set):
                raise ValueError('Numeric data types can be validated by a range (2-elem tuple), or enum (set)')
        self.tests = []
        if kind is tuple and len(rule) == 2:
            low, high = rule
            if low is not None:
                self.tests.append((lambda v: v >= low, 'fails minimum validation check >= {}!'.format(low)))
            if high is not None:
                self.tests.append((lambda v: v <= high, 'fails maximum validation check (<= {})!'.format(high)))
        elif kind is set:
            self.tests.append((lambda v: v in rule, 'is not an allowed value ({})!'.format(rule)))
        elif kind is str:
            if self.dtype is str:
                self.tests.append((lambda v: rule in v, 'does not match regex ({})!'.format(rule)))
        elif callable(rule):
            self.tests.append((rule, 'failed custom validation check!'))

    def accept(self, name, value):
        if isinstance(value, numbers.Real) and not isinstance(value, bool) and math.isnan(value):
            raise ValueError('Invalid value {} for attribute {}'.format(value, name))
        candidate = self.dtype(value) if self.dtype is not None else value
        for test, message in self.tests:
            if not test(candidate):
                raise ValueError('{}: {} {}'.format(name, candidate, message))
        return candidate


class ParamSpec(object):

    def __init__(self, specs):
        object.__setattr__(self, '_fields', {name: _Field(name, entry) for name, entry in specs.items()})
        object.__setattr__(self, '_values', {})

    # -- schema -------------------------------------------------------------------------------------------------------
    def add(self, specs):
        fresh = {name: _Field(name, entry) for name, entry in specs.items()}     # validated before anything is touched
        self._fields.update(fresh)

    def keys(self):
        return list(self._fields)

    def __contains__(self, name):
        return name in self._fields

    def get_dtype(self, name):
        return self._fields[name].dtype

    def get_default(self, name):
        return self._fields[name].default

    def _rule_of_kind(self, name, kind):
        rule = self._fields[name].rule
        return rule if type(rule) is kind and (kind is not tuple or len(rule) == 2) else None

    def get_min(self, name):                             # paramspec.py:75-103: what a rule of the matching kind says, else None
        rule = self._rule_of_kind(name, tuple)
        return None if rule is None else rule[0]

    def get_max(self, name):
        rule = self._rule_of_kind(name, tuple)
        return None if rule is None else rule[1]

    def get_enum(self, name):
        rule = self._rule_of_kind(name, set)
        return None if rule is None else set(rule)

    def get_regex(self, name):
        return self._rule_of_kind(name, str)

    # -- values -------------------------------------------------------------------------------------------------------
    def __getattr__(self, name):
        state = object.__getattribute__(self, '__dict__')
        if name in state.get('_values', ()):
            return state['_values'][name]
        if name in state.get('_fields', ()):
            return state['_fields'][name].default
        raise KeyError(name)

    def __setattr__(self, name, value):
        raise ValueError('Values cannot be set directly. Use the `update` method.')

    def get_value(self, name):
        return getattr(self, name)

    def update(self, **params):
        for name, value in params.items():
            field = self._fields.get(name)
            if field is None:
                raise ValueError('Unexpected parameter: {}!'.format(name))
            if value is not None:
                self._values[name] = field.accept(name, value)

    def to_dict(self):
        return {name: self._values.get(name, field.default) for name, field in self._fields.items()}

    def to_json(self):
        return {name: (value if utils.is_number(value) else str(value)) for name, value in self.to_dict().items()}

    def changed_params(self):
        return {name: value for name, value in self._values.items() if value != self._fields[name].default}

    def __repr__(self):
        return '{}({})'.format(type(self).__name__, self.to_dict())
