"""Host-side helpers with the reference's names and results (helpers/utils.py:54-125, 189-198, 257-263): number / vector tests,
the number formats of the training summaries, nested dictionary access, the argument lists of a model's repr, patch-shape labels.
Pinned on the reference's own outputs (tests/golden/make_golden.py -> tests/test_golden.py::test_utils_strings)."""
import re
from functools import reduce

import numpy as np

_numeric_types = {int, float, bool, np.bool_, np.float16, np.float32, np.float64, np.int8, np.int16, np.int32,
                  np.int64, np.uint8, np.uint16, np.uint32, np.uint64}


def is_number(value):
    return type(value) in _numeric_types


def is_numeric_type(t):
    return t in _numeric_types


def is_nan(value):
    if value is None:
        return True
    if is_number(value):
        return bool(np.isnan(value))
    return False


def is_vector(data):
    if isinstance(data, list) and all(is_number(x) for x in data):
        return True
    return isinstance(data, np.ndarray) and data.ndim == 1


def format_number_order(n):
    """1234 -> '1k', 2.5e6 -> '2M' (thousands suffix, no decimals)."""
    n = float(n)
    suffix = ('', 'k', 'M', 'B', 'T')
    idx = max(0, min(len(suffix) - 1, int(np.floor(0 if n == 0 else np.log10(abs(n)) / 3))))
    return '{:.0f}{}'.format(n / 10 ** (3 * idx), suffix[idx])


def format_number(x, digits=3):
    """Floats with `digits` significant digits in fixed notation ('0.100', '0.000100', '12.3'), everything else as str()."""
    if np.isnan(x):
        return 'nan'
    if np.isinf(x):
        return '∞'
    try:
        if isinstance(x, float) and x != 0:
            order = int(np.floor(np.log10(np.abs(x))))
            w = max(0, order) + (digits - 1)
            p = max(0, -order) + (digits - 1)
            return '{:{w}.{p}f}'.format(x, w=w, p=p)
        return '{}'.format(x)
    except Exception:                                                     # noqa
        return '?'


def match_option(x, options, regexp=False):
    """The option a (possibly abbreviated) name means: the single regexp match; else the single option that starts with x (or that x
    starts with); else the closest one in edit distance."""
    if regexp:
        matches = [y for y in options if re.match(x, y)]
        if len(matches) == 1:
            return matches[0]
        raise ValueError('No regexp match: "{}" to any of {}!'.format(x, options))
    start_match = [y.startswith(x) or x.startswith(y) for y in options]
    if sum(start_match) == 1:
        return options[start_match.index(True)]
    distances = [_edit_distance(x, y) for y in options]
    return options[distances.index(min(distances))]


def _edit_distance(a, b):
    """Levenshtein distance (insert / delete / substitute, cost 1 each) - the reference takes it from the Levenshtein package."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def get(data, key, default=None, sep='.'):
    """data['a']['b'] for key 'a.b'; a missing level yields {} (as the reference's reduce over dict.get does), `default` only when
    a level is not a dictionary of the expected kind."""
    try:
        return reduce(lambda c, k: c.get(k, {}), key.split(sep), data)
    except KeyError:
        return default


def join_args(args, sep=','):
    return sep.join('{}={}'.format(k, '"{}"'.format(v) if isinstance(v, str) else v) for k, v in args.items())


def format_patch_shape(patch_size):
    if patch_size is None:
        return '?'
    if any(x is None for x in patch_size):
        return '(rgb)' if patch_size[-1] == 3 else '(raw)'
    return '×'.join(str(x) for x in patch_size)
