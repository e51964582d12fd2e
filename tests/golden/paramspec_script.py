"""The scripted ParamSpec session behind the golden transcript (tests/golden/make_golden.py runs it on the reference's
helpers/paramspec.py, tests/test_golden.py::test_paramspec_transcript on the product's): a list of (label, thunk); a thunk's
outcome is its repr() or 'ValueError: <message>' / 'KeyError: <message>'."""


def _head(thunk, n):
    try:
        return thunk()
    except ValueError as e:
        raise ValueError(str(e)[:n])


def session(ps):
    """ps: the paramspec module under test (ParamSpec, numbers_in_range, item_passes)."""
    P = ps.ParamSpec
    h = P({'n': (5, int, (2, 6)), 'act': ('leaky_relu', str, {'leaky_relu', 'relu'}), 'x': (1.0, float, None),
           'tag': ('gbrg', str, 'g'), 'filters': ((), tuple, ps.numbers_in_range(int, 1, 1024)), 'lo': (0.5, float, (0, None)),
           'odd': ((1, 3), tuple, ps.item_passes(lambda v: v % 2 == 1)), 'free': (None, None, None)})
    steps = [
        ('defaults', lambda: (h.n, h.act, h.x, h.tag, h.filters, h.lo, h.odd, h.free)),
        ('keys', lambda: h.keys()),
        ('contains', lambda: ('n' in h, 'zz' in h)),
        ('update_cast', lambda: (h.update(n='3', x=2, filters=[32, 64]), h.n, h.x, h.filters)[1:]),
        ('changed', lambda: sorted(h.changed_params().items())),
        ('to_json', lambda: sorted(h.to_json().items(), key=lambda kv: kv[0])),
        ('to_dict', lambda: sorted(h.to_dict().items(), key=lambda kv: kv[0])),
        ('below_min', lambda: h.update(n=1)),
        ('above_max', lambda: h.update(n=7)),
        ('open_range_ok', lambda: (h.update(lo=1e9), h.lo)[1]),
        ('open_range_low', lambda: h.update(lo=-0.1)),
        ('enum_str', lambda: _head(lambda: h.update(act='gelu'), 35)),      # (the set's print order depends on the hash seed: keep the stable prefix only)
        ('substring_ok', lambda: (h.update(tag='rggb'), h.tag)[1]),
        ('substring_bad', lambda: h.update(tag='xyz')),
        ('custom_bad_type', lambda: h.update(filters=(32, 2.5))),
        ('custom_bad_range', lambda: h.update(filters=(0,))),
        ('item_passes_bad', lambda: h.update(odd=(1, 4))),
        ('item_passes_ok', lambda: (h.update(odd=(5, 7)), h.odd)[1]),
        ('unexpected', lambda: h.update(bogus=1)),
        ('nan', lambda: h.update(x=float('nan'))),
        ('none_keeps', lambda: (h.update(n=None), h.n)[1]),
        ('no_dtype', lambda: (h.update(free=[1, 'a']), h.free)[1]),
        ('setattr', lambda: setattr(h, 'n', 4)),
        ('getattr_unknown', lambda: h.nope),
        ('get_dtype', lambda: (h.get_dtype('n').__name__, h.get_dtype('free'))),
        ('get_default_value', lambda: (h.get_default('n'), h.get_value('n'))),
        ('get_min_max', lambda: (h.get_min('n'), h.get_max('n'), h.get_min('lo'), h.get_max('lo'), h.get_min('act'), h.get_max('x'))),
        ('get_enum', lambda: (sorted(h.get_enum('act')), h.get_enum('n'))),
        ('get_regex', lambda: (h.get_regex('tag'), h.get_regex('act'))),
        ('add', lambda: (h.add({'k': (3, int, {3, 5})}), h.k, 'k' in h)[1:]),
        ('enum_int', lambda: h.update(k=4)),
        ('add_bad_numeric_rule', lambda: h.add({'bad': (1, int, 'x')})),
        ('add_bad_string_rule', lambda: h.add({'bad': ('a', str, (1, 2))})),
        ('partial_update', lambda: h.update(k=5, n=9)),
        ('after_partial', lambda: (h.k, h.n)),
        ('repr_prefix', lambda: repr(h)[:12]),
    ]
    out = []
    for label, thunk in steps:
        try:
            out.append([label, repr(thunk())])
        except (ValueError, KeyError) as e:
            out.append([label, '{}: {}'.format(type(e).__name__, e)])
    return out
