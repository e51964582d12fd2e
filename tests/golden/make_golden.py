#!/usr/bin/env python3
"""
Generate golden vectors by importing the *importable* parts of the reference
(pkorus/neural-imaging @ /root/reference). Runs ONLY in the build container:
the reference never travels to the GPU box, the .npz it writes does.

What can be imported: helpers/kernels.py, helpers/stats.py,
helpers/image.py, compression/jpeg_helpers.py (jpeg_qtable, zigzag,
jpeg_qf_estimation) after stubbing imageio / jpylyzer / skimage / loguru.
Everything that touches TensorFlow cannot be imported here (no TF wheel, no
network) - those paths are "parity unpinned" and say so in oracle/*.py.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/reference_tables.npz)
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_tables.npz')


def _stub_modules():
    for name in ['imageio', 'jpylyzer', 'skimage', 'skimage.measure', 'skimage.metrics',
                 'skimage.transform', 'loguru', 'Levenshtein']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['jpylyzer'].jpylyzer = types.SimpleNamespace()
    sys.modules['loguru'].logger = types.SimpleNamespace(
        info=print, warning=print, debug=print, error=print)
    sk = sys.modules['skimage.measure']
    sk.compare_ssim = sk.compare_psnr = sk.compare_mse = None
    skm = sys.modules['skimage.metrics']
    skm.structural_similarity = skm.peak_signal_noise_ratio = skm.mean_squared_error = None
    # scipy.signal.gaussian moved to scipy.signal.windows in the scipy installed here
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, 'gaussian'):
        scipy.signal.gaussian = scipy.signal.windows.gaussian
    # numpy aliases removed in numpy>=1.24 that helpers/utils.py:25-27 still names
    for alias, typ in (('float', float), ('bool', bool), ('int', int)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)


def main():
    if not os.path.isdir(REF):
        raise SystemExit('reference tree not present - golden vectors can only be generated in the build container')
    _stub_modules()
    sys.path.insert(0, REF)
    from helpers import kernels, stats, image
    from compression import jpeg_helpers

    g = {}
    # 1. IJG quantisation tables, QF 1..100, luma/chroma  (jpeg_helpers.py:264-305) - bit-exact contract
    g['qtables'] = np.stack([np.stack([jpeg_helpers.jpeg_qtable(q, ch) for ch in (0, 1)])
                             for q in range(1, 101)]).astype(np.uint8)            # (100,2,8,8)
    # 2. zig-zag index matrix (jpeg_helpers.py:253-261)
    g['zigzag8'] = jpeg_helpers.zigzag(8).astype(np.uint16)
    # qf estimation round trip (jpeg_helpers.py:308-310)
    g['qf_est_luma'] = np.array([jpeg_helpers.jpeg_qf_estimation(jpeg_helpers.jpeg_qtable(q, 0), 0)
                                 for q in range(1, 101)], dtype=np.int32)
    # 3. constant kernels (helpers/kernels.py)
    for cfa in ('gbrg', 'rggb', 'bggr'):
        g['upk_' + cfa] = kernels.upsampling_kernel(cfa).astype(np.float64)
    for k in (3, 5, 7, 9, 11):                                     # INet / ClassicISP `kernel` (models/pipelines.py:242,419)
        g['bilin%d' % k] = kernels.bilin_kernel(k).astype(np.float64)
    d1k, d1b, d2k, d2b = kernels.gamma_kernels()
    g['gamma_d1k'], g['gamma_d1b'], g['gamma_d2k'], g['gamma_d2b'] = d1k, d1b, d2k, d2b
    for std in (0.5, 0.83, 1.0, 3.0, 7.0):
        g['gkern5_%g' % std] = kernels.gkern(5, std)
    f = np.array([[0, 0, 0, 0, 0], [0, -1, -2, -1, 0], [0, -2, 12, -2, 0], [0, -1, -2, -1, 0], [0, 0, 0, 0, 0]])
    g['fan_residual_init'] = kernels.repeat_2dfilter(f, 3)                         # models/layers.py:40-43
    g['center_mask_5_3'] = kernels.center_mask_2dfilter(5, 3)
    gk = np.array([[-0.0833, -0.1667, -0.0833], [-0.1667, 0, -0.1667], [-0.0833, -0.1667, -0.0833]])
    g['repeat_sharpen_base'] = kernels.repeat_2dfilter(gk, 3)
    # 4. hard entropy / histogram helpers on seeded latents (helpers/stats.py:107-138) - cross-check for the
    #    soft entropy estimator (tf_helpers.py:290-333), which cannot be imported (TF).
    rng = np.random.RandomState(1234)
    code_book = np.arange(-15, 17).astype(np.float64)
    lat = np.clip(np.round(rng.normal(0, 3.0, size=(4, 8, 8, 32))), -15, 16)
    g['latent_seeded'] = lat
    g['latent_codebook'] = code_book
    g['latent_hist'] = stats.hist(lat, code_book).astype(np.int64)
    g['latent_entropy'] = np.array(stats.entropy(lat, code_book))
    g['latent_bin_edges'] = stats.bin_edges(code_book)
    # 5. batch_gamma (helpers/image.py:22-28)
    xb = rng.uniform(0, 1, size=(3, 4, 4, 3)).astype(np.float32)
    g['batch_gamma_in'] = xb
    g['batch_gamma_out'] = image.batch_gamma(xb, 2.2)

    # 6. helpers/utils.py: number formats, labels, argument lists (strings; np.int is shimmed above as in the reference's numpy)
    from helpers import utils
    import json
    from collections import OrderedDict
    numbers = [0.1, 0.0001, 12.345, 1234.5678, 0.05, 1e-4, 250.0, 1.0, 0.0, 1001, 64, float('nan'), float('inf'), -0.25, 3]
    orders = [0, 7, 999, 1000, 1234, 2.5e6, 7763820, 3e9, 4e12, 5e15, -12000]
    shapes = [None, (64, 64, 4), (128, 128, 3), (None, None, 3), (None, None, 4), (1, 2)]
    args = [OrderedDict(), OrderedDict([('quality', 80), ('codec', 'soft'), ('trainable', False)]),
            OrderedDict([('kernel', 3), ('c_filters', (32, 32)), ('cfa_pattern', 'rggb'), ('alpha', 0.5)])]
    nested = {'nip': {'performance': {'psnr': {'validation': [30.0, 31.5]}}}, 'x': 1}
    keys = ['nip.performance.psnr.validation', 'nip.performance.ssim', 'missing.level', 'x']
    opts = ['sharpen', 'resample', 'gaussian', 'jpeg', 'awgn', 'gamma', 'median']
    g['utils_json'] = np.array(json.dumps({
        'format_number_in': [repr(v) for v in numbers], 'format_number': [utils.format_number(v) for v in numbers],
        'format_number_d2': [utils.format_number(v, 2) for v in numbers[:6]],
        'format_number_order_in': orders, 'format_number_order': [utils.format_number_order(v) for v in orders],
        'format_patch_shape_in': shapes, 'format_patch_shape': [utils.format_patch_shape(v) for v in shapes],
        'join_args_in': [list(a.items()) for a in args], 'join_args': [utils.join_args(a) for a in args],
        'get_keys': keys, 'get': [utils.get(nested, k, 'dflt') for k in keys],
        'is_nan': [bool(utils.is_nan(v)) for v in (None, float('nan'), 1.0, 'x', np.float32('nan'))],
        'is_vector': [bool(utils.is_vector(v)) for v in ([1, 2.0], [1, 'a'], np.zeros(3), np.zeros((2, 2)), (1, 2))],
        'match_option_in': ['sharp', 'gaus', 'resamp', 'jpeg:80'],       # (the edit-distance branch needs the absent Levenshtein)
        'match_option': [utils.match_option(v, opts) for v in ['sharp', 'gaus', 'resamp', 'jpeg:80']],
        'match_option_regexp': utils.match_option('^ga.s', opts, regexp=True),
    }))

    # 7. helpers/paramspec.py: a scripted session (values, casts, every validation error with its message)
    from helpers import paramspec
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import paramspec_script
    g['paramspec_transcript'] = np.array(json.dumps(paramspec_script.session(paramspec)))

    np.savez_compressed(OUT, **g)
    print('wrote', OUT, {k: v.shape for k, v in g.items()})


if __name__ == '__main__':
    main()
