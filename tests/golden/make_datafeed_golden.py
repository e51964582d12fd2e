"""
Writes tests/golden/datafeed_sample_patch.npz: outputs of the reference's own `sample_patch` (helpers/loading.py:132-211)
on small synthetic images, for every discard mode, together with a probe of the numpy RNG stream after the calls (so a
restatement must also consume np.random exactly like the reference does).

The function is lifted out of /root/reference/helpers/loading.py at generation time with `ast` and executed here (the
module itself imports imageio / loguru, which this image lacks); nothing of its text is stored.  `np.float`, which the
reference still uses, is aliased to `float` for the call (it was removed from numpy 1.24).

Run in the build container (needs /root/reference):   python tests/golden/make_datafeed_golden.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = '/root/reference/helpers/loading.py'


def reference_sample_patch():
    tree = ast.parse(open(SRC).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'sample_patch'][0]
    mod = ast.Module(body=[fn], type_ignores=[])

    class _NP(object):                                    # numpy with the removed alias restored
        def __getattr__(self, k):
            return float if k == 'float' else getattr(np, k)
    env = {'np': _NP()}
    exec(compile(mod, SRC, 'exec'), env)
    return env['sample_patch']


def test_images():
    """Three 96 x 128 images: mostly flat with textured islands; dark and smooth; busy everywhere."""
    rng = np.random.RandomState(77)
    yy, xx = np.mgrid[0:96, 0:128]
    a = np.full((96, 128, 3), 120.0)
    a[20:60, 70:120] += 90 * rng.uniform(-1, 1, (40, 50, 3))
    a[:, :40] += 18 * np.sin(xx[:, :40, None] / 3.0)
    b = 60 + 30 * np.sin(xx / 40.0)[..., None] + 25 * np.cos(yy / 25.0)[..., None] + rng.normal(0, 4, (96, 128, 3))
    b[50:, 60:] += 110
    c = 128 + 100 * rng.uniform(-1, 1, (96, 128, 3))
    return [np.clip(v, 0, 255).astype(np.uint8) for v in (a, b, c)]


if __name__ == '__main__':
    fn = reference_sample_patch()
    images = test_images()
    out = {'images': np.stack(images)}
    for mode in (None, 'flat', 'flat-aggressive', 'dark-n-textured'):
        for patch, attempts in ((32, 25), (48, 6), (96, 5), (128, 3)):
            np.random.seed(1000 + patch)
            coords = []
            for rep in range(12):
                for img in images:
                    coords.append(fn(img, patch, mode, attempts))
            key = '{}_{}_{}'.format(mode, patch, attempts)
            out['xy_' + key] = np.asarray(coords, np.int64)
            out['probe_' + key] = np.asarray([np.random.randint(0, 1 << 30)], np.int64)
    np.savez_compressed(os.path.join(HERE, 'datafeed_sample_patch.npz'), **out)
    print({k: v.shape for k, v in out.items() if k.startswith('xy')})
