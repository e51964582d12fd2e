"""
TEST INFRASTRUCTURE - CPU restatement of the reference's patch sampling (helpers/loading.py:132-211 `sample_patch`) and batch
cutting (helpers/dataset.py:89-131 `Dataset.next_training_batch`).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product never does.

Pinned: tests/golden/datafeed_sample_patch.npz holds the coordinates the reference's own function returned (and the state of
the numpy RNG stream afterwards) for every discard mode - tests/test_golden.py::test_sample_patch_* replays them through
`sample_patch` below, which is built on the same `Policy` that checks the device kernels.
"""
import numpy as np

MODES = {None: 0, 'flat': 1, 'flat-aggressive': 2, 'dark-n-textured': 3}


def patch_stats(rgb_image, xx, yy, p):
    """loading.py:166-168: variance and mean of the float64 patch / 255 over all three channels."""
    patch = rgb_image[yy:yy + p, xx:xx + p].astype(np.float64) / 255
    return float(np.var(patch)), float(np.mean(patch))


class Policy(object):
    """The accept / retry state machine of loading.py:171-206 for one image.  step() is fed one candidate at a time and
    answers (found, index of the candidate to take)."""

    def __init__(self, discard, max_attempts):
        if discard not in MODES:
            raise ValueError('Unrecognized discard mode: {}'.format(discard))
        self.discard, self.max_attempts, self.panic = discard, max_attempts, max_attempts
        self.best = None                                   # (index, intensity, variance)

    def step(self, k, var, mean, uniform):
        d = self.discard
        if not d:
            return True, k
        if d == 'flat':
            if var < 0.005:
                self.panic -= 1
                return (not self.panic > 0), k
            if var < 0.01:
                return uniform() > 0.5, k                  # the only place a uniform is consumed (:178)
            return True, k
        if d == 'flat-aggressive':
            if var < 0.02:
                if self.panic == self.max_attempts or var > self.best[2]:
                    self.best = (k, mean, var)
                self.panic -= 1
                found = not self.panic > 0
                return found, (self.best[0] if found else k)
            return True, k
        if 0 < var < 0.005 and 0.35 < mean < 0.99:         # dark-n-textured (:193-203)
            return True, k
        if self.panic == self.max_attempts or (var < 2 * self.best[2] and mean > 1.1 * self.best[1]):
            self.best = (k, mean, var)
        self.panic -= 1
        found = not self.panic > 0
        return found, (self.best[0] if found else k)


def sample_patch(rgb_image, rgb_patch_size=128, discard=None, max_attempts=25, rng=np.random):
    """loading.py:132-211 with the numpy global RNG consumed in the same order: randint for x (if the image is wider than
    the patch), randint for y (if taller), then the 'flat' coin flip when it is needed."""
    max_x = rgb_image.shape[1] - rgb_patch_size
    max_y = rgb_image.shape[0] - rgb_patch_size
    if not (max_x > 0 or max_y > 0):
        return 0, 0
    pol = Policy(discard, max_attempts)
    cands = []
    while True:
        xx = 2 * (rng.randint(0, max_x) // 2) if max_x > 0 else 0
        yy = 2 * (rng.randint(0, max_y) // 2) if max_y > 0 else 0
        cands.append((xx, yy))
        var, mean = patch_stats(rgb_image, xx, yy, rgb_patch_size) if discard else (0.0, 0.0)
        found, at = pol.step(len(cands) - 1, var, mean, rng.uniform)
        if found:
            return cands[at]


def select(rgb_image, cands, uniforms, rgb_patch_size, discard, max_attempts):
    """The same policy over a GIVEN candidate list (what the device kernels get): -> ((xx, yy), candidates consumed).  If the
    list runs out the last candidate is taken (the documented deviation of nimg_patch_select)."""
    pol = Policy(discard, max_attempts)
    for k, (xx, yy) in enumerate(cands):
        var, mean = patch_stats(rgb_image, xx, yy, rgb_patch_size) if discard else (0.0, 0.0)
        found, at = pol.step(k, var, mean, lambda: uniforms[k])
        if found:
            return tuple(cands[at]), k + 1
    return tuple(cands[-1]), len(cands)


def cut_batch(raw, rgb, image_idx, xy, rgb_patch_size):
    """dataset.py:119-126: x = raw crop / 65535, y = rgb crop / 255 (float64 quotient stored as float32)."""
    p, ps = rgb_patch_size, rgb_patch_size // 2
    x = None if raw is None else np.zeros((len(image_idx), ps, ps, 4), np.float32)
    y = None if rgb is None else np.zeros((len(image_idx), p, p, 3), np.float32)
    for b, (i, (xx, yy)) in enumerate(zip(image_idx, xy)):
        rx, ry = xx // 2, yy // 2
        if x is not None:
            x[b] = raw[i][ry:ry + ps, rx:rx + ps].astype(np.float64) / (2 ** 16 - 1)
        if y is not None:
            y[b] = rgb[i][yy:yy + p, xx:xx + p].astype(np.float64) / (2 ** 8 - 1)
    return x, y
