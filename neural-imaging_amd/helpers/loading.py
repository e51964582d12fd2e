"""
Finding and loading images, patch sampling - the host half of the data feed (reference helpers/loading.py).  RAW inputs are
*.npy RGGB stacks (H/2, W/2, 4) uint16 next to their rendered *.png (train_prepare_training_set.py writes them that way);
PNGs are decoded with Pillow (the reference uses imageio, absent here).  The device half - cutting batches out of
HBM-resident images - lives in helpers/dataset.DeviceDataset.
"""
import os
import re

import numpy as np

DISCARD_MODES = (None, 'flat', 'flat-aggressive', 'dark-n-textured')


def listdir(path, regex='.*\\..*'):
    path = os.path.expanduser(path)
    return sorted(f for f in os.listdir(path) if re.match(regex, f, re.IGNORECASE))


def discover_images(data_directory, n_images=120, v_images=30, extension='png', randomize=0):
    """-> (training names, validation names): the first n_images and the next v_images of the directory's *.<extension>
    listing (loading.py:14-44).  The listing is sorted; a non-zero `randomize` seeds numpy's GLOBAL generator and shuffles it
    (the split is reproducible, and the patch sampler continues on that stream).  -1 for one side with 0 for the other means
    "every file"."""
    names = listdir(data_directory, '.*\\.{}$'.format(extension))
    if randomize:
        np.random.seed(randomize)
        np.random.shuffle(names)
    if (n_images, v_images) == (0, -1):
        v_images = len(names)
    elif (n_images, v_images) == (-1, 0):
        n_images = len(names)
    if n_images + v_images > len(names):
        raise ValueError('Not enough images!')
    return names[:n_images], names[n_images:n_images + v_images]


def _read_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'), dtype=np.uint8)


def _read_pair(data_directory, name, extension, load):
    """(RAW stack or None, RGB image or None) of one file name."""
    stem = name[:-len(extension) - 1]
    raw = np.load(os.path.join(data_directory, stem + '.npy')) if 'x' in load else None
    rgb = _read_rgb(os.path.join(data_directory, name)) if 'y' in load else None
    return raw, rgb


def load_images(files, data_directory, extension='png', load='xy'):
    """Full-resolution (raw, rgb) pairs -> {'x': (n, H/2, W/2, 4) uint16, 'y': (n, H, W, 3) uint8} (loading.py:47-88); an
    empty file list answers 1x1x1x1 zeros per kind, like the reference."""
    if not files:
        return {k: np.zeros((1, 1, 1, 1)) for k in load}
    pairs = [_read_pair(data_directory, f, extension, load) for f in files]
    out = {}
    if 'x' in load:
        out['x'] = np.stack([p[0] for p in pairs]).astype(np.uint16, copy=False)
    if 'y' in load:
        out['y'] = np.stack([p[1] for p in pairs])
    return out


def load_patches(files, data_directory, patch_size=128, n_patches=100, discard='flat-aggressive', extension='png',
                 load='xy'):
    """n_patches random (raw, rgb) patches per image; patch_size counts RAW pixels (loading.py:91-129).  The corners are
    drawn on the RGB image (100 attempts per patch), which is therefore always read."""
    rgb_side = 2 * patch_size
    cut = {k: [] for k in load}
    for name in files:
        raw, _ = _read_pair(data_directory, name, extension, 'x' if 'x' in load else '')
        rgb = _read_rgb(os.path.join(data_directory, name))
        for _ in range(n_patches):
            xx, yy = sample_patch(rgb, rgb_side, discard, 100)
            if 'x' in cut:
                cut['x'].append(raw[yy // 2:yy // 2 + patch_size, xx // 2:xx // 2 + patch_size])
            if 'y' in cut:
                cut['y'].append(rgb[yy:yy + rgb_side, xx:xx + rgb_side])
    shapes = {'x': (0, patch_size, patch_size, 4), 'y': (0, rgb_side, rgb_side, 3)}
    types = {'x': np.uint16, 'y': np.uint8}
    return {k: (np.stack(v).astype(types[k], copy=False) if v else np.zeros(shapes[k], types[k])) for k, v in cut.items()}


def sample_patch(rgb_image, rgb_patch_size=128, discard=None, max_attempts=25):
    """(x, y) of a patch, even numbers for the Bayer alignment, drawn from the numpy global RNG under one of the discard
    policies (loading.py:132-211):
      flat             retry while the patch variance is < 0.005 (panic after max_attempts), accept 0.005..0.01 on a coin flip
      flat-aggressive  retry while variance < 0.02, falling back to the most textured candidate seen
      dark-n-textured  want 0 < variance < 0.005 and 0.35 < mean < 0.99, falling back to the best smooth-and-bright one
    """
    if discard not in DISCARD_MODES:
        raise ValueError('Unrecognized discard mode: {}'.format(discard))
    max_x = rgb_image.shape[1] - rgb_patch_size
    max_y = rgb_image.shape[0] - rgb_patch_size
    xx, yy = 0, 0
    if max_x <= 0 and max_y <= 0:
        return xx, yy
    remaining = max_attempts
    fallback = None                         # (xx, yy, mean, variance) of the best rejected candidate
    while True:
        xx = 2 * (np.random.randint(0, max_x) // 2) if max_x > 0 else 0
        yy = 2 * (np.random.randint(0, max_y) // 2) if max_y > 0 else 0
        if not discard:
            return xx, yy
        patch = rgb_image[yy:yy + rgb_patch_size, xx:xx + rgb_patch_size].astype(np.float64) / 255
        variance, intensity = np.var(patch), np.mean(patch)
        if discard == 'flat':
            if variance >= 0.01 or (variance >= 0.005 and np.random.uniform() > 0.5):
                return xx, yy
            if variance >= 0.005:
                continue                     # lost the coin flip: a retry that does not count as an attempt
            remaining -= 1
            if remaining <= 0:
                return xx, yy
            continue
        if discard == 'flat-aggressive':
            if variance >= 0.02:
                return xx, yy
            if remaining == max_attempts or variance > fallback[3]:
                fallback = (xx, yy, intensity, variance)
        else:
            if 0 < variance < 0.005 and 0.35 < intensity < 0.99:
                return xx, yy
            if remaining == max_attempts or (variance < 2 * fallback[3] and intensity > 1.1 * fallback[2]):
                fallback = (xx, yy, intensity, variance)
        remaining -= 1
        if remaining <= 0:
            return fallback[0], fallback[1]
