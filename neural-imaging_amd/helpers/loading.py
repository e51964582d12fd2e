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
    """Training / validation split of the directory listing (loading.py:14-44): sorted names, shuffled by the numpy
    global RNG seeded with `randomize` when it is non-zero; (0, -1) / (-1, 0) take every file for one side."""
    files = listdir(data_directory, '.*\\.{}$'.format(extension))
    if randomize:
        np.random.seed(randomize)
        np.random.shuffle(files)
    if n_images == 0 and v_images == -1:
        v_images = len(files)
    if n_images == -1 and v_images == 0:
        n_images = len(files)
    if len(files) < n_images + v_images:
        raise ValueError('Not enough images!')
    return files[0:n_images], files[n_images:(n_images + v_images)]


def _read_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'), dtype=np.uint8)


def load_images(files, data_directory, extension='png', load='xy'):
    """Full-resolution (raw, rgb) pairs -> {'x': (n, H/2, W/2, 4) uint16, 'y': (n, H, W, 3) uint8} (loading.py:47-88)."""
    n_images = len(files)
    if n_images == 0:
        return {k: np.zeros(shape=(1, 1, 1, 1)) for k in load}
    first = _read_rgb(os.path.join(data_directory, files[0]))
    res = (first.shape[0] >> 1, first.shape[1] >> 1)
    data = {}
    if 'x' in load:
        data['x'] = np.zeros((n_images, res[0], res[1], 4), dtype=np.uint16)
    if 'y' in load:
        data['y'] = np.zeros((n_images, 2 * res[0], 2 * res[1], 3), dtype=np.uint8)
    for i, file in enumerate(files):
        if 'x' in data:
            data['x'][i] = np.load(os.path.join(data_directory, file.replace('.{}'.format(extension), '.npy')))
        if 'y' in data:
            data['y'][i] = _read_rgb(os.path.join(data_directory, file))
    return data


def load_patches(files, data_directory, patch_size=128, n_patches=100, discard='flat-aggressive', extension='png',
                 load='xy'):
    """n_patches random (raw, rgb) patches per image; patch_size counts RAW pixels (loading.py:91-129)."""
    max_attempts = 100
    data = {}
    if 'x' in load:
        data['x'] = np.zeros((len(files) * n_patches, patch_size, patch_size, 4), dtype=np.uint16)
    if 'y' in load:
        data['y'] = np.zeros((len(files) * n_patches, 2 * patch_size, 2 * patch_size, 3), dtype=np.uint8)
    for i, file in enumerate(files):
        image_x = np.load(os.path.join(data_directory, file.replace('.{}'.format(extension), '.npy'))) if 'x' in data \
            else None
        # the reference samples on the RGB image; without it ('x' only) the coordinates fall back to (0, 0) there too
        image_y = _read_rgb(os.path.join(data_directory, file))
        for b in range(n_patches):
            xx, yy = sample_patch(image_y, 2 * patch_size, discard, max_attempts)
            rx, ry = xx // 2, yy // 2
            if 'x' in data:
                data['x'][i * n_patches + b] = image_x[ry:ry + patch_size, rx:rx + patch_size, :]
            if 'y' in data:
                data['y'][i * n_patches + b] = image_y[yy:yy + 2 * patch_size, xx:xx + 2 * patch_size, :]
    return data


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
