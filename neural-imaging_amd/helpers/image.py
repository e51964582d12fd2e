"""
Batch-level image helpers of the reference's helpers/image.py that the training harnesses call.

batch_gamma (helpers/image.py:22-28): x ** (1 / gamma) clipped to [0, 1], gamma a float for the whole batch or one value per image
drawn uniformly from [0.25, 3) when not given - the 'gamma' augmentation of the codec's pre-training loop
(training/compression.py:197).  Host arrays stay numpy; a device batch (torch tensor) is transformed where it lives.
"""
import numpy as np


def batch_gamma(batch_p, gamma=None, rng=None):
    n = len(batch_p)
    if gamma is None:
        draw = (rng or np.random).uniform(low=0.25, high=3, size=(n, 1, 1, 1))
        gamma = np.array(draw, dtype=np.float32)
    elif type(gamma) is float:
        gamma = gamma * np.ones((n, 1, 1, 1))
    if isinstance(batch_p, np.ndarray):
        return np.power(batch_p, 1 / gamma).clip(0, 1)
    import torch
    g = torch.as_tensor(np.asarray(1 / gamma, np.float32), device=batch_p.device)
    return torch.pow(batch_p, g).clamp_(0, 1)
