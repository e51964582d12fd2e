"""
DCN pre-training loop - counterpart of the reference's training/compression.py:123-309: per batch
flips and the per-image gamma augmentation (:195-197), dcn.training_step(batch, lr), lr x0.5 every 1000 epochs by default (train_dcn.py:105-107),
validation every `validation_schedule` epochs with progress.json + checkpoint (only then, like the reference), early stop
on a converged or deteriorating validation SSIM (:282-295); a non-finite loss aborts.
"""
import json
import os
from collections import OrderedDict

import numpy as np

from .. import parallel
from ..helpers.image import batch_gamma
from . import validation


def train_dcn(dcn, training, data, directory='./data/models/dcn/playground/', overwrite=False):
    spec = {'n_epochs': 1500, 'batch_size': 50, 'patch_size': dcn.patch_size, 'learning_rate': 1e-4,
            'learning_rate_reduction_schedule': 1000, 'learning_rate_reduction_factor': 0.5,
            'validation_schedule': 100, 'augmentation_probs': {'resize': 0.0, 'flip_h': 0.5, 'flip_v': 0.5, 'gamma': 0.5},   # train_dcn.py:112-117
            'seed': 1234,
            'convergence_threshold': 1e-5}
    spec.update(training or {})
    out = os.path.join(directory, dcn.model_code.split('/')[0], dcn.scoped_name)
    if os.path.exists(out) and not overwrite:
        return out
    rng = np.random.RandomState(spec['seed'])
    if spec['augmentation_probs'].get('resize', 0.0) > 0:
        # training/compression.py:176-192 samples a larger patch and shrinks it with skimage's anti-aliased resize (absent here);
        # train_dcn.py fixes the probability at 0
        raise NotImplementedError('the resize augmentation (skimage.transform.resize) is not built; train_dcn.py runs with 0.0')
    world, rank = parallel.world_size(), parallel.rank()
    if spec['batch_size'] % world:
        raise ValueError('batch_size {} does not split over {} ranks'.format(spec['batch_size'], world))
    n_batches = data.count_training // spec['batch_size']
    lr = spec['learning_rate']
    summary = OrderedDict([('model', dcn.summary()), ('epochs', spec['n_epochs']), ('batch', spec['batch_size']),
                           ('lr', spec['learning_rate'])])
    for epoch in range(spec['n_epochs']):
        stats = {'loss': [], 'entropy': []}
        for batch_id in range(n_batches):
            bx = data.next_training_batch(batch_id, spec['batch_size'], spec['patch_size'])
            bx = bx[1] if isinstance(bx, tuple) else bx
            flips = [ax for ax, key in ((2, 'flip_h'), (1, 'flip_v')) if rng.uniform() < spec['augmentation_probs'][key]]
            gamma = rng.uniform() < spec['augmentation_probs'].get('gamma', 0.0)         # (drawn in the reference's order)
            if hasattr(bx, 't'):                  # DeviceArray (DeviceDataset): flip on the device
                import torch
                bx = torch.flip(bx.t, flips).contiguous() if flips else bx.t
            else:
                for ax in flips:
                    bx = np.flip(bx, ax)
                bx = np.ascontiguousarray(bx)
            if gamma:                             # one exponent per image of the GLOBAL batch: every rank draws the same ones
                bx = batch_gamma(bx, rng=rng)
            if world > 1:          # batch_size is the global batch; every rank trains on its contiguous shard
                bx = parallel.shard_batch(bx, rank, world).contiguous() if hasattr(bx, 'contiguous') else \
                    np.ascontiguousarray(parallel.shard_batch(bx, rank, world))
            res = dcn.training_step(bx, lr, sync=False)         # device scalars, read once per epoch
            stats['loss'].append(res['loss'])
            stats['entropy'].append(res['entropy'])
        stats = {k: [float(v) for v in vs] for k, vs in stats.items()}
        if not np.isfinite(stats['loss']).all():
            raise RuntimeError('DCN training diverged (non-finite loss)')
        dcn.log_metric('loss', 'training', stats['loss'])
        dcn.log_metric('entropy', 'training', stats['entropy'])
        if epoch % spec['validation_schedule'] == 0:
            vals = validation.validate_dcn(dcn, data, out, epoch=epoch)
            for k, v in vals.items():
                dcn.log_metric(k, 'validation', v)
            if rank == 0:
                os.makedirs(out, exist_ok=True)
                with open(os.path.join(out, 'progress.json'), 'w') as f:
                    json.dump({'performance': dcn.performance, 'summary': summary, 'args': dcn.get_hyperparameters()}, f,
                              indent=4, default=lambda o: float(o))
                dcn.save_model(out, epoch, save_args=True, quiet=True)
            # convergence / deterioration of the validation SSIM over the last n_tail samplings (compression.py:282-295)
            vs, n_tail = dcn.performance['ssim']['validation'], 5
            if len(vs) > 5:
                current, previous = np.mean(vs[-n_tail:]), np.mean(vs[-(n_tail + 1):-1])
                if abs((current - previous) / previous) < spec['convergence_threshold'] or current < 0.9 * previous:
                    break
        if epoch > 0 and epoch % spec['learning_rate_reduction_schedule'] == 0:
            lr *= spec['learning_rate_reduction_factor']
    return out
