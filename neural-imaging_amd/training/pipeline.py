"""
NIP pre-training loop - counterpart of the reference's training/pipeline.py:105-256: lr schedule
dict {epoch: lr} (default {0: 1e-4}), per-batch model.training_step(bx, by, lr), validation every
`validation_schedule` epochs (PSNR / loss), lr x0.95 when the validation loss exceeds 1.2 x its best (:224-227), early
stop on a converged validation loss (:230-238), progress.json + checkpoint, `resume` reloads weights + progress (and continues
with the epoch AFTER the recorded one - the reference restarts at the recorded epoch itself and logs it twice, :155).
train_nip_bare (:259-302): the same loop without validation, logging or files.
"""
import json
import os
from collections import OrderedDict

import numpy as np

from .. import parallel
from . import validation


def save_progress(model, training_summary, out_directory):
    os.makedirs(out_directory, exist_ok=True)
    with open(os.path.join(out_directory, 'progress.json'), 'w') as f:
        json.dump({'performance': model.performance, 'summary': training_summary, 'args': model.get_hyperparameters()},
                  f, indent=4, default=lambda o: float(o))


def train_nip_model(model, camera_name, n_epochs=10000, lr_schedule=None, validation_loss_threshold=1e-3,
                    validation_schedule=100, resume=False, patch_size=64, batch_size=20, data=None,
                    out_directory_root='./data/models/nip', save_best=False, discard='flat'):
    if data is None:
        raise ValueError('Training data seems not to be loaded!')
    try:
        bx, by = data.next_training_batch(0, 5, patch_size * 2)
        if bx.shape != (5, patch_size, patch_size, 4) or by.shape != (5, 2 * patch_size, 2 * patch_size, 3):
            raise ValueError('The training batch returned by the dataset instance is of invalid size!')
    except Exception as e:
        raise ValueError('Data set error: {}'.format(e))
    if batch_size > data.count_training or batch_size > data.count_validation:
        raise ValueError('Batch size ({}) exceeds dataset size ({}/{})!'.format(batch_size, data.count_training,
                                                                                data.count_validation))
    out_directory = os.path.join(out_directory_root, camera_name, model.model_code, model.scoped_name)
    if os.path.exists(out_directory) and not resume:
        return out_directory
    start_epoch = 0
    if resume:
        if not os.path.isfile(os.path.join(out_directory, 'progress.json')):
            raise FileNotFoundError('Could not open file {}'.format(os.path.join(out_directory, 'progress.json')))
        model.load_model(out_directory)
        with open(os.path.join(out_directory, 'progress.json')) as f:
            prog = json.load(f)
        model.performance = prog['performance']
        start_epoch = prog['summary'].get('Epoch', 0) + 1
    if lr_schedule is None:
        lr_schedule = {0: 1e-4}
    elif isinstance(lr_schedule, float):
        lr_schedule = {0: lr_schedule}
    n_batches = data.count_training // batch_size
    n_tail = 5
    summary = OrderedDict([('Camera', camera_name), ('Architecture', model.summary()), ('Max epochs', n_epochs),
                           ('Learning rate', {str(k): v for k, v in lr_schedule.items()}), ('# batches', n_batches),
                           ('Patch size', patch_size), ('Batch size', batch_size),
                           ('Validation schedule', validation_schedule), ('Start epoch', start_epoch),
                           ('Saved checkpoint', None), ('Output directory', out_directory)])
    # the rate in force at the first epoch trained: the entry of the largest schedule key <= start_epoch (a resumed run never meets
    # key 0 inside the loop; the reference then falls back to 1e-4 whatever the schedule says - ADVICE r05)
    learning_rate = 1e-4
    for k in sorted(lr_schedule):
        if k <= start_epoch:
            learning_rate = lr_schedule[k]
    epoch = start_epoch
    world, rank = parallel.world_size(), parallel.rank()
    if batch_size % world:
        raise ValueError('batch_size {} does not split over {} ranks'.format(batch_size, world))
    if start_epoch >= n_epochs:
        # resuming a finished run trains nothing: leave its progress file and checkpoint as they are (no checkpoint under an epoch
        # that was never trained, no 'Epoch' that grows with every call)
        return out_directory
    for epoch in range(start_epoch, n_epochs):
        if epoch in lr_schedule:
            learning_rate = lr_schedule[epoch]
        losses = []
        for batch_id in range(n_batches):
            bx, by = data.next_training_batch(batch_id, batch_size, patch_size, discard=discard)
            if world > 1:          # batch_size is the global batch; every rank trains on its contiguous shard
                bx, by = parallel.shard_batch(bx, rank, world), parallel.shard_batch(by, rank, world)
            losses.append(model.training_step(bx, by, learning_rate))          # device scalars, read once per epoch
        model.log_metric('loss', 'training', [float(v) for v in losses])
        if epoch % validation_schedule == 0:
            ssims, psnrs, v_losses = validation.validate_nip(model, data, out_directory, epoch=epoch,
                                                             loss_type=model.loss_metric)
            model.log_metric('ssim', 'validation', ssims)
            model.log_metric('psnr', 'validation', psnrs)
            model.log_metric('loss', 'validation', v_losses)
            summary['Epoch'] = epoch
            vl = model.performance['loss']['validation']
            keep = not save_best or (len(vl) > 2 and vl[-1] <= min(vl))
            if keep:
                summary['Saved checkpoint'] = epoch
            if rank == 0:
                save_progress(model, summary, out_directory)
                if keep:
                    model.save_model(out_directory, epoch, quiet=True)
            if len(vl) > 5 and vl[-1] > 1.2 * min(vl):
                learning_rate = max(learning_rate * 0.95, 1e-7)
            if validation_loss_threshold is not None and len(vl) > 10:
                current, previous = np.mean(vl[-n_tail:-1]), np.mean(vl[-(n_tail + 1):-2])
                if abs((current - previous) / previous) < validation_loss_threshold:
                    break
    summary['Epoch'] = epoch
    vl = model.performance['loss']['validation']
    keep = not save_best or (len(vl) > 0 and vl[-1] <= min(vl))
    if keep:
        summary['Saved checkpoint'] = epoch
    if rank == 0:
        if keep:
            model.save_model(out_directory, epoch, quiet=True)
        save_progress(model, summary, out_directory)
    return out_directory


def train_nip_bare(model, camera_name, n_epochs=10000, lr_schedule=None, validation_loss_threshold=1e-3,
                   validation_schedule=100, resume=False, patch_size=64, batch_size=20, data=None,
                   out_directory_root='./data/models/nip', save_best=False, discard='flat'):
    """The bare loop of the reference (training/pipeline.py:259-302): training steps only, at a learning rate of 1e-3 (the
    schedule argument is accepted and, as there, never consulted), from a Dataset or from any iterable of (x, y) batches."""
    out_directory = os.path.join(out_directory_root, camera_name, model.model_code, model.scoped_name)
    learning_rate = 1e-3
    world, rank = parallel.world_size(), parallel.rank()
    for _ in range(n_epochs):
        if hasattr(data, 'next_training_batch'):
            batches = (data.next_training_batch(b, batch_size, patch_size, discard=discard)
                       for b in range(data.count_training // batch_size))
        else:
            batches = iter(data)
        for bx, by in batches:
            if world > 1:
                bx, by = parallel.shard_batch(bx, rank, world), parallel.shard_batch(by, rank, world)
            model.training_step(bx, by, learning_rate)
    return out_directory
