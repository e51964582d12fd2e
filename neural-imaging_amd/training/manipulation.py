"""
Epoch loop of the manipulation-classification workflow - the build's counterpart of the reference's
training/manipulation.py:36-335: same training-spec keys, same step call
`flow.training_step(batch_x, batch_y, lambda_nip, lambda_dcn, augment, learning_rate)`, learning rate x0.9 every 100
epochs starting at epoch 0 (:268-269), validation every `validation_schedule` epochs (FAN accuracy / confusion, NIP
PSNR), `training.json` with the reference's keys, checkpoints per validation, and the same output directory naming
root/camera/NIP/{ln-x|fixed-nip}/{lc-x|fixed-codec}/run (:111-123) with "directory exists => skip" idempotence.
`data` is any object with the reference Dataset's duck type (helpers/dataset.py) - see SyntheticDataset below.
"""
import os
import shutil
from collections import OrderedDict, deque

import numpy as np

from .. import parallel
from ..helpers import utils
from ..models import compression, jpeg
from . import validation


def default_training_specs():
    return {'use_pretrained_nip': True, 'patch_size': 64, 'batch_size': 10, 'validation_schedule': 50,
            'n_epochs': 1001, 'learning_rate': 1e-4, 'run_number': 0, 'lambda_nip': 0.1, 'lambda_dcn': 0,
            'augment': False}


class SyntheticDataset(object):
    """Stand-in with the interface of helpers/dataset.Dataset (next_training_batch / next_validation_batch /
    count_* / is_raw_and_rgb / summary): natural-image-like RGB patches and their RGGB-ordered Bayer stacks of a GBRG mosaic."""

    def __init__(self, n_training=40, n_validation=20, patch_size=64, seed=1234, raw=True):
        rng = np.random.default_rng(seed)
        self._loaded_data = 'xy' if raw else 'y'
        self.count_training, self.count_validation = n_training, n_validation
        self._rgb_patch = 2 * patch_size if raw else patch_size

        def make(n):
            h = self._rgb_patch
            base = rng.random((n, h // 8 + 1, h // 8 + 1, 3))
            img = np.kron(base, np.ones((1, 8, 8, 1)))[:, :h, :h, :]
            img = 0.7 * img + 0.3 * np.linspace(0, 1, h)[None, None, :, None] + rng.normal(0, 0.03, size=(n, h, h, 3))
            return (np.round(np.clip(img, 0, 1) * 255) / 255).astype(np.float32)
        self._y = {'training': make(n_training), 'validation': make(n_validation)}

    @staticmethod
    def _bayer(rgb):
        # RGGB-ordered stack of a GBRG mosaic (helpers/raw.py:204-225 stack_bayer): R (1,0), G1 (0,0), G2 (1,1), B (0,1)
        return np.stack([rgb[:, 1::2, 0::2, 0], rgb[:, 0::2, 0::2, 1], rgb[:, 1::2, 1::2, 1], rgb[:, 0::2, 1::2, 2]],
                        axis=-1).astype(np.float32)

    def is_raw_and_rgb(self):
        return self._loaded_data == 'xy'

    def _batch(self, split, batch_id, batch_size):
        y = self._y[split][batch_id * batch_size:(batch_id + 1) * batch_size]
        return (self._bayer(y), y) if self._loaded_data == 'xy' else y

    def next_training_batch(self, batch_id, batch_size, rgb_patch_size=None, discard=None):
        return self._batch('training', batch_id, batch_size)

    def next_validation_batch(self, batch_id, batch_size):
        return self._batch('validation', batch_id, batch_size)

    def summary(self):
        return 'synthetic {} : {} training + {} validation patches'.format(self._loaded_data, self.count_training,
                                                                         self.count_validation)


def train_manipulation_nip(flow, training, data, directories=None, overwrite=False):
    directories_def = {'root': './data/m/', 'nip_snapshots': './data/models/nip/'}
    directories_def.update(directories or {})
    directories = directories_def
    spec = default_training_specs()
    spec.update(training or {})
    training = spec
    required = {'camera_name', 'use_pretrained_nip', 'lambda_nip', 'lambda_dcn', 'run_number', 'n_epochs',
                'learning_rate', 'augment'}
    if any(k not in training for k in required):
        raise RuntimeError('Missing keys in the training dictionary! {}'.format(required.difference(training.keys())))
    if data is None:
        raise ValueError('Training data seems not to be loaded!')
    ps = training['patch_size']
    try:
        if data.is_raw_and_rgb():
            bx, by = data.next_training_batch(0, 1, ps * 2)
            if bx.shape != (1, ps, ps, 4) or by.shape != (1, 2 * ps, 2 * ps, 3):
                raise ValueError('The RAW+RGB training batch is of invalid size! {}'.format(bx.shape))
        else:
            bx = data.next_training_batch(0, 1, ps * 2)
            if bx.shape != (1, 2 * ps, 2 * ps, 3):
                raise ValueError('The RGB training batch is of invalid size! {}'.format(bx.shape))
    except Exception as e:
        raise ValueError('Data set error: {}'.format(e))

    save_dir = [directories['root'], training['camera_name'], flow.nip.class_name,
                'ln-{:0.4f}'.format(training['lambda_nip']) if flow.is_trainable('nip') else 'fixed-nip',
                'lc-{:0.4f}'.format(training['lambda_dcn']) if flow.is_trainable('dcn') else 'fixed-codec',
                '{:03d}'.format(training['run_number'])]
    save_dir = os.path.join(*save_dir)
    model_directory = os.path.join(save_dir, 'models')
    if os.path.exists(save_dir) and not overwrite:
        return model_directory
    if flow.is_trainable('nip') and flow.nip.count_parameters() == 0:
        raise ValueError('It looks like you`re trying to optimize a NIP with no trainable parameters!')

    decay_schedule, decay_rate = 100, 0.90
    learning_rate = training['learning_rate']
    n_batches = data.count_training // training['batch_size']
    if training['use_pretrained_nip'] and flow.nip.count_parameters() > 0:
        nip_dirname = os.path.join(directories['nip_snapshots'], training['camera_name'], flow.nip.model_code)
        if os.path.isdir(nip_dirname):
            flow.nip.load_model(nip_dirname)

    loss_epoch = {k: deque(maxlen=n_batches) for k in ('nip', 'fan')}
    summary = OrderedDict()
    summary['Problem'] = flow.summary()
    summary['Dataset'] = data.summary()
    summary['Camera name'] = training['camera_name']
    summary['Classes'] = '{}'.format(flow._forensics_classes)
    summary['FAN model'] = flow.fan.summary()
    summary['NIP model'] = flow.nip.summary()
    summary['Channel Downsampling'] = flow._distribution['downsampling']
    summary['Channel Compression'] = flow.codec.summary() if flow.codec is not None else 'n/a'
    summary['Channel Compression Parameters'] = str(flow._distribution['compression_params'])
    summary['Joint optimization'] = '{}'.format(flow.trainable_models)
    summary['NIP Regularization'] = utils.format_number(training['lambda_nip'])                  # (:171-186: the reference's formats)
    summary['DCN Regularization'] = utils.format_number(training['lambda_dcn'])
    summary['NIP loss'] = '{}'.format(flow.nip.loss_metric)
    summary['Use pre-trained NIP'] = str(training['use_pretrained_nip'])
    summary['# Epochs'] = utils.format_number(training['n_epochs'])
    summary['Patch size'] = utils.format_number(ps)
    summary['Batch size'] = utils.format_number(training['batch_size'])
    summary['Learning rate'] = utils.format_number(training['learning_rate'])
    summary['Learning rate decay schedule'] = utils.format_number(decay_schedule)
    summary['Learning rate decay rate'] = utils.format_number(decay_rate)
    summary['Validation schedule'] = training['validation_schedule']
    summary['Augmentation'] = str(training['augment'])
    summary['# train. images'] = utils.format_number(data.count_training)
    summary['# valid. images'] = utils.format_number(data.count_validation)
    summary['Batch shape'] = '{}'.format(tuple(bx.shape))
    summary['NIP input patch'] = '{}'.format(flow.nip.x.shape)
    summary['NIP output patch'] = '{}'.format(flow.nip.y.shape)
    summary['FAN input patch'] = '{}'.format(flow.fan.x.shape)

    world, rank = parallel.world_size(), parallel.rank()
    if training['batch_size'] % world:
        raise ValueError('batch_size {} does not split over {} ranks'.format(training['batch_size'], world))
    codec_is_dcn = flow.is_trainable('dcn') and isinstance(flow.codec, compression.DCN)

    def validate_and_save(epoch, final=False):
        """One validation pass over every trained model, training.json and the checkpoints (training/manipulation.py:224-259
        inside the loop, :300-334 after it).  Every rank validates (the learned codec's entropy is a collective under data
        parallelism); rank 0 writes."""
        accuracy, conf = validation.validate_fan(flow, data)
        flow.fan.log_metric('accuracy', 'validation', accuracy)
        flow.fan.performance['confusion'] = conf.tolist()
        if flow.is_trainable('nip') and data.is_raw_and_rgb():
            values = validation.validate_nip(flow.nip, data, save_dir, epoch=epoch,
                                             loss_type='L2' if final else flow.nip.loss_metric)
            for metric, arr in zip(['ssim', 'psnr', 'loss'], values):
                flow.nip.log_metric(metric, 'validation', arr)
        if codec_is_dcn:
            for metric, value in validation.validate_dcn(flow.codec, data, save_dir, epoch=epoch).items():
                flow.codec.log_metric(metric, 'validation', value)
        elif flow.is_trainable('dcn') and isinstance(flow.codec, jpeg.JPEG) and not final:      # trainable tables (:241-242)
            for metric, value in validation.validate_jpeg(flow.codec, data).items():
                flow.codec.log_metric(metric, 'validation', value)
        elif flow.is_trainable('dcn') and not final:
            raise NotImplementedError('Validation for {} codec doesn\'t seem to be implemented'.format(flow.codec))
        if rank != 0:
            return
        validation.save_training_progress(summary, flow, save_dir, quiet=True)
        flow.fan.save_model(os.path.join(model_directory, flow.fan.scoped_name), epoch, quiet=True)
        if flow.is_trainable('nip'):
            flow.nip.save_model(os.path.join(model_directory, flow.nip.scoped_name), epoch, quiet=True)
        if codec_is_dcn:
            codec_dir = os.path.join(model_directory, flow.codec.scoped_name)
            flow.codec.save_model(codec_dir, epoch, quiet=True)
            src = flow._distribution.get('compression_params', {}).get('dirname')
            src = None if src is None else os.path.join(src, flow.codec.scoped_name, 'progress.json')
            if final and src is not None and os.path.isfile(src):          # the pre-training record travels with the model
                shutil.copyfile(src, os.path.join(codec_dir, 'progress.json'))

    epoch = 0
    for epoch in range(0, training['n_epochs']):
        for batch_id in range(n_batches):
            if data._loaded_data == 'xy':
                batch_x, batch_y = data.next_training_batch(batch_id, training['batch_size'], 2 * ps)
            else:
                batch_x = data.next_training_batch(batch_id, training['batch_size'], 2 * ps)
                batch_y = batch_x
            if world > 1:          # the global batch is the reference's batch; every rank trains on its contiguous shard
                batch_x, batch_y = parallel.shard_batch(batch_x, rank, world), parallel.shard_batch(batch_y, rank, world)
            comb_loss, comp_loss = flow.training_step(batch_x, batch_y, training['lambda_nip'], training['lambda_dcn'],
                                                      training['augment'], learning_rate)
            loss_epoch['fan'].append(comb_loss)            # lazy device values: read once per epoch, not once per step
            loss_epoch['nip'].append(comp_loss['nip'])
        if getattr(flow, '_nan_check', 'eager') == 'deferred':
            flow.check_nan()                               # NaN steps of this epoch (their updates were skipped on the device)
        for name, model in (('nip', flow.nip), ('fan', flow.fan)):
            model.log_metric('loss', 'training', [float(v) for v in loss_epoch[name]])
        if epoch % training['validation_schedule'] == 0:
            validate_and_save(epoch)
        if epoch % decay_schedule == 0:
            learning_rate *= decay_rate
    validate_and_save(epoch, final=True)
    return model_directory
