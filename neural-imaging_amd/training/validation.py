"""
Validation helpers and the training-progress JSON of the reference's training/validation.py:
  validate_fan (:163-202)           confusion matrix / accuracy through flow.run_workflow_to_decisions
  validate_nip (:96-160, numbers only)  PSNR + loss of the developed validation patches (no matplotlib dashboards)
  save_training_progress (:301-352) training.json with the same keys
SSIM / PSNR come from helpers/metrics.py (device-side SSIM kernel, skimage semantics).
"""
import json
import os
from collections import OrderedDict

import numpy as np

from ..helpers import metrics


def psnr(a, b, max_val=1.0):
    mse = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2, axis=(1, 2, 3))
    return 10 * np.log10(max_val ** 2 / np.maximum(mse, 1e-30))


def validate_fan(flow, data, get_labels=False):
    batch_size = int(np.minimum(10, data.count_validation))
    n_batches = data.count_validation // batch_size
    n_classes = flow.n_classes
    conf = np.zeros((n_classes, n_classes))
    out_labels, accuracies = [], []
    for batch in range(n_batches):
        batch_x = data.next_validation_batch(batch, batch_size)
        if isinstance(batch_x, tuple):
            batch_x = batch_x[0]
        batch_y = flow._batch_labels(len(batch_x))
        predicted_labels = flow.run_workflow_to_decisions(batch_x)
        if get_labels:
            out_labels += [x for x in predicted_labels]
        for c in range(n_classes):
            for c_ in range(n_classes):
                conf[c, c_] += np.sum((batch_y == c) * (predicted_labels == c_))
        accuracies.append(np.mean(predicted_labels == batch_y))
    if out_labels:
        return np.mean(accuracies), conf / (n_batches * batch_size), out_labels
    return np.mean(accuracies), conf / (n_batches * batch_size)


def validate_nip(model, data, out_directory=None, savefig=False, epoch=0, show_ref=False, loss_type='L2'):
    """Returns (ssims, psnrs, losses) over the validation set, one value per image."""
    ssims, psnrs, losses = [], [], []
    batch_size = int(np.minimum(10, data.count_validation))
    for batch in range(data.count_validation // batch_size):
        bx, by = data.next_validation_batch(batch, batch_size)
        by = np.asarray(by)                       # a DeviceDataset answers DeviceArrays
        developed = model.process(bx).numpy().clip(0, 1)
        psnrs.extend(psnr(developed, by).tolist())
        ssims.extend(np.atleast_1d(metrics.ssim(developed, by)).tolist())
        d = 255.0 * (developed - by)
        losses.extend((np.mean(d ** 2, axis=(1, 2, 3)) if loss_type == 'L2' else np.mean(np.abs(d), axis=(1, 2, 3))).tolist())
    return ssims, psnrs, losses


def validate_dcn(dcn, data, out_directory=None, savefig=False, epoch=0, show_ref=False):
    """Returns {'ssim','psnr','entropy','loss'} lists over the validation set (training/validation.py:44-93)."""
    out = {'ssim': [], 'psnr': [], 'entropy': [], 'loss': []}
    batch_size = int(np.minimum(10, data.count_validation))
    for batch in range(data.count_validation // batch_size):
        by = data.next_validation_batch(batch, batch_size)
        by = by[1] if isinstance(by, tuple) else by
        y, ent = dcn.process(by, return_entropy=True)
        by = np.asarray(by)
        out['psnr'].extend(psnr(y.numpy(), by).tolist())
        out['ssim'].extend(np.atleast_1d(metrics.ssim(np.clip(y.numpy(), 0, 1), by)).tolist())
        out['entropy'].append(float(ent))
        out['loss'].append(float(np.sqrt(2 * dcn.loss(by, y, float(ent)))))
    return out


def save_training_progress(training_summary, flow, root_dir, quiet=False):
    training = OrderedDict()
    training['summary'] = training_summary
    training['distribution'] = flow._distribution
    training['manipulations'] = flow._forensics_classes
    training['nip'] = OrderedDict()
    training['nip']['model'] = flow.nip.class_name
    training['nip']['init'] = repr(flow.nip)
    training['nip']['args'] = flow.nip._h.to_json() if hasattr(flow.nip, '_h') else {}
    training['nip']['performance'] = flow.nip.performance
    training['forensics'] = OrderedDict()
    training['forensics']['model'] = flow.fan.class_name
    training['forensics']['init'] = repr(flow.fan)
    training['forensics']['args'] = flow.fan._h.to_json()
    training['forensics']['performance'] = flow.fan.performance
    if flow.codec is not None:
        training['codec'] = OrderedDict()
        training['codec']['model'] = flow.codec.class_name
        training['codec']['init'] = repr(flow.codec)
        if hasattr(flow.codec, '_h'):
            training['codec']['args'] = flow.codec._h.to_json()
        if hasattr(flow.codec, 'performance'):
            training['codec']['performance'] = flow.codec.performance
    os.makedirs(root_dir, exist_ok=True)
    with open(os.path.join(root_dir, 'training.json'), 'w') as f:
        json.dump(training, f, indent=4, default=lambda o: float(o))
