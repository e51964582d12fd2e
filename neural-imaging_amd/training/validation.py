"""
Validation passes and the training-progress record - device-side counterpart of the reference's training/validation.py.

The reference validates on the host: ten patches at a time, `.numpy()` after every batch, numpy loops over class pairs
(validate_fan :163-202), skimage per image (validate_nip :96-160, validate_dcn :44-93).  Here a validation pass queues
everything on the GPU - the channel forward, the decisions + confusion counts (`nimg_confusion_accumulate`), SSIM / PSNR
kernels - and reads the results back ONCE at the end of the pass.  What callers get is unchanged:

  validate_fan(flow, data[, get_labels]) -> (accuracy, confusion / n_patches[, predicted labels])       (:163-202)
  validate_nip(model, data, ...)         -> (ssims, psnrs, losses), one value per validation patch        (:96-160)
  validate_dcn(dcn, data, ...)           -> {'ssim', 'psnr', 'loss', 'entropy'} over the whole validation set (:44-93)
  validate_jpeg(jpeg, data[, batch_size])-> {'psnr', 'ssim', 'entropy'}: the differentiable JPEG codec (trainable tables) (:19-41)
  save_training_progress(...)            -> <root>/training.json with the reference's keys               (:301-352)
The matplotlib dashboards of the reference (figures per epoch) are not produced.
"""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

from .. import ops
from ..device import to_device


def _validation_batches(data):
    """(batch size, number of batches) of a validation pass: ten patches at a time, the tail dropped (:166-167)."""
    size = int(min(10, data.count_validation))
    return size, data.count_validation // size


def _first(batch):
    return batch[0] if isinstance(batch, tuple) else batch


def validate_fan(flow, data, get_labels=False):
    """Accuracy and confusion matrix of the channel's classifier on the validation patches.  Row = true class, column =
    decision, normalised by the number of validation patches, so every row sums to 1."""
    size, count = _validation_batches(data)
    k = flow.n_classes
    conf = torch.zeros((k, k), dtype=torch.int64, device=flow.device)
    truth = flow._device_labels(size)                   # [0]*size + [1]*size + ... (workflows/...:257-258), int32
    decisions = []
    for b in range(count):
        probs = flow.run_workflow(_first(data.next_validation_batch(b, size)))[-1].t
        pred = ops.confusion_accumulate(probs, truth, conf, want_pred=get_labels)
        if get_labels:
            decisions.append(pred)
    counts = conf.cpu().numpy().astype(np.float64)      # the one device-to-host copy of the pass
    patches = count * size
    accuracy = float(np.trace(counts) / (patches * k))  # = mean of the per-batch accuracies (equal batch sizes)
    if get_labels:
        return accuracy, counts / patches, torch.cat(decisions).cpu().numpy().tolist()
    return accuracy, counts / patches


def _psnr_device(a, b):
    d = a.double() - b.double()
    return 10.0 * torch.log10(1.0 / (d * d).reshape(d.shape[0], -1).mean(dim=1).clamp_min(1e-30))


def validate_nip(model, data, out_directory=None, savefig=False, epoch=0, show_ref=False, loss_type='L2'):
    """(ssims, psnrs, losses) of the developed validation patches, one value per patch: skimage-flavoured SSIM and PSNR of the
    clipped output, loss = mean (255 d)^2 (L2) or mean |255 d| (L1) per patch."""
    size, count = _validation_batches(data)
    ssims, psnrs, losses = [], [], []
    for b in range(count):
        bx, by = data.next_validation_batch(b, size)
        target = to_device(by, model.device)
        developed = model.process(bx).t.clamp(0.0, 1.0).contiguous()
        ssims.append(ops.ssim(developed, target, mode='skimage', max_val=1.0).double())
        psnrs.append(_psnr_device(developed, target))
        d = 255.0 * (developed.double() - target.double())
        per = (d * d) if loss_type == 'L2' else d.abs()
        losses.append(per.reshape(per.shape[0], -1).mean(dim=1))
    if not count:
        return [], [], []
    packed = torch.stack([torch.cat(ssims), torch.cat(psnrs), torch.cat(losses)]).cpu().numpy()
    return packed[0].tolist(), packed[1].tolist(), packed[2].tolist()


def validate_dcn(dcn, data, out_directory=None, savefig=False, epoch=0, show_ref=False):
    """The learned codec on the WHOLE validation set in one batch (the entropy is a batch statistic): mean SSIM, mean PSNR, the
    codec's training loss l2_loss + w H, and the entropy."""
    target = data.next_validation_batch(0, data.count_validation)
    target = to_device(target[-1] if isinstance(target, tuple) else target, dcn.device)
    y, ent = dcn.process(target, return_entropy=True)
    y = y.t
    ssim = ops.ssim(y.clamp(0.0, 1.0).contiguous(), target, mode='skimage', max_val=1.0).double().mean()
    psnr = _psnr_device(y, target).mean()
    l2 = ops.l2_loss(target, y)[0].double().reshape(())
    packed = torch.stack([ssim, psnr, l2, ent.t.double().reshape(())]).cpu().numpy()
    return {'ssim': float(packed[0]), 'psnr': float(packed[1]),
            'loss': float(packed[2] + dcn._h.entropy_weight * packed[3]), 'entropy': float(packed[3])}


def validate_jpeg(jpeg, data, batch_size=1):
    """Mean SSIM / PSNR of the differentiable JPEG codec over the validation images, `batch_size` at a time (the tail dropped);
    the entropy is NaN, as JPEG.process reports it (models/jpeg.py:245-249)."""
    from ..models.jpeg import JPEG
    if not isinstance(jpeg, JPEG):
        raise ValueError('Codec needs to be as instance of {} but is {}'.format(JPEG, getattr(jpeg, 'class_name', type(jpeg))))
    batch_size = int(min(batch_size, data.count_validation))
    ssims, psnrs, ents = [], [], []
    for b in range(data.count_validation // batch_size):
        bx = data.next_validation_batch(b, batch_size)
        target = to_device(bx[-1] if isinstance(bx, tuple) else bx, jpeg.device)
        y, ent = jpeg.process(target, return_entropy=True)
        y = y.t
        ssims.append(ops.ssim(y.clamp(0.0, 1.0).contiguous(), target, mode='skimage', max_val=1.0).double())
        psnrs.append(_psnr_device(y, target))
        ents.append(ent)
    if not ssims:
        return {'psnr': float('nan'), 'ssim': float('nan'), 'entropy': float('nan')}
    packed = torch.stack([torch.cat(ssims).mean(), torch.cat(psnrs).mean()]).cpu().numpy()
    return {'psnr': float(packed[1]), 'ssim': float(packed[0]), 'entropy': float(np.mean(ents))}


def _model_record(model, with_args=True):
    rec = OrderedDict([('model', model.class_name), ('init', repr(model))])
    if with_args:
        rec['args'] = model._h.to_json() if hasattr(model, '_h') else {}
    if hasattr(model, 'performance'):
        rec['performance'] = model.performance
    return rec


def save_training_progress(training_summary, flow, root_dir, quiet=False):
    """<root_dir>/training.json: summary, channel configuration, classes and one record per model (model / init / args /
    performance) under 'nip', 'forensics' and - when the channel has a codec - 'codec'."""
    record = OrderedDict([('summary', training_summary), ('distribution', flow._distribution),
                          ('manipulations', flow._forensics_classes), ('nip', _model_record(flow.nip)),
                          ('forensics', _model_record(flow.fan))])
    if flow.codec is not None:
        record['codec'] = _model_record(flow.codec, with_args=hasattr(flow.codec, '_h'))
    os.makedirs(root_dir, exist_ok=True)
    path = os.path.join(root_dir, 'training.json')
    with open(path, 'w') as f:
        json.dump(record, f, indent=4, default=float)
    if not quiet:
        print('> Training progress --> {}'.format(path))
    return path
