"""Diagnostic: is the eager step deterministic, and where does a captured replay first differ from it (gradients or update)?"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import graphs, ops  # noqa: E402
from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification  # noqa: E402
from util import bayer_from_rgb, natural_images  # noqa: E402

dev = torch.device('cuda', 0)
ops.set_compute(sys.argv[1] if len(sys.argv) > 1 else 'f32')
dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
rgb = natural_images(2, 64, 64, seed=31)
raw = bayer_from_rgb(rgb)
bx, by = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)
kw = dict(lambda_nip=0.1, learning_rate=1e-3)


def make():
    return ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev,
                                      nan_check='deferred')


def state(wf):
    return [t.detach().cpu().numpy().copy() for t in (wf.fan._model.flat_grad, wf.nip._model.flat_grad, wf.fan._model.flat,
                                                       wf.nip._model.flat)]


def diff(a, b, what):
    names = ['fan grad', 'nip grad', 'fan params', 'nip params']
    for n, x, y in zip(names, a, b):
        d = np.abs(x - y)
        print('{:28s} {:10s} max|d| {:.3e}  (max|x| {:.3e})  differing {} / {}'.format(what, n, d.max(), np.abs(x).max(),
                                                                                  int((d > 0).sum()), d.size))


a, b = make(), make()
for _ in range(2):
    a.training_step(bx, by, **kw)
    b.training_step(bx, by, **kw)
diff(state(a), state(b), 'eager vs eager, 2 steps')
runner = graphs.CapturedStep(b, bx, by, learning_rate=1e-3, lambda_nip=0.1, warmup=1)
a.training_step(bx, by, **kw)          # a: 3 eager steps; b: 3 eager (2 + 1 warm-up) so far
diff(state(a), state(b), 'eager 3 vs eager 2+warm-up')
a.training_step(bx, by, **kw)
runner.step()
diff(state(a), state(b), 'eager step 4 vs replay')
a.training_step(bx, by, **kw)
runner.step()
diff(state(a), state(b), 'eager step 5 vs replay 2')
print('rate dev', float(runner._rate_dev.item()), 'expected', ops.adam_lr_t(1e-3, 5))
