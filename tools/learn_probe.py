#!/usr/bin/env python3
"""Does the channel learn? INet + FAN on synthetic patches, accuracy over steps. Diagnostic only."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
from util import bayer_from_rgb, natural_images
mode, lr = sys.argv[1], float(sys.argv[2])
ops.set_compute(mode)
dev = torch.device('cuda', 0)
dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
wf = ManipulationClassification(sys.argv[3] if len(sys.argv) > 3 else 'INet', manipulations=['sharpen:1', 'gaussian:1'], distribution=dist, trainable={'nip'},
                                raw_patch_size=32, device=dev)
rgb = natural_images(96, 64, 64, seed=100); raw = bayer_from_rgb(rgb)
vr = natural_images(32, 64, 64, seed=200); vraw = bayer_from_rgb(vr)
rng = np.random.RandomState(0)
t0 = time.time()
for step in range(301):
    idx = rng.choice(96, 8, replace=False)
    loss, parts = wf.training_step(raw[idx], rgb[idx], lambda_nip=0.1, learning_rate=lr)
    if step % 50 == 0:
        dec = wf.run_workflow_to_decisions(vraw)
        lab = np.repeat(np.arange(3), 32)
        print(step, 'loss %.3f ce %.3f acc %.3f  (%.1fs)' % (float(loss), float(parts['ce']), float(np.mean(np.asarray(dec) == lab)), time.time() - t0), flush=True)
