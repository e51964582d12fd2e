#!/usr/bin/env python3
"""Three UNet training steps (throughput mode, B = 64 RAW 128 x 128 by default) for a kernel trace of ONE step, layer by layer:
    cd /tmp && NIMG_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/unet_one_step.py
    python tools/launch_trace.py <dir>/*/*_kernel_trace.csv"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
from neural_imaging_amd.models import pipelines
from util import bayer_from_rgb, natural_images

_lib.load()
ops.set_compute('bf16')
dev = torch.device('cuda', 0)
b, raw = int(os.environ.get('B', 64)), int(os.environ.get('RAW', 128))
net = pipelines.UNet(patch_size=raw, device=dev)
rgb = natural_images(b, 2 * raw, 2 * raw, seed=5)
x, tgt = torch.from_numpy(bayer_from_rgb(rgb)).to(dev), torch.from_numpy(rgb).to(dev)
for _ in range(3):
    net.training_step(x, tgt, learning_rate=1e-4)
    torch.cuda.synchronize()
