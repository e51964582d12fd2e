#!/usr/bin/env python3
"""ConstrainedConv2D-class convolution (5x5, 3 -> 3, SYMMETRIC pad, 320 x 256 x 256) in isolation. Diagnostic only."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
dev = torch.device('cuda', 0)
x = torch.rand((320, 256, 256, 3), device=dev)
w = torch.randn((5, 5, 3, 3), device=dev) * 0.1
out = torch.empty_like(x)
for _ in range(2):
    ops.conv2d(x, w, pad_mode=1, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv2d(x, w, pad_mode=1, out=out)
e1.record()
torch.cuda.synchronize()
print('fewout 5x5 3->3 x320 @256: %.3f ms' % (e0.elapsed_time(e1) / 10))
