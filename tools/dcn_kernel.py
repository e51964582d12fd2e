#!/usr/bin/env python3
"""Launch only the TwitterDCN residual-block convolution (3x3, 128 -> 128 @ 64x64; 12 of them are 7.2 of the codec's 9.74 GMAC
per image) a few times - the target of the rocprofv3 PMC passes behind bench.py's c3 / c5 roofline.traffic."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
ops.set_compute(sys.argv[2] if len(sys.argv) > 2 else 'bf16')
dev = torch.device('cuda', 0)
x = torch.randn((n, 64, 64, 128), device=dev)
w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
b = torch.zeros((128,), device=dev)
out = torch.empty((n, 64, 64, 128), device=dev)
for _ in range(3):
    ops.conv2d(x, w, b, act='leaky_relu', out=out)
torch.cuda.synchronize()
print('images {}; algorithmic bytes per launch: in {:.1f} MB + out {:.1f} MB'.format(n, x.numel() * 4 / 1e6, out.numel() * 4 / 1e6))
