#!/usr/bin/env python3
"""Dump the fused conv + LReLU + pool output of the 5x5 ring kernels for fixed inputs (A/B of two library builds: run once per
NIMG_LIBPATH, then compare the dumps with `cmp`)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops

ops.set_compute('bf16')
dev = torch.device('cuda', 0)
gen = torch.Generator(device='cpu').manual_seed(7)
out = []
for (n, hw, cin, cout) in [(3, 64, 64, 128), (2, 32, 128, 256), (2, 128, 32, 64), (2, 48, 64, 64)]:
    x = torch.randn((n, hw, hw, cin), generator=gen).to(dev).to(torch.bfloat16)
    w = (torch.randn((5, 5, cin, cout), generator=gen) * 0.05).to(dev)
    b = torch.randn((cout,), generator=gen).to(dev)
    pooled, idx = ops.conv2d_pool(x, w, b, out_bf16=True)
    out += [pooled.view(torch.int16).cpu(), idx.cpu()]
torch.save(out, sys.argv[1])
print('saved', sys.argv[1], sum(t.numel() for t in out))
