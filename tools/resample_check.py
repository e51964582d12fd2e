#!/usr/bin/env python3
"""Dump resample forward / backward outputs for fixed inputs (A/B of the sparse-axis kernel forms: run once with and once
without NIMG_SPARSE_AXIS_SCALAR=1, compare the files)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd.helpers import tf_helpers as th
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(5)
out = []
for (n, h, w, f) in [(2, 64, 64, 50), (1, 48, 48, 50), (3, 20, 20, 75), (1, 256, 256, 50), (2, 128, 128, 30)]:
    x = torch.rand((n, h, w, 3), generator=gen).to(dev)
    op = th.Resample()
    y, ctx = op.forward(x, f, training=True)
    dy = torch.randn((n, h, w, 3), generator=gen).to(dev)
    out += [y.cpu(), op.backward(ctx, dy).cpu()]
torch.save(out, sys.argv[1])
print('saved', sys.argv[1])
