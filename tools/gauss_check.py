#!/usr/bin/env python3
"""Dump gaussian and sharpen forward / backward outputs for fixed inputs (A/B of two library builds: run once per NIMG_LIBPATH, compare)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
from neural_imaging_amd.helpers import tf_helpers as th
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(3)
out = []
for (n, h, w) in [(2, 64, 64), (1, 50, 37), (3, 16, 16), (1, 256, 256)]:
    x = (torch.rand((n, h, w, 3), generator=gen) * 1.4 - 0.2).to(dev)
    op = th.Gaussian()
    y, ctx = op.forward(x, 0.83, training=True)
    dy = torch.randn((n, h, w, 3), generator=gen).to(dev)
    dx = op.backward(ctx, dy)
    out += [y.cpu(), dx.cpu()]
    sh = th.Sharpen()
    y2, ctx2 = sh.forward(x.clamp(0, 1), 1.0, training=True)
    out += [y2.cpu(), sh.backward(ctx2, dy).cpu()]
torch.save(out, sys.argv[1])
print('saved', sys.argv[1])
