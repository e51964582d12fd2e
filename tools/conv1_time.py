#!/usr/bin/env python3
"""FAN conv1 trio (3 -> 32, 5x5, 256x256, 320 images) in isolation: fused conv+pool forward, pooled-gradient wgrad and
dgrad. Diagnostic only."""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--images', type=int, default=320)
args = ap.parse_args()
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
ops.set_compute(args.dtype)
dev = torch.device('cuda', 0)
n = args.images
x = torch.rand((n, 256, 256, 3), device=dev)
w = torch.randn((5, 5, 3, 32), device=dev) * 0.1
b = torch.zeros((32,), device=dev)
pooled, idx = ops.conv2d_pool(x, w, b)
g = torch.randn_like(pooled)
if args.dtype == 'bf16':
    g = g.to(torch.bfloat16)            # the product path hands the pooled gradient over as bf16
dw = torch.empty_like(w)
db = torch.empty_like(b)


def timed(name, fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('%-28s %.3f ms' % (name, e0.elapsed_time(e1) / args.reps))


timed('conv1 fwd + pool', lambda: ops.conv2d_pool(x, w, b))
timed('conv1 fwd (full res)', lambda: ops.conv2d(x, w, b, act='leaky_relu'))
if args.dtype == 'bf16':
    timed('conv1 wgrad (pooled g)', lambda: ops.conv2d_wgrad_pooled(x, g, idx, 5, dw=dw, db=db))
    timed('conv1 dgrad (pooled g)', lambda: ops.conv2d_dgrad_pooled(g, idx, w))
