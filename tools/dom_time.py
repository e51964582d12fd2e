#!/usr/bin/env python3
"""Time the dominant convolution shapes in isolation (HIP events). Diagnostic only."""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--zeros', action='store_true')
ap.add_argument('--lib', default=None)
args = ap.parse_args()
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from neural_imaging_amd import ops
ops.set_compute(args.dtype)
dev = torch.device('cuda', 0)
for (n, hw, cin, cout, ks) in [(320, 128, 32, 64, 5), (320, 64, 64, 128, 5), (320, 32, 128, 256, 5), (64, 128, 64, 64, 3),
                               (64, 64, 128, 128, 3)]:
    x = torch.zeros((n, hw, hw, cin), device=dev) if args.zeros else torch.randn((n, hw, hw, cin), device=dev)
    w = torch.randn((ks, ks, cin, cout), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    out = torch.empty((n, hw, hw, cout), device=dev)
    wb = ops.weights_bf16(w, 0) if args.dtype == 'bf16' else None
    for _ in range(2):
        ops.conv2d(x, w, b, act='leaky_relu', out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ops.conv2d(x, w, b, act='leaky_relu', out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    fl = 2.0 * ks * ks * cin * cout * hw * hw * n
    print('conv %dx%d %d->%d @%d x%d: %.3f ms  %.0f TFLOP/s (incl. weight repack)' % (ks, ks, cin, cout, hw, n, ms, fl / ms / 1e9))
# conv2 dgrad class: 64 -> 32 channels at 128x128
for (n, hw, cin, cout, ks) in [(320, 128, 64, 32, 5)]:
    x = torch.randn((n, hw, hw, cin), device=dev)
    w = torch.randn((ks, ks, cin, cout), device=dev) * 0.05
    out = torch.empty((n, hw, hw, cout), device=dev)
    for _ in range(2):
        ops.conv2d(x, w, None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ops.conv2d(x, w, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    print('conv %dx%d %d->%d @%d x%d: %.3f ms' % (ks, ks, cin, cout, hw, n, ms))
