#!/usr/bin/env python3
"""A/B timing of single conv launches (FAN conv2/3/4 shapes): fwd, dgrad, wgrad, in the selected compute mode."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--images', type=int, default=320)
args = ap.parse_args()
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
ops.set_compute(args.dtype)
dev = torch.device('cuda', 0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n = args.images
for (h, cin, cout, ks) in [(128, 32, 64, 5), (64, 64, 128, 5), (32, 128, 256, 5), (128, 32, 32, 3), (64, 64, 64, 3), (16, 256, 256, 3)]:
    nn = n if ks == 5 else n // 5
    x = torch.randn((nn, h, h, cin), device=dev); w = torch.randn((ks, ks, cin, cout), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev); dz = torch.randn((nn, h, h, cout), device=dev)
    fl = 2.0 * ks * ks * cin * cout * h * h * nn / 1e12
    tf = timed(lambda: ops.conv2d(x, w, b, act='leaky_relu'))
    td = timed(lambda: ops.conv2d_dgrad(dz, w, (h, h)))
    tw = timed(lambda: ops.conv2d_wgrad(x, dz, ks))
    print('{:3d}x{}^2 {:3d}->{:3d} k{} | fwd {:6.3f} ms {:6.1f} TF | dgrad {:6.3f} ms {:6.1f} TF | wgrad {:6.3f} ms {:6.1f} TF'.format(
        nn, h, cin, cout, ks, tf, fl / tf * 1e3, td, fl / td * 1e3, tw, fl / tw * 1e3), flush=True)
