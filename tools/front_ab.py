#!/usr/bin/env python3
"""Timing of the FAN front-end kernels (3-channel side) in both compute modes."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=320)
args = ap.parse_args()
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
dev = torch.device('cuda', 0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n, h = args.images, 256
x = torch.rand((n, h, h, 3), device=dev)
for mode in ('f32', 'bf16'):
    ops.set_compute(mode)
    for cout in (8, 32, 64):
        w = torch.randn((5, 5, 3, cout), device=dev) * 0.1
        b = torch.zeros((cout,), device=dev)
        dz = torch.randn((n, h, h, cout), device=dev)
        tf = timed(lambda: ops.conv2d(x, w, b, act='leaky_relu'))
        tw = timed(lambda: ops.conv2d_wgrad(x, dz, 5))
        td = timed(lambda: ops.conv2d_dgrad(dz, w, (h, h))) if cout == 32 else float('nan')
        print('{} 3->{:2d}: fwd {:.3f} ms (out {:.2f} GB -> {:.2f} TB/s) | wgrad {:.3f} ms | dgrad {:.3f} ms'.format(
            mode, cout, tf, dz.numel() * 4 / 1e9, dz.numel() * 4 / 1e9 / tf, tw, td), flush=True)
    w3 = torch.randn((5, 5, 3, 3), device=dev)
    d3 = torch.randn((n, h, h, 3), device=dev)
    print('{} 3->3 : fwd(sym) {:.3f} ms | wgrad {:.3f} ms'.format(
        mode, timed(lambda: ops.conv2d(x, w3, None, pads=(2, 2), out_hw=(h, h), pad_mode=1)),
        timed(lambda: ops.conv2d_wgrad(x, d3, 5, pads=(2, 2), pad_mode=1))), flush=True)
    y = torch.randn((n, h, h, 32), device=dev)
    print('{} maxpool fwd {:.3f} ms | bwd {:.3f} ms'.format(mode, timed(lambda: ops.maxpool2(y)),
          timed(lambda: ops.maxpool2_bwd(torch.randn((n, h // 2, h // 2, 32), device=dev), y))), flush=True)
