#!/usr/bin/env python3
"""Fit T = rounds x (P + chunks x C) for the fused 5x5 convolution (conv + LReLU + pool, bf16 in / pooled bf16 out) by timing
it at several input-channel counts with everything else fixed.  Diagnostic only (HIP events on torch's stream)."""
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


def timed(run, reps=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


CASES = [(320, 64, 128), (320, 64, 64), (160, 64, 128), (320, 32, 256)]
if os.environ.get('FIT_QUICK'):
    CASES = [(320, 64, 128), (320, 32, 256)]
for (n, hw, cout) in CASES:
    rows = []
    for cin in ((64, 128) if os.environ.get('FIT_QUICK') else (16, 32, 64, 128, 256)):
        x = torch.randn((n, hw, hw, cin), device=dev).to(torch.bfloat16)
        w = torch.randn((5, 5, cin, cout), device=dev) * 0.05
        b = torch.zeros((cout,), device=dev)
        ms = timed(lambda: ops.conv2d_pool(x, w, b, out_bf16=True))
        fl = 2.0 * 25 * cin * cout * hw * hw * n
        rows.append((cin, ms))
        print('n %d @%d cout %d cin %3d: %.3f ms  %.0f TFLOP/s' % (n, hw, cout, cin, ms, fl / ms / 1e9), flush=True)
    wgs = n * (hw // 16) ** 2 * (cout // 64)
    rounds = wgs / 512.0
    (c0, t0), (c1, t1) = rows[1 if len(rows) > 2 else 0], rows[-1]
    per_chunk = (t1 - t0) / ((c1 - c0) / 16.0)
    fixed = t0 - per_chunk * (c0 / 16.0)
    print('  workgroups %d (%.1f rounds of 512): per chunk %.2f us/round, fixed %.2f us/round' %
          (wgs, rounds, 1e3 * per_chunk / rounds, 1e3 * fixed / rounds), flush=True)
