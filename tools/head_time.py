#!/usr/bin/env python3
"""The fused FAN head kernels (csrc/head.hip) against the generic kernels they replace, stand-alone, HIP events, 320 images of
16 x 16 x 256 (C4):   python tools/head_time.py [--images 320] [--reps 20]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=320)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--hw', type=int, default=16)
    ap.add_argument('--c', type=int, default=256)
    args = ap.parse_args()
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops
    _lib.load()
    ops.set_compute('bf16')
    dev = torch.device('cuda', 0)
    n, h, c, k = args.images, args.hw, args.c, 5
    x = (torch.randn((n, h, h, c), device=dev) * 0.7).to(torch.bfloat16)
    w = torch.randn((1, 1, c, c), device=dev) / np.sqrt(c)
    b = torch.randn((c,), device=dev) * 0.1
    wd = torch.randn((c, k), device=dev) * 0.1
    dlogits = torch.randn((n, k), device=dev) * 0.3
    labels = torch.zeros((n,), dtype=torch.int32, device=dev)

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / args.reps
    gap, mask, mask_p = ops.head_fwd(x, w, b)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    print('head_wgrad           %7.1f us' % timed(lambda: ops.head_wgrad(x, mask_p, dlogits, wd, dw.view(c, c), db)))
    dact = ops.head_dact(mask, dlogits, wd, x.shape)
    print('generic 1x1 wgrad    %7.1f us' % timed(lambda: ops.conv2d_wgrad(x, dact, 1, dw=dw, db=db)))
    print('head_fwd (mask)      %7.1f us' % timed(lambda: ops.head_fwd(x, w, b)))
    print('head_fwd (no mask)   %7.1f us' % timed(lambda: ops.head_fwd(x, w, b, want_mask=False)))
    print('head_dgrad (mask)    %7.1f us' % timed(lambda: ops.head_dgrad(mask, dlogits, wd, w, x, x.shape)))
    print('head_dgrad (no mask) %7.1f us' % timed(lambda: ops.head_dgrad(mask, dlogits, wd, w, None, x.shape)))
    print('head_dact            %7.1f us' % timed(lambda: ops.head_dact(mask, dlogits, wd, x.shape)))
    a = ops.conv2d(x, w, b, act='leaky_relu')
    print('generic 1x1 forward  %7.1f us' % timed(lambda: ops.conv2d(x, w, b, act='leaky_relu')))
    g2, probs, lp, dl = ops.fan_head_fwd(a, wd, b[:k].contiguous(), labels, 1.0 / n)
    print('generic gap + dense  %7.1f us' % timed(lambda: ops.fan_head_fwd(a, wd, b[:k].contiguous(), labels, 1.0 / n)))
    dwd, dbd = torch.empty_like(wd), torch.empty((k,), device=dev)
    dz, _ = ops.fan_head_bwd(a, g2, wd, dl, lp, 1.0 / n, dwd, dbd)
    print('generic head bwd     %7.1f us' % timed(lambda: ops.fan_head_bwd(a, g2, wd, dl, lp, 1.0 / n, dwd, dbd)))
    print('generic 1x1 dgrad    %7.1f us' % timed(lambda: ops.conv2d_dgrad(dz, w, (h, h), act_mask=x, out_bf16=True)))


if __name__ == '__main__':
    main()
