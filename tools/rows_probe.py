#!/usr/bin/env python3
"""conv3_rows (csrc/conv3_rows.hip) against the tile kernels on the UNet's level-1 shapes: equality of the bits and time per launch.
    python tools/rows_probe.py [--batch 64] [--reps 30]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--reps', type=int, default=30)
args = ap.parse_args()
_lib.load()
ops.set_compute('bf16')
dev = torch.device('cuda', 0)
n, h = args.batch, 128
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.randn(s, device=dev, generator=g)


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


def rows(x, w, b, x2=None, mode=0, mask=None, pool=False, act=True):
    c1, c2 = x.shape[3], 0 if x2 is None else x2.shape[3]
    out = torch.empty((n, h, h, 32), dtype=torch.bfloat16, device=dev)
    po = torch.empty((n, h // 2, h // 2, 32), dtype=torch.bfloat16, device=dev) if pool else None
    wb = ops.weights_bf16(w, mode)
    P = lambda t: None if t is None else t.data_ptr()
    _lib.call('nimg_conv3_rows_bf16', P(x), c1, P(x2), c2, P(wb), P(b), P(mask), P(out), None, P(po), n, h, h, 32, 1 if act else 0, 0.2,
              0, ops._stream())
    return out, po


for name, c1, c2 in (('ec12 / dc42 (32 -> 32)', 32, 0), ('dc41 (32 + 32 -> 32)', 32, 32), ('one 64-channel input', 64, 0)):
    x = rnd(n, h, h, c1).to(torch.bfloat16)
    x2 = rnd(n, h, h, c2).to(torch.bfloat16) if c2 else None
    w = rnd(3, 3, c1 + c2, 32) * 0.1
    b = rnd(32) * 0.1
    ref = ops.conv2d(x, w, b, x2=x2, act='leaky_relu', out_bf16=True)
    got, _ = rows(x, w, b, x2=x2)
    same = torch.equal(ref.view(torch.int16), got.view(torch.int16))
    t_ref = timed(lambda: ops.conv2d(x, w, b, x2=x2, act='leaky_relu', out_bf16=True))
    t_new = timed(lambda: rows(x, w, b, x2=x2))
    byts = (n * h * h * (c1 + c2) + n * h * h * 32) * 2
    print('%-26s forward  : tile %6.1f us  rows %6.1f us (%4.2f TB/s)  identical %s  maxdiff %.3g' % (
        name, t_ref, t_new, byts / t_new * 1e-6, same, float((ref.float() - got.float()).abs().max())), flush=True)
    if c1 == 32 and c2 == 0:
        ra, rp = ops.conv2d_and_pool(x, w, b)
        ga, gp = rows(x, w, b, pool=True)
        t_ref = timed(lambda: ops.conv2d_and_pool(x, w, b))
        t_new = timed(lambda: rows(x, w, b, pool=True))
        print('%-26s + pool   : tile %6.1f us  rows %6.1f us  identical %s / %s' % (
            name, t_ref, t_new, torch.equal(ra.view(torch.int16), ga.view(torch.int16)), torch.equal(rp.view(torch.int16), gp.view(torch.int16))))
        # input gradient form: flipped / transposed weights, the previous layer's LeakyReLU' from its stored activation
        dz = rnd(n, h, h, 32).to(torch.bfloat16)
        act_prev = rnd(n, h, h, 32).to(torch.bfloat16)
        ref = ops.conv2d_dgrad(dz, w, (h, h), act_mask=act_prev, out_bf16=True)
        got, _ = rows(dz, w, None, mode=1, mask=act_prev, act=False)
        t_ref = timed(lambda: ops.conv2d_dgrad(dz, w, (h, h), act_mask=act_prev, out_bf16=True))
        t_new = timed(lambda: rows(dz, w, None, mode=1, mask=act_prev, act=False))
        print('%-26s dgrad    : tile %6.1f us  rows %6.1f us  identical %s' % (
            name, t_ref, t_new, torch.equal(ref.view(torch.int16), got.view(torch.int16))))
        # the same with a float32 result (ec12's input gradient feeds the 4-channel weight-gradient kernel, which stages float32)
        def via_ops(rows_on, f32):
            ops.ROWS_CONV = rows_on
            return ops.conv2d_dgrad(dz, w, (h, h), act_mask=act_prev, out_bf16=not f32)
        ref, got = via_ops(False, True), via_ops(True, True)
        print('%-26s dgrad f32: tile %6.1f us  rows %6.1f us  identical %s' % (
            name, timed(lambda: via_ops(False, True)), timed(lambda: via_ops(True, True)), torch.equal(ref, got)))
        # decoder layer's input gradient: 32 -> 32 + 32 channels as two tensors
        w2 = rnd(3, 3, 64, 32) * 0.1
        def two(rows_on):
            ops.ROWS_CONV = rows_on
            o1 = torch.empty((n, h, h, 32), dtype=torch.bfloat16, device=dev)
            o2 = torch.empty_like(o1)
            ops.conv2d_dgrad(dz, w2, (h, h), out=o1, out2=o2)
            return o1, o2
        (r1, r2), (g1, g2) = two(False), two(True)
        print('%-26s dgrad 2x : tile %6.1f us  rows %6.1f us  identical %s / %s' % (
            'dc41 (32 -> 32 + 32)', timed(lambda: two(False)), timed(lambda: two(True)),
            torch.equal(r1.view(torch.int16), g1.view(torch.int16)), torch.equal(r2.view(torch.int16), g2.view(torch.int16))))
        ops.ROWS_CONV = True
