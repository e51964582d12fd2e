#!/usr/bin/env python3
"""UNet forward / backward of a training step, timed alone (HIP events on the launch stream; the backward pass joins its side
streams before the closing event), throughput mode, for several libraries / environment settings in turn:

    python tools/unet_time.py [--batch 64] [--raw 128] [--reps 20] [--rounds 2] [lib.so | NAME=VALUE[,NAME=VALUE]] ...

Each variant runs in its own child process (a process binds one libnimg.so and reads its A/B switches once)."""
import argparse
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def child(args):
    import torch
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops
    from neural_imaging_amd.models import pipelines
    from util import bayer_from_rgb, natural_images
    _lib.load()
    ops.set_compute('bf16')
    dev = torch.device('cuda', 0)
    net = pipelines.UNet(patch_size=args.raw, device=dev)
    rgb = natural_images(args.batch, 2 * args.raw, 2 * args.raw, seed=5)
    x, tgt = torch.from_numpy(bayer_from_rgb(rgb)).to(dev), torch.from_numpy(rgb).to(dev)

    def timed(fn, reps):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    state = {}

    def fwd():
        state['y'], state['ctx'] = net.forward(x, training=True)
    fwd()
    _, dy = ops.mse255(state['y'], tgt, grad_scale=1.0)
    res = {'fwd_us': timed(fwd, args.reps), 'bwd_us': timed(lambda: net.backward(state['ctx'], dy), args.reps),
           'step_us': timed(lambda: net.training_step(x, tgt, learning_rate=1e-4), args.reps)}
    print('UNET_TIME ' + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='*')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--raw', type=int, default=128)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child(args)
    variants = args.variants or [os.path.join(ROOT, 'neural-imaging_amd', 'libnimg.so')]
    for rnd in range(args.rounds):
        for v in variants:
            if '=' in v and not v.endswith('.so'):
                env = dict(os.environ, **dict(kv.split('=', 1) for kv in v.split(',')))
            else:
                env = dict(os.environ, NIMG_LIBPATH=os.path.abspath(v))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--batch', str(args.batch), '--raw',
                                  str(args.raw), '--reps', str(args.reps)], env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith('UNET_TIME ')]
            if not line:
                print('%-30s FAILED: %s' % (os.path.basename(v), out.stderr[-400:]))
                continue
            r = json.loads(line[0][10:])
            print('%-30s B=%d raw %d: forward %.0f us, backward %.0f us, training step %.0f us' % (
                os.path.basename(v), args.batch, args.raw, r['fwd_us'], r['bwd_us'], r['step_us']), flush=True)


if __name__ == '__main__':
    main()
