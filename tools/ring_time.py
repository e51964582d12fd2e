#!/usr/bin/env python3
"""The six 5x5 launches of a 320-image FAN step, exactly as the FAN issues them in throughput mode - conv2 / conv3 / conv4 forward
(conv + LeakyReLU + 2x2 max-pool fused, bf16 in / pooled bf16 + arg-max out) and their input gradients from the pooled gradient
(un-pooling folded into the staging) - timed with HIP events on the launch stream, several libraries in turn:

    python tools/ring_time.py [--reps 20] [--rounds 2] [--zeros] lib_a.so lib_b.so NAME=VALUE ...   (default: the product library;
    NAME=VALUE = the product library with that environment variable set, e.g. an A/B switch of the dispatch)

Each library runs in its own child process (a process binds one libnimg.so); `--rounds` alternates them so that clock drift of
the box shows up as spread between rounds instead of as a difference between builds."""
import argparse
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = (('conv2', 128, 32, 64), ('conv3', 64, 64, 128), ('conv4', 32, 128, 256))


def child(reps, zeros, n):
    import torch
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops
    _lib.load()
    ops.set_compute('bf16')
    dev = torch.device('cuda', 0)
    res = {}

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    mk = (lambda *s: torch.zeros(s, device=dev)) if zeros else (lambda *s: torch.randn(s, device=dev))
    for name, h, cin, cout in SHAPES:
        x = mk(n, h, h, cin).to(torch.bfloat16)
        w = torch.randn((5, 5, cin, cout), device=dev) * 0.05
        b = torch.zeros((cout,), device=dev)
        res[name + '_fwd'] = timed(lambda: ops.conv2d_pool(x, w, b, out_bf16=True))
        g = mk(n, h // 2, h // 2, cout).to(torch.bfloat16)
        idx = torch.randint(0, 4, (n, h // 2, h // 2, cout), device=dev, dtype=torch.uint8)
        mask = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
        res[name + '_dgrad'] = timed(lambda: ops.conv2d_dgrad_unpool(g, idx, w, act_mask=mask, out_bf16=True))
    print('RING_TIME ' + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='*')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--images', type=int, default=320)
    ap.add_argument('--zeros', action='store_true')
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child(args.reps, args.zeros, args.images)
    libs = args.libs or [os.path.join(ROOT, 'neural-imaging_amd', 'libnimg.so')]
    gflop = 2.0 * 25 * 32 * 64 * 128 * 128 * args.images / 1e9             # the same for the three layers
    for rnd in range(args.rounds):
        for lib in libs:
            if '=' in lib and not lib.endswith('.so'):          # NAME=VALUE[,NAME=VALUE]: the product library under that environment
                env = dict(os.environ, **dict(kv.split('=', 1) for kv in lib.split(',')))
            else:
                env = dict(os.environ, NIMG_LIBPATH=os.path.abspath(lib))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--reps', str(args.reps), '--images',
                                  str(args.images)] + (['--zeros'] if args.zeros else []), env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith('RING_TIME ')]
            if not line:
                print('%-28s FAILED: %s' % (os.path.basename(lib), out.stderr[-300:]))
                continue
            r = json.loads(line[0][10:])
            tot = sum(r.values())
            print('%-28s ' % os.path.basename(lib) + ' '.join('%s %.0f' % (k.replace('conv', 'c').replace('_fwd', 'f').replace('_dgrad', 'd'), 1e3 * v)
                                                            for k, v in r.items()) +
                  ' | total %.3f ms = %.0f TFLOP/s' % (tot, 6 * gflop / tot), flush=True)


if __name__ == '__main__':
    main()
