"""Probe: do the UNet's 3x3 forward kernels fill the machine?  Each layer at B = 64 in one launch, at B = 32 in one launch, and as
two B = 32 launches on two streams at the same time (HIP events around both).   python tools/halfbatch_probe.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
dev = torch.device('cuda', 0)
_lib.load()
ops.set_compute('bf16')
bf = torch.bfloat16
LAYERS = [('ec12', 128, 32, 32), ('ec22', 64, 64, 64), ('ec31', 32, 64, 128), ('ec32', 32, 128, 128), ('ec41', 16, 128, 256),
          ('ec42', 16, 256, 256), ('ec51', 8, 256, 512), ('ec52', 8, 512, 512)]
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
reps = 30
tot = [0.0, 0.0, 0.0]
for name, h, cin, cout in LAYERS:
    x = torch.randn((64, h, h, cin), device=dev).to(bf)
    w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    y = torch.empty((64, h, h, cout), device=dev, dtype=bf)
    xa, xb, ya, yb = x[:32], x[32:], y[:32], y[32:]
    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    full = timed(lambda: ops.conv2d(x, w, b, act='leaky_relu', out=y))
    half = timed(lambda: ops.conv2d(xa, w, b, act='leaky_relu', out=ya))
    def both():
        cur = torch.cuda.current_stream(dev)
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ops.conv2d(xa, w, b, act='leaky_relu', out=ya)
        with torch.cuda.stream(s2):
            ops.conv2d(xb, w, b, act='leaky_relu', out=yb)
        cur.wait_stream(s1); cur.wait_stream(s2)
    two = timed(both)
    tot[0] += full; tot[1] += half; tot[2] += two
    print('{:5s} {:3d}->{:3d} @{:3d}^2: B=64 {:6.1f} us   B=32 {:6.1f} us   2 x B=32 on two streams {:6.1f} us'.format(name, cin, cout, h, full, half, two))
print('sum: {:.1f} / {:.1f} / {:.1f} us'.format(*tot))
