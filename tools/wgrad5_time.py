"""Stand-alone timing of the FAN's 5x5 weight gradients (conv2 / conv3 / conv4 of a 320-image step) from the pooled gradient:
HIP events on the launch stream, TFLOP/s against 2 x 25 x Cin x Cout x pixels.  The kernel behind the call is chosen by the
environment (read once per process): default = csrc/wgrad5.hip (all taps in one wave), NIMG_NO_WGRAD5_ALLTAPS=1 = the 8-wave
conv_wgrad_bf16_kernel<5,...>, NIMG_WGRAD5_TH8 / NIMG_WGRAD5_KX3L / NIMG_WGRAD5_ALLTAPS_BLOCKS = variants.
   python tools/wgrad5_time.py [reps]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda', 0)
_lib.load()
ops.set_compute('bf16')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 320
tot = 0.0
for name, h, cin, cout in (('conv2', 128, 32, 64), ('conv3', 64, 64, 128), ('conv4', 32, 128, 256)):
    x = torch.randn((N, h, h, cin), device=dev).to(torch.bfloat16)
    g = torch.randn((N, h // 2, h // 2, cout), device=dev).to(torch.bfloat16)
    idx = torch.randint(0, 4, (N, h // 2, h // 2, cout), device=dev, dtype=torch.uint8)
    dw, db = torch.empty((5, 5, cin, cout), device=dev), torch.empty((cout,), device=dev)
    fn = lambda: ops.conv2d_wgrad_unpool(x, g, idx, 5, dw, db=db)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    fl = 2.0 * 25 * cin * cout * N * h * h
    print('{} wgrad (+ slab reduction) {:4d}->{:4d} @{:3d}^2: {:7.3f} ms  {:7.1f} TFLOP/s  checksum {:.6e}'.format(
        name, cin, cout, h, ms, fl / ms / 1e9, float(dw.double().abs().sum())))
print('total {:.3f} ms'.format(tot))
