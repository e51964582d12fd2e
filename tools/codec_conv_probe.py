"""The codec's 128 -> 128 3x3 layer at 64 x 64 (TwitterDCN residual blocks, models/compression.py:224-235), B images, bf16-stored:
us per launch of ops.conv2d under the current environment switches.   python tools/codec_conv_probe.py [B reps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
dev = torch.device('cuda', 0)
_lib.load()
ops.set_compute('bf16')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for h, cin, cout in ((64, 128, 128), (32, 128, 128), (128, 64, 64)):
    x = torch.randn((B, h, h, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    y = torch.empty((B, h, h, cout), device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.conv2d(x, w, b, act='leaky_relu', out=y)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print('B=%d %3d^2 %d->%d: %7.1f us  %6.1f TFLOP/s' % (B, h, cin, cout, us, 2.0 * 9 * cin * cout * B * h * h / us / 1e6), flush=True)
