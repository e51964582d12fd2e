"""Stand-alone timing of the UNet's 3x3 forward layers (B = 64, bf16-stored tensors): HIP events, us per launch.
   python tools/unet_fwd_time.py [reps]        (NIMG_LIBPATH selects the library build)"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
LAYERS = [('ec12', 128, 32, 0, 32), ('dc41', 128, 32, 32, 32), ('ec21', 64, 32, 0, 64), ('ec22', 64, 64, 0, 64), ('dc31', 64, 64, 64, 64),
          ('ec31', 32, 64, 0, 128), ('ec32', 32, 128, 0, 128), ('dc21', 32, 128, 128, 128), ('ec41', 16, 128, 0, 256),
          ('ec42', 16, 256, 0, 256), ('dc11', 16, 256, 256, 256), ('ec51', 8, 256, 0, 512), ('ec52', 8, 512, 0, 512)]
tot = 0.0
for name, h, c1, c2, cout in LAYERS:
    x = torch.randn((64, h, h, c1), device=dev).to(bf)
    x2 = torch.randn((64, h, h, c2), device=dev).to(bf) if c2 else None
    w = torch.randn((3, 3, c1 + c2, cout), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    y = torch.empty((64, h, h, cout), device=dev, dtype=bf)
    fn = lambda: ops.conv2d(x, w, b, x2=x2, act='leaky_relu', out=y)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    tot += us
    print('{:5s} {:3d}+{:3d}->{:3d} @{:3d}^2: {:6.1f} us  {:6.1f} TFLOP/s'.format(name, c1, c2, cout, h, us, 2.0 * 9 * (c1 + c2) * cout * 64 * h * h / us / 1e6))
print('sum {:.1f} us'.format(tot))
