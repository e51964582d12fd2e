"""Fixed cost of one UNet 3x3 launch: time against the number of 16-channel K chunks at fixed output shape (B = 64, bf16-stored
tensors), back-to-back launches on one stream, HIP events.   python tools/conv3_intercept.py [reps]"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for h, cout in ((8, 512), (16, 256), (32, 128), (64, 64)):
    line = []
    for cin in (16, 32, 64, 128, 256, 512):
        x = torch.randn((64, h, h, cin), device=dev).to(bf)
        w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
        b = torch.zeros((cout,), device=dev)
        y = torch.empty((64, h, h, cout), device=dev, dtype=bf)
        fn = lambda: ops.conv2d(x, w, b, act='leaky_relu', out=y)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        line.append('%4d: %5.1f' % (cin, 1e3 * e0.elapsed_time(e1) / reps))
    print('%3d^2 -> %3d channels, us by Cin | ' % (h, cout) + ' | '.join(line), flush=True)
# an empty kernel pair for the launch floor
z = torch.zeros((1024,), device=dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    z.add_(1.0)
e1.record(); torch.cuda.synchronize()
print('trivial torch kernel back to back: %.1f us' % (1e3 * e0.elapsed_time(e1) / 200))
