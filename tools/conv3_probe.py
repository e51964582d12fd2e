"""Driver for counter passes over ONE UNet 3x3 forward layer (default ec42: 256 -> 256 @ 16x16, B = 64, bf16-stored tensors).
   python tools/conv3_probe.py [h cin cout reps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
dev = torch.device('cuda', 0)
_lib.load()
ops.set_compute('bf16')
h, cin, cout, reps = (int(v) for v in (sys.argv[1:5] + ['16', '256', '256', '5'][len(sys.argv) - 1:]))
x = torch.randn((64, h, h, cin), device=dev).to(torch.bfloat16)
w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
b = torch.zeros((cout,), device=dev)
y = torch.empty((64, h, h, cout), device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.conv2d(x, w, b, act='leaky_relu', out=y)
torch.cuda.synchronize()
print('done', float(y.float().abs().sum()))
