import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
_lib.load()
dev = torch.device('cuda', 0)
n, h2, w2 = 3, 24, 40
g = torch.Generator(device='cpu').manual_seed(1)
parts = [torch.rand((n, h2, w2, 3), generator=g).to(dev) for _ in range(2)]
y, t = torch.rand((n, h2, w2, 3), generator=g).to(dev), torch.rand((n, h2, w2, 3), generator=g).to(dev)
s = ops.add_n(parts)
loss_ref, _ = ops.mse255(y, t, grad_scale=0.1, grad_out=s, accumulate=True)
dz_ref = ops.d2s_clip_bwd(s, 1.0)
loss, dz = ops.mse255_sum_s2d3(parts, y, t, 0.1)
bad = (dz != dz_ref).cpu().numpy()
print('loss', float(loss), float(loss_ref), 'bad frac', bad.mean())
print('bad by k', bad.reshape(-1, 12).mean(0))
print('bad by px', bad.mean(axis=(0, 1, 3)))
print('bad by row', bad.mean(axis=(0, 2, 3)))
