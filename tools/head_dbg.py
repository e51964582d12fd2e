import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, '/root/repo')
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
_lib.load()
dev = torch.device('cuda', 0)
ops.set_compute('bf16')
n, hw, c = 3, (16, 16), 256
g = torch.Generator(device='cpu').manual_seed(3)
x = (torch.randn((n, hw[0], hw[1], c), generator=g) * 0.7).to(dev).to(torch.bfloat16)
w = (torch.randn((1, 1, c, c), generator=g) * (1.0 / np.sqrt(c))).to(dev)
b = (torch.randn((c,), generator=g) * 0.1).to(dev)
gap, mask = ops.head_fwd(x, w, b)
a = ops.conv2d(x, w, b, act='leaky_relu').float()
bits = ((mask.view(n, 256, c // 32, 1) >> torch.arange(32, device=dev, dtype=torch.int32)) & 1)
want = (a > 0).view(n, 256, c // 32, 32).to(torch.int32)
bad = (bits != want).any(dim=3).cpu().numpy()      # (n, px, f)
print('bad words per f:', bad.sum(axis=(0, 1)))
print('bad words per px%32:', bad.reshape(n, 8, 32, 8).sum(axis=(0, 1, 3)))
print('bad words per wave:', bad.reshape(n, 8, 32, 8).sum(axis=(0, 2, 3)))
m = mask.view(n, 256, 8).cpu().numpy()
print('zero words:', (m == 0).sum(), 'of', m.size)
# is a bad word equal to the expected word of some other (px, f)?
wantw = (want.cpu().numpy().astype(np.int64) << np.arange(32)).sum(axis=3).astype(np.uint32).view(np.int32)
i = np.argwhere(bad)[0]
print('first bad', i, hex(m[tuple(i)] & 0xffffffff), 'expected', hex(wantw[tuple(i)] & 0xffffffff))
row = wantw[i[0], (i[1] // 32) * 32:(i[1] // 32) * 32 + 32]
print('matches expected word at', np.argwhere(row == m[tuple(i)]))
