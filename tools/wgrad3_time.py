"""Stand-alone timing of the UNet's 3x3 weight gradients (B = 64 raw patches of 128x128, bf16-stored tensors) with the all-taps
kernel (csrc/wgrad3.hip) and with the tile-at-a-time kernel it replaces (NIMG_NO_WGRAD3_ALLTAPS=1, read per call): HIP events on
the launch stream, slab reduction included.   python tools/wgrad3_time.py [reps]"""
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
N = 64
LAYERS = [('ec12 / dc42', 128, 32, 0, 32), ('dc41', 128, 32, 32, 32), ('ec21', 64, 32, 0, 64), ('ec22 / dc32', 64, 64, 0, 64),
          ('dc31', 64, 64, 64, 64), ('ec31', 32, 64, 0, 128), ('ec32 / dc22', 32, 128, 0, 128), ('dc21', 32, 128, 128, 128),
          ('ec41', 16, 128, 0, 256), ('ec42 / dc12', 16, 256, 0, 256), ('dc11', 16, 256, 256, 256), ('ec51', 8, 256, 0, 512),
          ('ec52', 8, 512, 0, 512)]
tot = {'new': 0.0, 'old': 0.0}
for name, h, c1, c2, cout in LAYERS:
    x1 = torch.randn((N, h, h, c1), device=dev).to(torch.bfloat16)
    x2 = torch.randn((N, h, h, c2), device=dev).to(torch.bfloat16) if c2 else None
    dz = torch.randn((N, h, h, cout), device=dev).to(torch.bfloat16)
    dw, db = torch.empty((3, 3, c1 + c2, cout), device=dev), torch.empty((cout,), device=dev)
    res = {}
    for tag in ('new', 'old', 'new', 'old'):
        # 'old' = the comparison arm: NIMG_WGRAD3_ALT='NAME=VALUE' sets that variable for it (default: the tile-at-a-time kernel)
        alt = os.environ.get('NIMG_WGRAD3_ALT', 'NIMG_NO_WGRAD3_ALLTAPS=1').split('=')
        if tag == 'old':
            os.environ[alt[0]] = alt[1]
        else:
            os.environ.pop(alt[0], None)
        fn = lambda: ops.conv2d_wgrad(x1, dz, 3, x2=x2, dw=dw, db=db)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[tag] = (e0.elapsed_time(e1) / reps, float(dw.double().abs().sum()))
    os.environ.pop(alt[0], None)
    fl = 2.0 * 9 * (c1 + c2) * cout * N * h * h
    mb = N * h * h * (c1 + c2 + cout) * 2 / 1e6
    print('{:12s} {:3d}+{:3d}->{:3d} @{:3d}^2: new {:6.1f} us ({:6.1f} TFLOP/s, {:5.2f} TB/s)   old {:6.1f} us   checksums {:.5e} / {:.5e}'.format(
        name, c1, c2, cout, h, 1e3 * res['new'][0], fl / res['new'][0] / 1e9, mb / res['new'][0] / 1e3, 1e3 * res['old'][0],
        res['new'][1], res['old'][1]))
    for k in tot:
        tot[k] += res[k][0]
print('total: new {:.3f} ms, old {:.3f} ms'.format(tot['new'], tot['old']))
