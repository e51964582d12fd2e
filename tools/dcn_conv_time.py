"""Stand-alone timing of the codec's dominant layer class (3x3, 128 -> 128 @64x64, float32-stored tensors) at several batch
sizes: forward (+ LeakyReLU), input gradient (+ mask), weight gradient; HIP events on the launch stream.
   python tools/dcn_conv_time.py [reps]"""
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
store = os.environ.get('DCN_STORE', 'f32')


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


c, hw = 128, 64
w = torch.randn((3, 3, c, c), device=dev) * 0.05
b = torch.zeros((c,), device=dev)
dw, db = torch.empty_like(w), torch.empty_like(b)
for n in (16, 48, 50, 64, 80):
    x = torch.randn((n, hw, hw, c), device=dev)
    dz = torch.randn((n, hw, hw, c), device=dev)
    if store == 'bf16':
        x, dz = x.to(torch.bfloat16), dz.to(torch.bfloat16)
    y = torch.empty((n, hw, hw, c), device=dev, dtype=x.dtype)
    fl = 2.0 * 9 * c * c * hw * hw * n
    t_f = timed(lambda: ops.conv2d(x, w, b, act='leaky_relu', out=y))
    t_d = timed(lambda: ops.conv2d_dgrad(dz, w, (hw, hw), act_mask=x, out=y))
    t_w = timed(lambda: ops.conv2d_wgrad(x, dz, 3, dw=dw, db=db))
    print('B %3d (%s): fwd %6.1f us %5.0f TF/s | dgrad %6.1f us %5.0f TF/s | wgrad %6.1f us %5.0f TF/s | per image %.2f us' % (
        n, store, 1e3 * t_f, fl / t_f / 1e9, 1e3 * t_d, fl / t_d / 1e9, 1e3 * t_w, fl / t_w / 1e9, 1e3 * (t_f + t_d + t_w) / n), flush=True)
