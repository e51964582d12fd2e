"""Stand-alone timing of the FAN front-end kernels at the bench shapes (320 images of 256x256), HIP events on the launch stream.
   python tools/front_time.py [reps]      -> one line per kernel: ms per launch, GB/s against its algorithmic bytes"""
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
N, H, W = 320, 256, 256
x = torch.rand((N, H, W, 3), device=dev)
nf = torch.randn((5, 5, 3, 3), device=dev)
w1 = torch.randn((5, 5, 3, 32), device=dev) * 0.1
b1 = torch.zeros((32,), device=dev)
_, c4 = ops.cconv3(x, nf, pad_mode=1, want_f32=False, want_c4=True)
pooled, idx = ops.conv1_pool_c4(c4, w1, b1, out_bf16=True)
g = torch.randn((N, H // 2, W // 2, 32), device=dev).to(torch.bfloat16)
dw, db = torch.empty((5, 5, 3, 32), device=dev), torch.empty((32,), device=dev)
dc = torch.empty((N, H, W, 3), device=dev)
dx = torch.empty_like(x)
wt = ops.flip_weights(nf)
dnf = torch.empty((5, 5, 3, 3), device=dev)
MB = 1e6
cases = [
    ('cconv3 fwd (f32 -> c4)', lambda: ops.cconv3(x, nf, pad_mode=1, want_f32=False, want_c4=True), (12 + 8) * N * H * W),
    ('cconv3 dgrad main (f32 -> f32)', lambda: ops.cconv3(dc, wt, pad_mode=0, out=dx), 24 * N * H * W),
    ('cconv3 dgrad border', lambda: _lib.call('nimg_cconv3_dgrad_border', dc.data_ptr(), nf.data_ptr(), dx.data_ptr(), N, H, W,
                                              torch.cuda.current_stream().cuda_stream), 0),
    ('conv1 + lrelu + pool fwd', lambda: ops.conv1_pool_c4(c4, w1, b1, out_bf16=True), (8 + (64 + 32) / 4) * N * H * W),
    ('conv1 wgrad (pooled g)', lambda: ops.conv1_wgrad_c4(c4, g, idx, dw=dw, db=db), (8 + (64 + 32) / 4) * N * H * W),
    ('conv1 dgrad (pooled g)', lambda: ops.conv1_dgrad_pooled(g, idx, w1, out=dc), (12 + (64 + 32) / 4) * N * H * W),
    ('constrained filter wgrad', lambda: ops.conv2d_wgrad(x, dc, 5, pads=(2, 2), pad_mode=1, dw=dnf), 24 * N * H * W),
    ('un-pool conv2 (64ch @128^2)', None, 0),
]
gp2 = torch.randn((N, 64, 64, 64), device=dev).to(torch.bfloat16)
ix2 = torch.randint(0, 4, (N, 64, 64, 64), device=dev, dtype=torch.uint8)
cases[-1] = ('un-pool (64 ch, 64^2 -> 128^2, bf16)', lambda: ops.maxpool2_unpool(gp2, ix2, None, apply_mask=False, out_bf16=True),
             N * 64 * 64 * 64 * (2 + 1 + 8))
xj = torch.rand((N, H, W, 3), device=dev)
q = ops.qtables_device(80, dev)
yj, mj, _, _ = ops.djpeg_fwd(xj, q, 'soft', want_mask=True)
cases.append(('dJPEG fwd', lambda: ops.djpeg_fwd(xj, q, 'soft', want_mask=True), 25 * N * H * W))
cases.append(('dJPEG bwd', lambda: ops.djpeg_bwd(xj, dc, mj, q, 'soft'), 37 * N * H * W))
for name, fn, nbytes in cases:
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('{:40s} {:8.3f} ms   {:7.0f} GB/s (algorithmic {:.0f} MB)'.format(name, ms, nbytes / ms / 1e6, nbytes / MB))
