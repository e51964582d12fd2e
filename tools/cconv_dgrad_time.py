"""ConstrainedConv2D input gradient at the bench shape (320 x 256 x 256 x 3), throughput mode: matrix-core main term vs the float32
stencil, and the border term; HIP events.   python tools/cconv_dgrad_time.py [reps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
_lib.load(); ops.set_compute('bf16')
dev = torch.device('cuda', 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, H, W = 320, 256, 256
dc = torch.randn((N, H, W, 3), device=dev); nf = torch.randn((5, 5, 3, 3), device=dev)
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
st = torch.cuda.current_stream().cuda_stream
wt = ops.flip_weights(nf); dx = torch.empty_like(dc)
print('matrix-core main term: %.1f us' % timed(lambda: _lib.call('nimg_conv5c3_bf16', dc.data_ptr(), wt.data_ptr(), dx.data_ptr(), N, H, W, st)))
print('float32 stencil main term: %.1f us' % timed(lambda: ops.cconv3(dc, wt, pad_mode=0, out=dx)))
print('border term: %.1f us' % timed(lambda: _lib.call('nimg_cconv3_dgrad_border', dc.data_ptr(), nf.data_ptr(), dx.data_ptr(), N, H, W, st)))
