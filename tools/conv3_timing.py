"""Where the K loop of conv3_dma_kernel spends a wave's time (diagnostic build: tools/build_variant.sh timing "-DNIMG_CONV3_TIMING -DNIMG_CONV3_VARIANTS"
conv_bf16, then NIMG_LIBPATH=neural-imaging_amd/libnimg_timing.so python tools/conv3_timing.py [h cin cout]): s_memtime sums per
wave of workgroups 0 .. 63, in shader cycles (what s_memtime counts on gfx950; ~2.1 cycles per ns at the clock these kernels
sustain)."""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
dev = torch.device('cuda', 0)
_lib.load()
ops.set_compute('bf16')
h, cin, cout = (int(v) for v in (sys.argv[1:4] + ['8', '512', '512'][len(sys.argv) - 1:]))
x = torch.randn((64, h, h, cin), device=dev).to(torch.bfloat16)
w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
b = torch.zeros((cout,), device=dev)
y = torch.empty((64, h, h, cout), device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv2d(x, w, b, act='leaky_relu', out=y)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(64 * 4 * 8, np.uint64)
rc = lib.nimg_debug_conv3_timing(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
assert rc == 0, rc
t = buf.reshape(64, 4, 8).astype(np.float64)
tick_ns = 1.0             # unit = one s_memtime tick = one shader cycle
names = ['transfer issue', 'reads + matrix', 'wait transfers', 'barrier', 'prologue', 'epilogue', 'chunks', 'kernel']
chunks = t[0, 0, 6]
print('%d^2, %d -> %d: %d chunks; per wave, averaged over 64 workgroups (shader cycles)' % (h, cin, cout, chunks))
for wv in range(4):
    print('  wave %d: ' % wv + ' | '.join('%s %7.0f' % (names[k], t[:, wv, k].mean() * tick_ns) for k in (0, 1, 2, 3, 4, 5, 7)))
print('  per chunk (cycles): ' + ' | '.join('%s %6.1f' % (names[k], t[:, :, k].mean() * tick_ns / chunks) for k in range(4)))
