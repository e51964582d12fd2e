"""Stand-alone timing of the FAN's three 5x5 input gradients from the pooled gradient (conv4 / conv3 / conv2 of a 320-image
step): sparse form (csrc/dgrad5s.hip, incl. its weight-image kernel) vs the ring kernels; HIP events on the launch stream.
   python tools/dgrad5_time.py [reps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
dev = torch.device('cuda', 0)
_lib.load()
ops.set_compute('bf16')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 320
for sparse in (False, True, False, True):
    ops.SPARSE_DGRAD = sparse
    tot = 0.0
    line = []
    for name, h, cin, cout in (('conv4', 32, 128, 256), ('conv3', 64, 64, 128), ('conv2', 128, 32, 64)):
        g = torch.randn((N, h // 2, h // 2, cout), device=dev).to(torch.bfloat16)
        idx = torch.randint(0, 4, (N, h // 2, h // 2, cout), device=dev, dtype=torch.uint8)
        w = torch.randn((5, 5, cin, cout), device=dev) * 0.05
        mask = torch.randn((N, h, h, cin), device=dev).to(torch.bfloat16)
        fn = lambda: ops.conv2d_dgrad_unpool(g, idx, w, act_mask=mask, out_bf16=True)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot += ms
        line.append('%s %.3f ms' % (name, ms))
    print('sparse' if sparse else 'ring  ', ' | '.join(line), '| total %.3f ms' % tot, flush=True)
