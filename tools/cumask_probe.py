#!/usr/bin/env python3
"""How does a CU-masked stream (nimg_stream_create_cu_mask: the low n bits) behave?  (1) ONE chip-filling kernel (FAN conv3 forward,
537 GFLOP, and a byte-bound dJPEG backward) on masked streams of n = 256 .. 32 CUs vs the default stream; (2) the same two kernels
CONCURRENTLY, matrix kernel on the masked stream and byte kernel on the default stream, against running them one behind the other.
    python tools/cumask_probe.py"""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops
    _lib.load()
    ops.set_compute('bf16')
    dev = torch.device('cuda', 0)
    n = 320
    x = torch.randn((n, 64, 64, 64), device=dev).to(torch.bfloat16)
    w = torch.randn((5, 5, 64, 128), device=dev) * 0.05
    b = torch.zeros((128,), device=dev)
    img = torch.rand((n, 256, 256, 3), device=dev)
    q = ops.qtables_device(80, dev)
    y, ctx = ops.djpeg_fwd(img, q, 'soft')[0], None
    dy = torch.randn_like(img)

    def mfma():
        ops.conv2d_pool(x, w, b, out_bf16=True)

    def bytes_():
        ops.add(img, dy, out=dy)

    def timed(fn, stream=None, reps=10):
        cur = torch.cuda.current_stream()
        s = stream or cur
        with torch.cuda.stream(s):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                fn()
            e1.record(s)
            torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    print('default stream: matrix kernel %.1f us, byte kernel %.1f us' % (timed(mfma), timed(bytes_)))
    plain = torch.cuda.Stream(device=dev)
    print('plain second stream: matrix kernel %.1f us, byte kernel %.1f us' % (timed(mfma, plain), timed(bytes_, plain)))
    streams = {}
    for cus in (256, 224, 192, 128, 64, 32):
        h = ctypes.c_void_p()
        _lib.call('nimg_stream_create_cu_mask', cus, ctypes.byref(h))
        streams[cus] = torch.cuda.ExternalStream(h.value, device=dev)
        print('masked stream, %3d CUs: matrix kernel %.1f us, byte kernel %.1f us' % (cus, timed(mfma, streams[cus]), timed(bytes_, streams[cus])))
    # concurrency: R repetitions of (matrix on side, byte x K on default), wall time by events on the default stream
    def both(side, reps=10, k=6):
        main_s = torch.cuda.current_stream()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for _ in range(reps):
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                mfma()
            for _ in range(k):
                bytes_()
            main_s.wait_stream(side)
        e1.record(main_s)
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    k = 6
    print('one behind the other: %.1f us' % (timed(mfma) + k * timed(bytes_)))
    print('concurrent, plain side stream: %.1f us' % both(plain, k=k))
    for cus in (224, 192, 128):
        print('concurrent, matrix kernel on %d CUs: %.1f us' % (cus, both(streams[cus], k=k)))


if __name__ == '__main__':
    main()
