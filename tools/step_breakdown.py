#!/usr/bin/env python3
"""Per-phase timing of one workflow training step (torch events on the launch stream). Diagnostic only."""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--raw-patch', type=int, default=128)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--dtype', default='f32')
    args = ap.parse_args()
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import ops
    ops.set_compute(args.dtype)
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    from util import bayer_from_rgb, natural_images
    dev = torch.device('cuda', 0)
    dist_cfg = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('UNet', distribution=dist_cfg, trainable={'nip'}, raw_patch_size=args.raw_patch,
                                    device=dev, nan_check='deferred')
    rgb = natural_images(args.batch, 2 * args.raw_patch, 2 * args.raw_patch, seed=1)
    x = torch.from_numpy(bayer_from_rgb(rgb)).to(dev)
    t = torch.from_numpy(rgb).to(dev)
    b = args.batch

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        print('{:28s} {:9.2f} ms'.format(name, 1e3 * (time.perf_counter() - t0)), flush=True)
        return out

    for rep in range(args.reps):
        print('--- rep', rep, flush=True)
        Y, nctx = timed('unet fwd', lambda: wf.nip.forward(x, training=True))
        m, mctxs = timed('manipulations fwd', lambda: wf._manipulations(Y, False, training=True))
        C, cctx = timed('djpeg fwd', lambda: wf.codec.forward(m, training=True))
        _, fctx = timed('fan fwd', lambda: wf.fan.forward(C, wf._device_labels(b), training=True))
        loss_ce, dC = timed('fan bwd', lambda: wf.fan.backward(fctx, need_input_grad=True))
        dc = timed('djpeg bwd', lambda: wf.codec.backward(cctx, dC))

        def manip_bwd():
            dY = dc[:b].clone()
            for k, (name, op) in enumerate(wf._operations.items()):
                ops.add(dY, op.backward(mctxs[k], dc[(k + 1) * b:(k + 2) * b]), out=dY)
            ops.mse255(Y, t, grad_scale=0.1, grad_out=dY, accumulate=True)
            return dY
        dY = timed('manipulations bwd', manip_bwd)
        timed('unet bwd', lambda: wf.nip.backward(nctx, dY))
        timed('adam x2', lambda: (wf.fan._model.adam(1e-4, rep + 1), wf.nip._model.adam(1e-4, rep + 1)))
    # finer: FAN layers
    print('--- FAN layer timing (fwd conv / dgrad / wgrad)', flush=True)
    n = 5 * b
    for (h, cin, cout, ks) in [(256, 3, 3, 5), (256, 3, 32, 5), (128, 32, 64, 5), (64, 64, 128, 5), (32, 128, 256, 5),
                               (16, 256, 256, 1)]:
        xx = torch.randn((n, h, h, cin), device=dev)
        w = torch.randn((ks, ks, cin, cout), device=dev) * 0.05
        bb = torch.zeros((cout,), device=dev)
        dz = torch.randn((n, h, h, cout), device=dev)
        tag = '{}x{}x{} {}->{} k{}'.format(n, h, h, cin, cout, ks)
        timed('fwd   ' + tag, lambda: ops.conv2d(xx, w, bb, act='leaky_relu'))
        timed('dgrad ' + tag, lambda: ops.conv2d_dgrad(dz, w, (h, h)))
        timed('wgrad ' + tag, lambda: ops.conv2d_wgrad(xx, dz, ks))
        timed('bgrad ' + tag, lambda: ops.bias_grad(dz))


if __name__ == '__main__':
    main()
