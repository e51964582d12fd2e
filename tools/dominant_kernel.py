#!/usr/bin/env python3
"""Launch only the dominant kernel (FAN conv3 forward, 5x5 64->128 @64x64, 320 images = B 64) a few times - the target
of the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) that feed bench.py's roofline.traffic."""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='f32')
ap.add_argument('--images', type=int, default=320)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--store-bf16', action='store_true', help='input stored as bf16 (the variant the FAN runs in throughput mode)')
ap.add_argument('--pool', action='store_true', help='the op as the FAN runs it: conv + LeakyReLU + 2x2 max-pool fused, bf16 pooled output + arg-max bytes')
args = ap.parse_args()
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
ops.set_compute(args.dtype)
dev = torch.device('cuda', 0)
x = torch.randn((args.images, 64, 64, 64), device=dev)
if args.store_bf16:
    x = x.to(torch.bfloat16)
w = torch.randn((5, 5, 64, 128), device=dev) * 0.05
b = torch.zeros((128,), device=dev)
out = torch.empty((args.images, 64, 64, 128), device=dev)
for _ in range(args.reps):
    if args.pool:
        pooled, idx = ops.conv2d_pool(x, w, b, out_bf16=args.store_bf16)
    else:
        ops.conv2d(x, w, b, act='leaky_relu', out=out)
torch.cuda.synchronize()
if args.pool:
    print('pooled output {:.1f} MB + arg-max {:.1f} MB'.format(pooled.numel() * pooled.element_size() / 1e6, idx.numel() / 1e6))
print('algorithmic bytes per launch: in {:.1f} MB + out {:.1f} MB + weights {:.2f} MB'.format(
    x.numel() * x.element_size() / 1e6, out.numel() * 4 / 1e6, w.numel() * 4 / 1e6))
