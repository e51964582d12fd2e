#!/usr/bin/env python3
"""Where the HOST time of an eager C4 step goes: cProfile over 20 steps of the bench workload (no graph), top functions by own time."""
import argparse, cProfile, importlib, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module('neural-imaging_amd')
import bench  # noqa: E402
from neural_imaging_amd import ops  # noqa: E402
ops.set_compute('bf16')

args = argparse.Namespace(workload='c4', batch=0, raw_patch=128, dtype='bf16', graph=False, seed=0)
for k, v in dict(gpus=1, steps=20, warmup=5).items():
    setattr(args, k, v)
dev = torch.device('cuda', 0)
wl = bench.WORKLOADS['c4'](args)
step = wl.build(dev, 0)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print('host time per step (launch only) %.2f ms, incl. GPU drain %.2f ms' % (1e3 * t_host, 1e3 * (time.perf_counter() - t0) / 20))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
