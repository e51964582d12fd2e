"""Would two half-batch steps in flight at once beat one full-batch step?  (diagnostic)  Two INDEPENDENT C4 workflows of B = 32 raw
patches, each on its own launch stream, issued alternately, against one workflow of B = 64 on the default stream - the
latency-bound UNet phases of one half could run beside the chip-filling FAN phases of the other.  No gradient exchange between
the halves: an upper bound of what a micro-batched step could return.   python tools/microbatch_probe.py [steps]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
import bench  # noqa: E402
from neural_imaging_amd import _lib, ops  # noqa: E402
from neural_imaging_amd.graphs import RunAhead  # noqa: E402
from neural_imaging_amd.models import forensics  # noqa: E402
from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification  # noqa: E402

dev = torch.device('cuda', 0)
_lib.load()
ops.set_compute('bf16')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dist_cfg = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}


def flow():
    return ManipulationClassification('UNet', manipulations=bench.MANIPS, distribution=dist_cfg, trainable={'nip'}, raw_patch_size=128,
                                      device=dev, nan_check='deferred')


def batch(b, seed):
    raw, rgb = bench.synthetic_batch(b, 128, seed=seed)
    return torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)


def run(name, fns, n):
    for _ in range(5):
        for f in fns:
            f()
    torch.cuda.synchronize()
    pace = RunAhead(3)
    t = time.perf_counter()
    for _ in range(n):
        for f in fns:
            f()
        pace()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t) / n
    print('{:44s} {:7.3f} ms per 64 patches = {:6.0f} patches/s'.format(name, ms, 64e3 / ms))


forensics.LATE_MIN_IMAGES = int(os.environ.get('PROBE_LATE_MIN', '160'))
full = flow()
fx, fy = batch(64, 1234)
run('one step of B = 64', [lambda: full.training_step(fx, fy, lambda_nip=0.1, learning_rate=1e-4)], steps)

halves = [flow(), flow()]
data = [batch(32, 1234), batch(32, 4321)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def half(k):
    def f():
        with torch.cuda.stream(streams[k]):
            halves[k].training_step(data[k][0], data[k][1], lambda_nip=0.1, learning_rate=1e-4)
    return f


run('two steps of B = 32, one stream (serial)', [lambda: halves[0].training_step(*data[0], lambda_nip=0.1, learning_rate=1e-4),
                                                 lambda: halves[1].training_step(*data[1], lambda_nip=0.1, learning_rate=1e-4)], steps)
run('two steps of B = 32 on two launch streams', [half(0), half(1)], steps)
run('one step of B = 64 (again)', [lambda: full.training_step(fx, fy, lambda_nip=0.1, learning_rate=1e-4)], steps)
