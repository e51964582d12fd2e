"""Would the UNet forward / backward of a step run faster as two half-batch chains on two launch streams?  (diagnostic; the deep
levels cost the same at half the batch: profiles/r03_k_halfbatch_probe.txt).  One UNet (shared weights), B = 64 on one stream
against its two halves of 32 on two streams, HIP events around fork ... join.   python tools/unet_split_probe.py [reps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
from neural_imaging_amd.models import pipelines
from util import bayer_from_rgb, natural_images
_lib.load()
ops.set_compute('bf16')
dev = torch.device('cuda', 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, RAW = 64, 128
net = pipelines.UNet(patch_size=RAW, device=dev)
rgb = natural_images(B, 2 * RAW, 2 * RAW, seed=5)
x, tgt = torch.from_numpy(bayer_from_rgb(rgb)).to(dev), torch.from_numpy(rgb).to(dev)
xs = [x[:B // 2].contiguous(), x[B // 2:].contiguous()]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
main = torch.cuda.current_stream(dev)
state = {}


def full():
    state['y'], state['ctx'] = net.forward(x, training=True)


def split():
    ev = torch.cuda.Event()
    ev.record(main)
    for k in range(2):
        streams[k].wait_event(ev)
        with torch.cuda.stream(streams[k]):
            state['y%d' % k], state['ctx%d' % k] = net.forward(xs[k], training=True)
        e = torch.cuda.Event()
        e.record(streams[k])
        main.wait_event(e)


def serial_halves():
    for k in range(2):
        state['y%d' % k], state['ctx%d' % k] = net.forward(xs[k], training=True)


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for rnd in range(2):
    print('forward B = 64, one stream            %7.1f us' % timed(full))
    print('forward 2 x B = 32, one stream        %7.1f us' % timed(serial_halves))
    print('forward 2 x B = 32, two streams       %7.1f us' % timed(split), flush=True)
full()
split()
torch.cuda.synchronize()
y = torch.cat([state['y0'], state['y1']])
print('same result:', bool(torch.equal(y, state['y'])))
