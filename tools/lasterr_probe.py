import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
from util import natural_images
hip = ctypes.CDLL('libamdhip64.so')
hip.hipGetErrorName.restype = ctypes.c_char_p
def last(tag):
    e = hip.hipGetLastError()
    print(tag, e, hip.hipGetErrorName(e))
dev = torch.device('cuda', 0)
_lib.load()
last('after load')
x = natural_images(1, 64, 64, seed=1)
xt = torch.from_numpy(x).to(dev)
last('after H2D copy of a numpy-backed tensor')
q = ops.qtables_device(50, dev)
last('after qtables_device')
try:
    y = ops.djpeg_fwd(xt, q, 'soft', want_idx=True)
    print('djpeg ok')
except Exception as e:
    print('djpeg failed', e)
last('after djpeg')
y = ops.djpeg_fwd(xt, q, 'soft', want_idx=True)
print('second djpeg ok')
