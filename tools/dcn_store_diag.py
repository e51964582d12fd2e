#!/usr/bin/env python3
"""Per-parameter relative difference of the TwitterDCN gradients between bf16 and float32 STORAGE (both on bf16 operands)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
from neural_imaging_amd.models import compression
from util import natural_images
dev = torch.device('cuda', 0)
x = torch.from_numpy(natural_images(2, 64, 64, seed=23)).to(dev)
ops.set_compute('bf16')
out, ctxs = {}, {}
for name, store, nochain in (('chain', True, False), ('nochain', True, True), ('f32store', False, False)):
    ops.STORE_BF16 = store
    if nochain:
        os.environ['NIMG_NO_S2D_CHAIN'] = '1'
    else:
        os.environ.pop('NIMG_NO_S2D_CHAIN', None)
    dcn = compression.TwitterDCN(patch_size=64, device=dev)
    y, ent, ctx = dcn.forward(x, training=True)
    _, dy = ops.l2_loss(x, y, grad_scale=1.0)
    dcn.backward(ctx, dy, entropy_coef=250.0)
    ops.join_side_stream()
    torch.cuda.synchronize()
    out[name] = {k: v.clone() for k, v in dcn._model.g.items()}
    ctxs[name] = {k: v.float().clone() for k, v in ctx[0].items() if isinstance(v, torch.Tensor)}
for k in ('e2', 'er1a', 'n1', 'n2', 'zl'):
    print('fwd', k, 'chain vs nochain', float((ctxs['chain'][k] - ctxs['nochain'][k]).abs().max()), ' nochain vs f32store',
          float((ctxs['nochain'][k] - ctxs['f32store'][k]).abs().max()))
for k, gb in out['f32store'].items():
    r1 = float((out['chain'][k] - out['nochain'][k]).abs().max()) / (float(gb.abs().max()) + 1e-30)
    r2 = float((out['nochain'][k] - gb).abs().max()) / (float(gb.abs().max()) + 1e-30)
    print('%-24s chain-vs-nochain %.3e   nochain-vs-f32store %.3e' % (k, r1, r2))
