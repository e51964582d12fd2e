"""Diagnostic: NIP (UNet) pre-training loss curves in the compute modes, same initialisation, same batches.
   python tools/nip_diag.py [steps] [lr]"""
import importlib, os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import train_parity as tp
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import _lib, ops
_lib.load()
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
pool = tp.make_pool(512, 128, 7000, dev)
held = tp.make_pool(128, 128, 9000, dev)
variants = {'f32': ('f32', True), 'bf16': ('bf16', True), 'bf16f': ('bf16', False)}
for mode, store in [variants[v] for v in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['f32', 'bf16', 'bf16f'])]:
    ops.set_compute(mode)
    ops.STORE_BF16 = store
    wf = tp.make_flow(dev, 128)
    rng = np.random.RandomState(11)
    losses = []
    for s in range(steps):
        idx = torch.from_numpy(rng.choice(512, 64, replace=False)).to(dev)
        loss = wf.nip.training_step(pool[0][idx], pool[1][idx], learning_rate=lr)
        if s % max(steps // 20, 1) == 0 or s == steps - 1:
            losses.append(round(float(loss), 2))
    ev, _ = tp.evaluate(wf, held[0], held[1], 64)
    print(mode, 'store_bf16' if store else 'store_f32', 'psnr %.2f' % ev['isp_psnr_db'], 'mse255 at 20 points:', losses, flush=True)
ops.STORE_BF16 = True
