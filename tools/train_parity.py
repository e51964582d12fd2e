#!/usr/bin/env python3
"""
Learning parity of the two compute modes on a channel that has actually LEARNED (BASELINE.json
"PSNR/acc parity"; reference floors config/tests/framework.json:29-37).

Two copies of the C4 channel (UNet -> [native, sharpen:1, resample:50, gaussian:0.83, jpeg:80] -> dJPEG(80, soft) -> FAN) start
from ONE initialisation and see the SAME batches, one in throughput mode (bf16 MFMA operands, bf16-stored internal tensors),
one in float32 parity mode.  Recipe = the reference's own order: the NIP is pre-trained alone on its L2 loss
(train_nip.py -> training/pipeline.py), then the channel is trained jointly (train_manipulation.py, lambda_nip 0.1, nip + fan
trainable).  Every `--log-every` steps both are evaluated on held-out patches: FAN accuracy, CE, ISP PSNR.  At the end:
accuracy / PSNR deltas, decision agreement between the two trained channels, and the cross evaluation (the float32-trained
weights run through the bf16 kernels and vice versa).

    python tools/train_parity.py --steps 3000 --pretrain 1500 --out gpurun_out/r03/train_parity.json

The functions are also what bench.py's parity leg calls (shorter recipe, same code).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

MANIPS = ['sharpen:1', 'resample:50', 'gaussian:0.83', 'jpeg:80']


def make_flow(dev, raw_patch=128, seed=0):
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    torch.manual_seed(seed)
    dist_cfg = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    return ManipulationClassification('UNet', manipulations=MANIPS, distribution=dist_cfg, trainable={'nip'},
                                      raw_patch_size=raw_patch, device=dev, nan_check='deferred')


def make_pool(n, raw_patch, seed, dev, chunk=128):
    """n synthetic scenes resident in HBM: (raw (n,p,p,4), rgb (n,2p,2p,3))."""
    from util import bayer_from_rgb_t, scene_images
    rgbs = [scene_images(min(chunk, n - i), 2 * raw_patch, 2 * raw_patch, seed=seed + i, device=dev) for i in range(0, n, chunk)]
    rgb = torch.cat(rgbs, dim=0)
    return bayer_from_rgb_t(rgb), rgb


def evaluate(wf, raw, rgb, batch=64):
    """Held-out evaluation -> dict + the decision vector (n_classes * n,), ordered [batch][class][patch]."""
    n = raw.shape[0]
    mse, nll, hits, decs, count = 0.0, 0.0, 0, [], 0
    per_class = torch.zeros(wf.n_classes, device=raw.device)
    pair_hits = 0                    # first / last class ('native' / 'jpeg:80' in C4) taken as ONE class
    for i in range(0, n - batch + 1, batch):
        res = wf.run_workflow(raw[i:i + batch])
        Y, probs = res[0].t.float(), res[-1].t.float()
        labels = wf._device_labels(batch).long()
        d = probs.argmax(dim=1)
        decs.append(d)
        hits += int((d == labels).sum().item())
        per_class += (d == labels).float().view(wf.n_classes, batch).sum(dim=1)
        dm = d.view(wf.n_classes, batch)[[0, wf.n_classes - 1]]
        pair_hits += int(((dm == 0) | (dm == wf.n_classes - 1)).sum().item())
        nll += float((-probs.gather(1, labels[:, None]).clamp_min(1e-7).log()).sum().item())
        mse += float(((Y - rgb[i:i + batch]) ** 2).mean().item())
        count += 1
    nd = count * batch * wf.n_classes
    return {'fan_accuracy': hits / nd, 'ce': nll / nd, 'isp_psnr_db': float(10 * np.log10(1.0 / (mse / count))),
            'per_class_accuracy': [round(float(v), 4) for v in (per_class / (count * batch)).tolist()],
            'first_or_last_class_accuracy': pair_hits / (2.0 * count * batch)}, torch.cat(decs)


def pretrain_nip(wf, pool, steps, lr, batch, seed):
    raw, rgb = pool
    rng = np.random.RandomState(seed)
    for _ in range(steps):
        idx = torch.from_numpy(rng.choice(raw.shape[0], batch, replace=False)).to(raw.device)
        wf.nip.training_step(raw[idx], rgb[idx], learning_rate=lr)
    st = wf.nip._model
    st.m = st.v = None                      # the joint optimiser is a new Adam (workflows/manipulation_classification.py:150)
    st.step = 0


def train_joint(wf, pool, steps, lr, batch, seed, held=None, log_every=0, log=None, t0=None):
    raw, rgb = pool
    rng = np.random.RandomState(seed)
    traj = []
    for s in range(1, steps + 1):
        idx = torch.from_numpy(rng.choice(raw.shape[0], batch, replace=False)).to(raw.device)
        loss, parts = wf.training_step(raw[idx], rgb[idx], lambda_nip=0.1, learning_rate=lr)
        if log_every and (s % log_every == 0 or s == steps):
            ev, _ = evaluate(wf, held[0], held[1], batch)
            ev.update(step=s, train_ce=float(parts['ce']), train_nip=float(parts['nip']))
            if t0 is not None:
                ev['wall_s'] = round(time.time() - t0, 1)
            traj.append(ev)
            if log:
                log(ev)
    wf.check_nan()
    return traj


def copy_weights(src, dst):
    dst.nip._model.flat.copy_(src.nip._model.flat)
    dst.fan._model.flat.copy_(src.fan._model.flat)


def compare(wfs, held, batch):
    """wfs: {'bf16': flow trained in bf16 mode, 'f32': flow trained in f32 mode} -> parity dict (each trained channel evaluated
    in its own mode; decision agreement; then each set of weights through the OTHER mode's kernels)."""
    from neural_imaging_amd import ops
    out, dec = {}, {}
    for mode, wf in wfs.items():
        ops.set_compute(mode)
        out[mode], dec[mode] = evaluate(wf, held[0], held[1], batch)
    out['decision_agreement_of_the_two_trained_channels'] = float((dec['bf16'] == dec['f32']).float().mean().item())
    out['fan_accuracy_delta'] = out['bf16']['fan_accuracy'] - out['f32']['fan_accuracy']
    out['isp_psnr_delta_db'] = out['bf16']['isp_psnr_db'] - out['f32']['isp_psnr_db']
    # inference parity at ONE checkpoint: the float32-trained weights through the bf16 kernels
    ops.set_compute('bf16')
    ev, d = evaluate(wfs['f32'], held[0], held[1], batch)
    out['f32_weights_in_bf16_mode'] = dict(ev, decision_agreement_with_f32_mode=float((d == dec['f32']).float().mean().item()))
    ops.set_compute('f32')
    ev, d = evaluate(wfs['bf16'], held[0], held[1], batch)
    out['bf16_weights_in_f32_mode'] = dict(ev, decision_agreement_with_bf16_mode=float((d == dec['bf16']).float().mean().item()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--pretrain', type=int, default=1000)
    ap.add_argument('--pretrain-lr', type=float, default=1e-3)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--raw-patch', type=int, default=128)
    ap.add_argument('--pool', type=int, default=512)
    ap.add_argument('--heldout', type=int, default=256)
    ap.add_argument('--log-every', type=int, default=250)
    ap.add_argument('--modes', default='bf16,f32')
    ap.add_argument('--seed', type=int, default=12, help='seed of the batch order of the joint phase')
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops
    _lib.load()
    dev = torch.device('cuda', 0)
    pool = make_pool(args.pool, args.raw_patch, 7000, dev)
    held = make_pool(args.heldout, args.raw_patch, 9000, dev)
    res = {'recipe': vars(args), 'trajectory': {}}
    wfs = {}
    for mode in args.modes.split(','):
        ops.set_compute(mode)
        wf = make_flow(dev, args.raw_patch)
        t0 = time.time()
        pretrain_nip(wf, pool, args.pretrain, args.pretrain_lr, args.batch, seed=11)
        ev, _ = evaluate(wf, held[0], held[1], args.batch)
        ev.update(step=0, wall_s=round(time.time() - t0, 1))
        print(mode, 'after NIP pre-training', json.dumps(ev), flush=True)
        traj = [ev] + train_joint(wf, pool, args.steps, args.lr, args.batch, seed=args.seed, held=held, log_every=args.log_every,
                                  log=lambda e, m=mode: print(m, json.dumps(e), flush=True), t0=t0)
        res['trajectory'][mode] = traj
        wfs[mode] = wf
    if len(wfs) == 2:
        res['parity'] = compare(wfs, held, args.batch)
        print(json.dumps(res['parity'], indent=1))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
