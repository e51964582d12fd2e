#!/usr/bin/env python3
"""
Root cause of the trained-parity slide: FAN accuracy at the bench's fixed checkpoint went
0.79 (r03) -> 0.71 (r04) -> 0.58 (r05), class 'sharpen' at 0.0.  This script runs the SAME seeded recipe

    (a) fully in float32 mode,  (b) fully in bf16 mode          - from identical initial weights, identical batches,

under several recipes, and logs the held-out trajectory every `--log-every` joint steps (accuracy, per-class accuracy, CE,
PSNR), so that "recipe" vs "precision" vs "kernel change" can be told apart:

    base      the bench's r03 - r05 recipe: NIP pre-training (1500 x lr 3e-4), then the joint phase at lr 1e-4
    warm:N    the same, with N FAN-only steps (NIP frozen, the reference's `--train` without nip; lr 1e-4) before the NIP joins
    seeds     the base recipe under other batch orders of the joint phase (is step 600 a coin flip?)

    python tools/parity_rootcause.py --recipes base,warm:300 --modes bf16,f32 --out gpurun_out/r06/parity_rootcause.json
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def run(tp, ops, dev, mode, recipe, pool, held, args, seed):
    # mode 'bf16s32': bf16 matrix operands, FAN- / UNet-internal tensors STORED as float32 (ops.STORE_BF16 off)
    ops.STORE_BF16 = mode != 'bf16s32'
    ops.set_compute('bf16' if mode.startswith('bf16') else mode)
    wf = tp.make_flow(dev, args.raw_patch)
    t0 = time.time()
    tp.pretrain_nip(wf, pool, args.pretrain, args.pretrain_lr, args.batch, seed=11)
    traj = []

    def log(e, phase):
        e = dict(e, phase=phase)
        traj.append(e)
        print(mode, recipe, seed, json.dumps(e), flush=True)

    ev, _ = tp.evaluate(wf, held[0], held[1], args.batch)
    ev.update(step=0, wall_s=round(time.time() - t0, 1))
    log(ev, 'pretrained')
    warm = int(recipe.split(':')[1]) if recipe.startswith('warm') else 0
    if warm:
        wf._trainable.discard('nip')
        tp.train_joint(wf, pool, warm, args.warm_lr, args.batch, seed=seed + 100, held=held, log_every=args.log_every,
                       log=lambda e: log(e, 'fan-only'), t0=t0)
        wf._trainable.add('nip')
    tp.train_joint(wf, pool, args.steps, args.lr, args.batch, seed=seed, held=held, log_every=args.log_every,
                   log=lambda e: log(e, 'joint'), t0=t0)
    return wf, traj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--recipes', default='base,warm:300')
    ap.add_argument('--modes', default='bf16,f32')
    ap.add_argument('--seeds', default='12')
    ap.add_argument('--steps', type=int, default=600)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--warm-lr', type=float, default=1e-4)
    ap.add_argument('--pretrain', type=int, default=1500)
    ap.add_argument('--pretrain-lr', type=float, default=3e-4)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--raw-patch', type=int, default=128)
    ap.add_argument('--log-every', type=int, default=50)
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops
    import train_parity as tp
    _lib.load()
    dev = torch.device('cuda', 0)
    pool = tp.make_pool(512, args.raw_patch, 7000, dev)
    held = tp.make_pool(256, args.raw_patch, 9000, dev)
    res = {'args': vars(args), 'runs': []}
    for recipe in args.recipes.split(','):
        for seed in [int(s) for s in args.seeds.split(',')]:
            wfs = {}
            entry = {'recipe': recipe, 'seed': seed, 'trajectory': {}}
            for mode in args.modes.split(','):
                wfs[mode], entry['trajectory'][mode] = run(tp, ops, dev, mode, recipe, pool, held, args, seed)
            if set(wfs) == {'bf16', 'f32'}:
                entry['parity'] = tp.compare(wfs, held, args.batch)
                print(recipe, seed, 'PARITY', json.dumps({k: v for k, v in entry['parity'].items() if not isinstance(v, dict)}),
                      flush=True)
            res['runs'].append(entry)
            del wfs
            torch.cuda.empty_cache()
            if args.out:
                os.makedirs(os.path.dirname(args.out), exist_ok=True)
                with open(args.out, 'w') as f:
                    json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
