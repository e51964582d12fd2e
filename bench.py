#!/usr/bin/env python3
"""
bench.py - raw patches/s (forward + backward + Adam) of the imaging channel  UNet ISP -> manipulations -> dJPEG -> FAN
at 256x256 RGB (BASELINE.json metric; workload = configs[3] = SURVEY 8d "C4").

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = ManipulationClassification.training_step on a synthetic batch already resident in HBM: B raw patches
(B,128,128,4) -> UNet -> (5B,256,256,3) [native, sharpen:1, resample:50, gaussian:0.83, jpeg:80] -> dJPEG(80) -> FAN ->
CE + 0.1 * mse255 -> backward to the UNet and FAN weights -> gradient all-reduce (N>1) -> Keras Adam.  Weak scaling:
every rank processes its own B patches.  Prints ONE JSON line (rank 0).

Extra objects on the line:
  roofline     - dominant kernel (the FAN 5x5 convolution family) timed live with HIP events on the launch stream:
                 algorithmic FLOPs of one launch / average launch time, against the dense MFMA peak of the compute
                 dtype; traffic = HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_dominant_kernel.json:
                 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc runs), scaled to this launch's image count

--dtype bf16 (default) = throughput mode: bf16 MFMA operands, float32 accumulation, float32 tensors / master weights /
optimizer.  --dtype f32 = parity mode (exact float32 MFMA; the mode the 1e-4 parity tests run in); at N=1 the default run
also times a few parity-mode steps and reports them under config.f32_parity_mode.
  cpu_baseline - the oracle's CPU port (torch float32, all host cores) of the SAME step on a bounded sample
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

F32_MFMA_PEAK_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: bf16 MFMA dense (the 5 PF headline includes 2:1 sparsity)
GMAC_FWD_PER_PATCH = 3.079 + 5 * 2.705          # UNet + 5 x FAN, forward
GFLOP_PER_PATCH = 2 * 3 * GMAC_FWD_PER_PATCH    # fwd + dgrad + wgrad


def synthetic_batch(b, raw_patch, seed):
    from util import bayer_from_rgb, natural_images
    rgb = natural_images(b, 2 * raw_patch, 2 * raw_patch, seed=seed)
    return bayer_from_rgb(rgb), rgb


def time_dominant_kernel(dev, b_images, reps=20):
    """FAN conv3 forward (5x5, 64 -> 128 @ 64x64, the 839 MMAC/image layer) - one launch, HIP events on the
    stream it runs on (the kernels are launched on torch's current stream, so torch.cuda.Event brackets them)."""
    from neural_imaging_amd import ops
    n = b_images
    x = torch.randn((n, 64, 64, 64), device=dev)
    stored_bf16 = ops.COMPUTE == 'bf16' and ops.STORE_BF16
    if stored_bf16:                   # the FAN's pooled activations live in HBM as bf16 in throughput mode
        x = x.to(torch.bfloat16)
    w = torch.randn((5, 5, 64, 128), device=dev) * 0.05
    b = torch.zeros((128,), device=dev)
    out = torch.empty((n, 64, 64, 128), device=dev)
    # throughput mode: the op exactly as the FAN runs it - convolution + LeakyReLU + 2x2 max-pool in one kernel, bf16
    # pooled output + arg-max bytes (same kernel, same FLOPs; only the epilogue's store volume differs from the plain call)
    run = (lambda: ops.conv2d_pool(x, w, b, out_bf16=True)) if stored_bf16 else \
        (lambda: ops.conv2d(x, w, b, act='leaky_relu', out=out))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * 25 * 64 * 128 * 64 * 64 * n
    from neural_imaging_amd import ops as _o
    kname = 'conv_fwd_kernel<5,1,16,16,1,64,8>' if _o.COMPUTE == 'f32' else (
        'conv_fwd_bf16_kernel<5,1,16,16,1,64,INB=true,BUF=true>' if stored_bf16 else 'conv_fwd_bf16_kernel<5,1,16,16,1,64>')
    traffic = None
    try:                                                   # measured once with rocprofv3 --pmc, see profiles/README.md
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_dominant_kernel.json')) as f:
            pmc = json.load(f)['bf16_stored_input_pooled' if stored_bf16 else _o.COMPUTE]
        traffic = pmc['traffic_bytes_per_launch'] * n / pmc['images']
    except (OSError, KeyError, ValueError):
        pass
    return {'kernel': kname + ' (FAN conv3 fwd{}, {}x64x64x64->128)'.format(' + LReLU + pool' if stored_bf16 else '', n),
            'traffic': traffic,
            'flops_per_launch': flops, 'ms_per_launch': ms, 'tflops': flops / (ms * 1e-3) / 1e12}


def _usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))          # more threads than this only adds barrier overhead for these layer sizes


def cpu_baseline_worker(raw_patch, budget_s):
    """Oracle port on the host cores: torch float32 CPU, same step, B=2 raw patches, bounded to ~budget_s."""
    from oracle import workflow as owf
    cores = _usable_cores()
    torch.set_num_threads(cores)
    b = 2
    wf = owf.Workflow(trainable=('nip',), jpeg_quality=80, dtype=torch.float32)
    raw, rgb = synthetic_batch(b, raw_patch, seed=99)
    bx, by = torch.from_numpy(raw), torch.from_numpy(rgb)
    wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)       # warm-up
    t0, steps = time.time(), 0
    while steps < 5 and (time.time() - t0) < budget_s:
        wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)
        steps += 1
    dt = (time.time() - t0) / max(steps, 1)
    return {'value': b / dt, 'unit': 'patches/s', 'cores': cores, 'kind': 'port',
            'sample': '{} step(s) of B={} raw patches {}x{}x4, torch-CPU float32 restatement of the same step '
                      '(TF2 unavailable)'.format(steps, b, raw_patch, raw_patch)}


def cpu_baseline(raw_patch, budget_s=15.0, hard_timeout_s=120):
    """Run the CPU leg in a child process so a badly provisioned host can never stall the GPU measurement."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--raw-patch',
                              str(raw_patch), '--cpu-budget', str(budget_s)], capture_output=True, text=True,
                             timeout=hard_timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith('{'):
                return json.loads(ln)
        return {'value': None, 'unit': 'patches/s', 'cores': _usable_cores(), 'kind': 'port',
                'sample': 'cpu worker failed: ' + out.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'patches/s', 'cores': _usable_cores(), 'kind': 'port',
                'sample': 'cpu worker exceeded {} s'.format(hard_timeout_s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64, help='raw patches per GPU per step (SURVEY 8d C4: 64)')
    ap.add_argument('--raw-patch', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='bf16',
                    help='arithmetic type of the convolution GEMMs (bf16 = MFMA throughput mode, f32 accumulate; '
                         'f32 = parity mode)')
    ap.add_argument('--no-parity-mode', action='store_true', help='skip the extra float32 parity-mode timing at N=1')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default: nccl = RCCL); gloo is for a '
                                                     'functional check of the N>1 path on a one-GPU box')
    ap.add_argument('--single-device', action='store_true', help='functional check only: every rank uses cuda:0')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-budget', type=float, default=15.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(args.raw_patch, args.cpu_budget)))
        return

    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, parallel
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU; there is no CPU fallback')
    if args.single_device:
        os.environ['LOCAL_RANK'] = '0'
    world = parallel.init_from_env(args.backend)
    rank = parallel.rank()
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    _lib.load()
    from neural_imaging_amd import ops as _ops
    _ops.set_compute(args.dtype)
    if world != args.gpus and rank == 0:
        print('warning: --gpus {} but WORLD_SIZE {}'.format(args.gpus, world), file=sys.stderr)

    dist_cfg = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('UNet', manipulations=['sharpen:1', 'resample:50', 'gaussian:0.83', 'jpeg:80'],
                                    distribution=dist_cfg, trainable={'nip'}, raw_patch_size=args.raw_patch,
                                    device=dev, nan_check='deferred')
    raw, rgb = synthetic_batch(args.batch, args.raw_patch, seed=1234 + rank)
    bx, by = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, parts = wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)
    barrier()
    dt = time.perf_counter() - t0
    wf.check_nan()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = 1e3 * dt / args.steps
        value = world * args.batch * args.steps / dt
        dom = time_dominant_kernel(dev, 5 * args.batch)
        peak = F32_MFMA_PEAK_TFLOPS if args.dtype == 'f32' else BF16_MFMA_PEAK_TFLOPS
        line = {
            'metric': 'patches/s (fwd+bwd) ISP->JPEG->FAN @256^2', 'value': value, 'unit': 'patches/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic (natural-image-like RAW/RGB pairs, random-init weights)',
            'config': {'workload': 'train_manipulation UNet->[native,sharpen:1,resample:50,gaussian:0.83,jpeg:80]->'
                                   'dJPEG(QF80,soft)->FAN, ds none, trainable nip+fan, lambda_nip 0.1',
                       'raw_patch': args.raw_patch, 'rgb_patch': 2 * args.raw_patch,
                       'batch_per_gpu': args.batch, 'global_batch': world * args.batch, 'parallelism': 'dp%d' % world,
                       'loss': float(loss), 'achieved_tflops_whole_step': value * GFLOP_PER_PATCH / 1e3,
                       # SURVEY 8d compulsory-traffic model of the whole step against 8 TB/s: one kernel per reference op
                       # (343.8 MB bf16 / 687.5 MB f32 per raw patch) and the fused figure (141.9 / 283.8 MB)
                       'hbm_frac_whole_step_op_by_op': value / world * (343.8e6 if args.dtype == 'bf16' else 687.5e6) / 8e12,
                       'hbm_frac_whole_step_fused': value / world * (141.9e6 if args.dtype == 'bf16' else 283.8e6) / 8e12},
            'roofline': {'bound': 'mfma', 'achieved': dom['tflops'], 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': dom['tflops'] / peak, 'traffic': dom['traffic'], 'kernel': dom['kernel'],
                         'ms_per_launch': dom['ms_per_launch'], 'flops_per_launch': dom['flops_per_launch']},
        }
        if world == 1 and args.dtype == 'bf16' and not args.no_parity_mode:
            # PSNR / decision parity of the two compute modes on the same weights and batch (BASELINE metric: "PSNR/acc
            # parity"): ISP output and classifier decisions in throughput mode vs float32 mode
            out_b = wf.run_workflow(bx)
            _ops.set_compute('f32')                           # same step, exact float32 MFMA (the parity-test mode)
            out_f = wf.run_workflow(bx)
            yb, yf = out_b[0].t.float(), out_f[0].t.float()
            mse_modes = float(((yb - yf) ** 2).mean().item())
            agree = float((out_b[-1].t.argmax(dim=1) == out_f[-1].t.argmax(dim=1)).float().mean().item())
            mode_parity = {'isp_psnr_bf16_vs_f32_db': 10 * np.log10(1.0 / max(mse_modes, 1e-20)),
                           'fan_decision_agreement': agree,
                           'isp_psnr_vs_target_db': {'bf16': 10 * np.log10(1.0 / float(((yb - by) ** 2).mean().item())),
                                                     'f32': 10 * np.log10(1.0 / float(((yf - by) ** 2).mean().item()))}}
            wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-4)
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t1) / 3
            dom32 = time_dominant_kernel(dev, 5 * args.batch)
            line['config']['f32_parity_mode'] = {
                'patches_per_s': args.batch / dt32, 'ms_per_step': 1e3 * dt32,
                'dominant_kernel_tflops': dom32['tflops'], 'dominant_kernel_frac_of_f32_mfma_peak':
                    dom32['tflops'] / F32_MFMA_PEAK_TFLOPS, 'parity_of_modes': mode_parity}
            _ops.set_compute(args.dtype)
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(args.raw_patch)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
