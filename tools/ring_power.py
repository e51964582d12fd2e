#!/usr/bin/env python3
"""Socket power / shader clock trace over back-to-back launches of the dominant kernel (conv5_ring_kernel<128>: FAN conv3 forward,
320 x 64 x 64 x 64 -> 128, conv + LeakyReLU + max-pool) - the evidence VERDICT r04 item 2(a) asks for behind the "power-bound"
reading of profiles/r04_b_ring_ablation.txt.

    python tools/ring_power.py [--launches 4000] [--out profiles/r05_ring_power] [lib.so | NAME=VALUE] ...

Per variant (own child process; default: the product library with random data, then with all-zero data - same instruction
stream, far fewer toggling bits): a sampler thread reads amdsmi (gpu_metrics: socket power, per-XCD gfx clock, throttle /
PPT-residency accumulators; power cap; violation status) every ~4 ms - sysfs hwmon as a fall-back - while the launch thread queues
the launches; HIP events time the launches.  Writes <out>.json (all samples) and <out>.txt (summary: idle / loaded power against
the cap, clock under load, kernel time, TFLOP/s)."""
import argparse
import glob
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOP = 2.0 * 320 * 64 * 64 * 25 * 64 * 128


class Sampler(threading.Thread):
    def __init__(self, period=0.004):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_flag, self.info = period, [], False, {}
        self.smi = self.h = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.smi, self.h = amdsmi, amdsmi.amdsmi_get_processor_handles()[0]
            try:
                self.info['power_cap'] = {k: (int(v) if isinstance(v, int) else str(v))
                                          for k, v in amdsmi.amdsmi_get_power_cap_info(self.h).items()}
            except Exception as e:                                        # noqa
                self.info['power_cap_error'] = repr(e)
        except Exception as e:                                            # noqa
            self.info['amdsmi_error'] = repr(e)
        self.hwmon = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_*'))
        self.info['hwmon'] = self.hwmon

    def one(self):
        s = {'t': time.perf_counter()}
        if self.smi is not None:
            try:
                m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
                for k in ('current_socket_power', 'average_socket_power', 'current_gfxclk', 'average_gfxclk_frequency',
                          'current_gfxclks', 'temperature_hotspot', 'throttle_status', 'indep_throttle_status',
                          'accumulation_counter', 'prochot_residency_acc', 'ppt_residency_acc', 'socket_thm_residency_acc',
                          'vr_thm_residency_acc', 'hbm_thm_residency_acc', 'gfxclk_lock_status', 'average_gfx_activity',
                          'energy_accumulator', 'system_clock_counter', 'firmware_timestamp'):
                    if k in m:
                        v = m[k]
                        s[k] = [x for x in v if isinstance(x, (int, float))][:8] if isinstance(v, (list, tuple)) else v
            except Exception as e:                                        # noqa
                s['metrics_error'] = repr(e)[:80]
            try:
                p = self.smi.amdsmi_get_power_info(self.h)
                s['power_info'] = {k: v for k, v in p.items() if isinstance(v, (int, float, str))}
            except Exception:                                             # noqa
                pass
        for f in self.hwmon:
            try:
                s[os.path.basename(f)] = int(open(f).read())
            except Exception:                                             # noqa
                pass
        return s

    def run(self):
        while not self.stop_flag:
            self.samples.append(self.one())
            time.sleep(self.period)

    def violations(self):
        if self.smi is None:
            return None
        try:
            v = self.smi.amdsmi_get_violation_status(self.h)
            return {k: (x if isinstance(x, (int, float, str)) else str(x)) for k, x in v.items()}
        except Exception as e:                                            # noqa
            return {'error': repr(e)[:120]}


def child(args):
    import torch
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops
    _lib.load()
    ops.set_compute('bf16')
    dev = torch.device('cuda', 0)
    mk = (lambda *s: torch.zeros(s, device=dev)) if args.zeros else (lambda *s: torch.randn(s, device=dev))
    x = mk(320, 64, 64, 64).to(torch.bfloat16)
    w = (torch.zeros if args.zeros else torch.randn)((5, 5, 64, 128), device=dev) * 0.05
    b = torch.zeros((128,), device=dev)
    fn = lambda: ops.conv2d_pool(x, w, b, out_bf16=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    smp = Sampler()
    v0 = smp.violations()
    smp.start()
    time.sleep(0.5)                                                       # idle stretch
    e = [torch.cuda.Event(enable_timing=True) for _ in range(args.launches // 200 + 1)]
    t_start = time.perf_counter()
    e[0].record()
    for i in range(args.launches):
        fn()
        if (i + 1) % 200 == 0:
            e[(i + 1) // 200].record()
            if (i + 1) % 800 == 0:
                e[(i + 1) // 200].synchronize()                           # keep the queue bounded (allocator) without draining it
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    time.sleep(0.3)
    smp.stop_flag = True
    smp.join()
    v1 = smp.violations()
    blocks = [e[k].elapsed_time(e[k + 1]) / 200 for k in range(len(e) - 1)]
    print('RING_POWER ' + json.dumps({'zeros': bool(args.zeros), 'launches': args.launches, 't_start': t_start, 't_end': t_end,
                                      'ms_per_launch_blocks': blocks, 'info': smp.info, 'violations_before': v0,
                                      'violations_after': v1, 'samples': smp.samples}), flush=True)


def summarise(tag, r):
    load = [s for s in r['samples'] if r['t_start'] + 0.15 <= s['t'] <= r['t_end'] - 0.02]
    idle = [s for s in r['samples'] if s['t'] < r['t_start'] - 0.05]

    def power(ss):
        for key, scale in (('current_socket_power', 1.0), ('average_socket_power', 1.0), ('power1_input', 1e-6),
                           ('power1_average', 1e-6)):
            v = [s[key] * scale for s in ss if isinstance(s.get(key), (int, float)) and 0 < s[key] * scale < 5000]
            if v:
                return key, sum(v) / len(v), max(v)
        return None, float('nan'), float('nan')

    def clock(ss):
        v = []
        for s in ss:
            c = s.get('current_gfxclks') or ([s['current_gfxclk']] if isinstance(s.get('current_gfxclk'), (int, float)) else [])
            c = [x for x in c if isinstance(x, (int, float)) and 0 < x < 10000]
            if c:
                v.append(sum(c) / len(c))
        return (sum(v) / len(v), min(v), max(v)) if v else (float('nan'),) * 3
    pk, pl, plmax = power(load)
    _, pi, _ = power(idle)
    cl = clock(load)
    ci = clock(idle)
    cap = r['info'].get('power_cap', {})
    capw = None
    for k in ('power_cap', 'default_power_cap', 'max_power_cap'):
        if isinstance(cap.get(k), int) and cap[k] > 0:
            capw = cap[k] / (1e6 if cap[k] > 100000 else 1.0)
            break
    ms = r['ms_per_launch_blocks']
    msm = sum(ms[1:]) / max(1, len(ms) - 1)
    ppt = None
    a = [s for s in load if isinstance(s.get('ppt_residency_acc'), int) and isinstance(s.get('accumulation_counter'), int)]
    if len(a) >= 2 and a[-1]['accumulation_counter'] > a[0]['accumulation_counter']:
        ppt = (a[-1]['ppt_residency_acc'] - a[0]['ppt_residency_acc']) / float(a[-1]['accumulation_counter'] - a[0]['accumulation_counter'])
    lines = ['== {} ({} data, {} launches, {} samples under load at {:.1f} ms cadence)'.format(
        tag, 'zero' if r['zeros'] else 'random', r['launches'], len(load),
        1e3 * (load[-1]['t'] - load[0]['t']) / max(1, len(load) - 1) if len(load) > 1 else float('nan')),
        '   kernel {:.4f} ms/launch = {:.0f} TFLOP/s = {:.3f} of 2.5 PF (first 200 launches {:.4f} ms)'.format(
            msm, FLOP / msm * 1e-9, FLOP / msm * 1e-9 / 2500, ms[0]),
        '   socket power [{}]: idle {:.0f} W, under load mean {:.0f} W / max {:.0f} W; cap {} W -> {:.1f} % of the cap'.format(
            pk, pi, pl, plmax, capw, 100 * pl / capw if capw else float('nan')),
        '   gfx clock: idle {:.0f} MHz, under load mean {:.0f} (min {:.0f}, max {:.0f}) MHz of 2400'.format(ci[0], *cl),
        '   PPT (power-limit) residency under load: {}'.format('n/a' if ppt is None else '{:.1f} % of the firmware samples'.format(100 * ppt)),
        '   violation status after: {}'.format({k: v for k, v in (r['violations_after'] or {}).items()
                                               if 'ppt' in k.lower() or 'acc_counter' in k.lower() or 'error' in k})]
    return '\n'.join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='*')
    ap.add_argument('--launches', type=int, default=4000)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_ring_power'))
    ap.add_argument('--zeros', action='store_true')
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child(args)
    variants = args.variants or ['product']
    allres, text = {}, []
    for v in variants:
        for zeros in (False, True):
            env = dict(os.environ)
            if v.endswith('.so'):
                env['NIMG_LIBPATH'] = os.path.abspath(v)
            elif '=' in v:
                env.update(dict(kv.split('=', 1) for kv in v.split(',')))
            cmd = [sys.executable, os.path.abspath(__file__), '--child', '--launches', str(args.launches)] + (['--zeros'] if zeros else [])
            out = subprocess.run(cmd, env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith('RING_POWER ')]
            tag = os.path.basename(v) + ('/zeros' if zeros else '/random')
            if not line:
                text.append('== {} FAILED: {}'.format(tag, out.stderr[-600:]))
                continue
            r = json.loads(line[0][11:])
            allres[tag] = r
            text.append(summarise(tag, r))
            print(text[-1], flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out + '.json', 'w') as f:
        json.dump(allres, f)
    with open(args.out + '.txt', 'w') as f:
        f.write('\n'.join(text) + '\n')


if __name__ == '__main__':
    main()
