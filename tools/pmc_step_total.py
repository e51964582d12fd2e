#!/usr/bin/env python3
"""Counter-based HBM traffic of ONE whole training step.

Input: the two rocprofv3 counter_collection CSVs of `bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode`
collected in SEPARATE passes (`tools/gpu_check.sh <tag> pmc:FETCH_SIZE pmc:WRITE_SIZE`).  The last of the three steps is cut out
of the dispatch stream - from the UNet's weights_bf16_batch_kernel launch (the first kernel of a step) to the step's last
adam_kernel - and EVERY dispatch inside it is summed: bytes = 2 x FETCH_SIZE (gfx950 half-count correction of
MI355X_MICROARCH.md's HBM section; KB as reported) + WRITE_SIZE.  Per-kernel sums are kept so the total can be audited.

    python tools/pmc_step_total.py <FETCH csv> <WRITE csv> [batch [c4|c3|c5]] > profiles/r03_pmc_step_total.json
"""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_stamp import csrc_sha16


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)[:100]


def last_step(path, counter, workload='c4'):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    starts = [i for i, r in enumerate(rows) if 'weights_bf16_batch_kernel' in r['Kernel_Name']]
    adams = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    if workload != 'c4':
        # configs 3 / 5: a step = everything after the previous step's last Adam launch up to its own last Adam launch
        if len(adams) < 3 or len(adams) % 3:
            raise SystemExit('{}: expected 3 steps, found {} Adam launches'.format(path, len(adams)))
        aps = len(adams) // 3
        return rows[adams[2 * aps - 1] + 1:adams[3 * aps - 1] + 1]
    # two image launches per step (UNet first, FAN second), two Adam launches per step (FAN, UNet)
    if len(starts) < 6 or len(adams) < 6:
        raise SystemExit('{}: expected 3 steps (6 weight-image and 6 Adam launches), found {} / {}'.format(path, len(starts), len(adams)))
    lo, hi = starts[4], adams[5]
    return rows[lo:hi + 1]


def main():
    workload = sys.argv[4] if len(sys.argv) > 4 else 'c4'
    fetch, write = last_step(sys.argv[1], 'FETCH_SIZE', workload), last_step(sys.argv[2], 'WRITE_SIZE', workload)
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    if [short(r['Kernel_Name']) for r in fetch] != [short(r['Kernel_Name']) for r in write]:
        raise SystemExit('the two passes do not list the same dispatch sequence')
    per, total_f, total_w, dur = {}, 0.0, 0.0, 0.0
    for f, w in zip(fetch, write):
        k = short(f['Kernel_Name'])
        fb, wb = 2.0 * float(f['Counter_Value']) * 1024.0, float(w['Counter_Value']) * 1024.0
        us = (int(f['End_Timestamp']) - int(f['Start_Timestamp'])) / 1e3
        e = per.setdefault(k, {'launches': 0, 'fetch_bytes_x2': 0.0, 'write_bytes': 0.0, 'us_under_pmc': 0.0})
        e['launches'] += 1
        e['fetch_bytes_x2'] += fb
        e['write_bytes'] += wb
        e['us_under_pmc'] += us
        total_f += fb
        total_w += wb
        dur += us
    for e in per.values():
        e['bytes'] = e['fetch_bytes_x2'] + e['write_bytes']
        e['TBps_under_pmc'] = e['bytes'] / max(e['us_under_pmc'], 1e-9) / 1e6
    out = {'what': 'HBM bytes of one {} training step (all {} dispatches of the last of 3 eager steps), 2 x FETCH_SIZE + WRITE_SIZE, '
                   'separate rocprofv3 --pmc passes'.format(workload.upper(), len(fetch)),
           'csrc_sha16': csrc_sha16(), 'batch_raw_patches': batch, 'dispatches': len(fetch),
           'fetch_bytes_x2': total_f, 'write_bytes': total_w, 'bytes_per_step': total_f + total_w,
           'bytes_per_raw_patch': (total_f + total_w) / batch,
           'kernel_us_under_pmc': dur,
           'per_kernel': dict(sorted(per.items(), key=lambda kv: -kv[1]['bytes']))}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
