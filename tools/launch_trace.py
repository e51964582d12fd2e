#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of `bench.py --steps 2 --warmup 1 --no-graph ...` -> one training step launch by launch:
start offset, duration, grid, queue (q1 = the launch stream, q2.. = side streams).   python tools/launch_trace.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'weights_bf16_batch_kernel' in r['Kernel_Name']]
adams = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = starts[-2], adams[-1]          # the last step: from its first weight-image launch (UNet) to its last Adam launch
qcol = 'Queue_Id' if 'Queue_Id' in rows[0] else None
queues = {}
t0 = int(rows[lo]['Start_Timestamp'])
busy = {}
for r in rows[lo:hi + 1]:
    q = r[qcol] if qcol else '0'
    name = queues.setdefault(q, 'q%d' % (len(queues) + 1))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy[name] = busy.get(name, 0) + (e - s)
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)[:84]
    print('%9.1f %8.1f us  grid %8s  %s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Grid_Size', '?'), name, n))
end = max(int(r['End_Timestamp']) for r in rows[lo:hi + 1])
print('# step: %.1f us wall; kernel time per queue: %s' % ((end - t0) / 1e3, ', '.join('%s %.0f us' % (k, v / 1e3) for k, v in busy.items())))
