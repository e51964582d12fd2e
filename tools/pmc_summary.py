#!/usr/bin/env python3
"""rocprofv3 counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes of tools/pmc_collect.sh) -> JSON: per kernel, the HBM bytes
of its LAST dispatch: 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM) +
WRITE_SIZE, both in KB as reported."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for path in sorted(glob.glob(os.path.join(root, '*_SIZE.csv'))):
    driver, counter = os.path.basename(path)[:-4].split('_', 1)
    with open(path) as f:
        rows = list(csv.DictReader(f))
    if not rows:
        continue
    kcol = 'Kernel_Name' if 'Kernel_Name' in rows[0] else [c for c in rows[0] if 'ernel' in c and 'ame' in c][0]
    ncol = 'Counter_Name' if 'Counter_Name' in rows[0] else None
    vcol = 'Counter_Value' if 'Counter_Value' in rows[0] else None
    per = {}
    for r in rows:
        if ncol and r[ncol] != counter:
            continue
        v = float(r[vcol]) if vcol else float(r[counter])
        per.setdefault(r[kcol], []).append(v)
    for k, vals in per.items():
        e = out.setdefault(driver, {}).setdefault(k.replace('(anonymous namespace)::', '')[:110], {})
        e[counter + '_KB_last'] = vals[-1]
        e[counter + '_launches'] = len(vals)
for driver in out.values():
    for e in driver.values():
        f, w = e.get('FETCH_SIZE_KB_last'), e.get('WRITE_SIZE_KB_last')
        if f is not None and w is not None:
            e['traffic_bytes_per_launch'] = (2 * f + w) * 1024
print(json.dumps(out, indent=1))
