#!/bin/bash
# session ax: the gap BETWEEN consecutive steps (last Adam launch -> first kernel of the next step) under graph replay and eager launches
OUT=$PWD/gpurun_out/r03_ax; mkdir -p $OUT
export TMPDIR=/tmp
for mode in --force-graph --no-graph; do
rocprofv3 --kernel-trace -d $OUT/prof -o c4 --output-format csv -- python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 6 --warmup 2 $mode > $OUT/bench$mode.json 2>$OUT/err.txt
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$f" "$mode" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# adam launches come in pairs (FAN, UNet); the second of a pair ends a step
ends = idx[1::2]
out = []
for a, b in zip(ends[:-1], ends[1:]):
    e_prev = int(rows[a]['End_Timestamp'])
    nxt = rows[a + 1]
    gap = (int(nxt['Start_Timestamp']) - e_prev) / 1e3
    span = (int(rows[b]['End_Timestamp']) - e_prev) / 1e3
    out.append((gap, span, nxt['Kernel_Name'][:40]))
print(sys.argv[2], 'per step: (gap after the previous Adam us, step period us, first kernel)')
for o in out: print('   %8.1f %9.1f  %s' % o)
PY
rm -rf $OUT/prof
done
