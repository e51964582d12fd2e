#!/bin/bash
# session al: per-launch kernel trace of one config-3 step (launch order, duration, grid)
OUT=$PWD/gpurun_out/r03_al; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof -o c3 --output-format csv -- python bench.py --workload c3 --no-cpu-baseline --no-parity-mode --steps 3 --warmup 2 --no-graph > $OUT/bench.json 2>$OUT/err.txt
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/c3_last_step.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = from the last s2d2_affine launch on
idx = [i for i, r in enumerate(rows) if 's2d2_affine' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    print('%9.1f %8.1f us  grid %-8s q%-3s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
          r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Queue_Id', '?'), name))
PY
rm -rf $OUT/prof
wc -l $OUT/c3_last_step.txt
