#!/bin/bash
# session ao: per-launch kernel trace of one config-4 step (launch order, duration, grid, queue)
OUT=$PWD/gpurun_out/r03_ao; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof -o c4 --output-format csv -- python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 3 --warmup 2 --no-graph > $OUT/bench.json 2>$OUT/err.txt
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/c4_last_step.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# a step ends with its 2 adam launches: take the launches between the adam pair before last and the last pair
start, end = idx[-4] + 2 if len(idx) >= 4 else 0, idx[-1] + 1
# the roofline leg launches after the steps: restrict to the last training step = between adam groups
start = idx[-3] + 1
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:end]:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    print('%9.1f %8.1f us  grid %-8s q%-3s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
          r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Queue_Id', '?'), name))
PY
rm -rf $OUT/prof
wc -l $OUT/c4_last_step.txt
