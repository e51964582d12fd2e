#!/bin/bash
# session au: per-launch kernel trace of one config-4 step under GRAPH REPLAY (to compare with r03_ao = eager launches)
OUT=$PWD/gpurun_out/r03_au; mkdir -p $OUT
export TMPDIR=/tmp
for mode in --force-graph --no-graph; do
rocprofv3 --kernel-trace -d $OUT/prof -o c4 --output-format csv -- python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 3 --warmup 2 $mode > $OUT/bench$mode.json 2>$OUT/err.txt
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/c4_last_step$mode.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
start, end = idx[-3] + 1, idx[-1] + 1
t0 = int(rows[start]['Start_Timestamp'])
busy = 0
last_end = t0
gaps = 0
for r in rows[start:end]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    print('%9.1f %8.1f us  grid %-8s q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Queue_Id', '?'), name))
    if s > last_end: gaps += s - last_end
    last_end = max(last_end, e)
print('# step span %.1f us, idle (no kernel on any queue) %.1f us' % ((last_end - t0) / 1e3, gaps / 1e3))
PY
rm -rf $OUT/prof
tail -1 $OUT/c4_last_step$mode.txt
done
