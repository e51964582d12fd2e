#!/bin/bash
# session ai: kernel-time tables of configs 3 and 5 (rocprofv3 --kernel-trace --stats)
OUT=$PWD/gpurun_out/r03_ai; mkdir -p $OUT
export TMPDIR=/tmp
for wl in c3 c5; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$wl -o $wl --output-format csv -- python bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 10 --warmup 4 --no-graph > $OUT/bench_$wl.json 2>$OUT/err_$wl.txt
  f=$(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/${wl}_kernel_stats.csv
  head -30 $f | cut -c1-200
  rm -rf $OUT/prof_$wl
done
