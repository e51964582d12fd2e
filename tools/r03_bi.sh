#!/bin/bash
# session bi: counter-based HBM traffic of one config-3 / config-5 step on the last build (FETCH_SIZE, WRITE_SIZE: separate passes)
OUT=$PWD/gpurun_out/r03_bi; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
for wl in c3 c5; do
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$C -o p -- \
       python $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode > $OUT/pmc_${wl}_$C.log 2>&1)
    f=$(find $OUT/pmc_${wl}_$C -name "*counter_collection.csv" | head -1); cp $f $OUT/${wl}_$C.csv; rm -rf $OUT/pmc_${wl}_$C
  done
  python tools/pmc_step_total.py $OUT/${wl}_FETCH_SIZE.csv $OUT/${wl}_WRITE_SIZE.csv $([ $wl = c3 ] && echo 50 || echo 16) $wl > $OUT/pmc_${wl}_step_total.json 2>$OUT/pmc_${wl}_err.txt
  head -9 $OUT/pmc_${wl}_step_total.json | tail -5; cat $OUT/pmc_${wl}_err.txt
  rm -f $OUT/${wl}_FETCH_SIZE.csv $OUT/${wl}_WRITE_SIZE.csv
done
