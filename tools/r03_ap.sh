#!/bin/bash
# session ap: the full GPU suite + smoke on the current build, counter passes (FETCH_SIZE, WRITE_SIZE - separate) and kernel-time
# tables of configs 3 and 5, the default bench line
OUT=$PWD/gpurun_out/r03_ap; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
for wl in c3 c5; do
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$C -o p -- \
       python $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode > $OUT/pmc_${wl}_$C.log 2>&1)
    f=$(find $OUT/pmc_${wl}_$C -name "*counter_collection.csv" | head -1); cp $f $OUT/${wl}_$C.csv; rm -rf $OUT/pmc_${wl}_$C
  done
  python tools/pmc_step_total.py $OUT/${wl}_FETCH_SIZE.csv $OUT/${wl}_WRITE_SIZE.csv $([ $wl = c3 ] && echo 50 || echo 16) $wl > $OUT/pmc_${wl}_step_total.json 2>$OUT/pmc_${wl}_err.txt
  head -12 $OUT/pmc_${wl}_step_total.json; cat $OUT/pmc_${wl}_err.txt
  rm -f $OUT/${wl}_FETCH_SIZE.csv $OUT/${wl}_WRITE_SIZE.csv
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_$wl -o $wl --output-format csv -- python $ROOT/bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 10 --warmup 4 --no-graph > $OUT/bench_prof_$wl.json 2>$OUT/err_$wl.txt)
  cp $(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1) $OUT/${wl}_kernel_stats.csv; rm -rf $OUT/prof_$wl
done
timeout 900 python bench.py > $OUT/bench_c4.json 2>$OUT/bench_c4.err; tail -c 2500 $OUT/bench_c4.json
