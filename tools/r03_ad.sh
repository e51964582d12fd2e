#!/bin/bash
# round 3, session ad: full GPU suite, smoke, default bench line, kernel stats, one-step trace, whole-step PMC passes
OUT=gpurun_out/r03_ad; mkdir -p $OUT
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/gpu_check.sh r03_ad tests c4 prof trace pmc:FETCH_SIZE pmc:WRITE_SIZE 2>&1 | grep -v "^\"" | tail -40
F=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_step_total.py $F $W 64 > $OUT/pmc_step_total.json 2> $OUT/pmc_step_total.err; head -c 500 $OUT/pmc_step_total.json; cat $OUT/pmc_step_total.err
cp $F $OUT/pmc_step_FETCH_SIZE.csv; cp $W $OUT/pmc_step_WRITE_SIZE.csv; rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
