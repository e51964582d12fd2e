#!/bin/bash
# round 3, session e: full GPU suite, default bench line (new parity leg), kernel stats, whole-step PMC passes
OUT=gpurun_out/r03_e; mkdir -p $OUT
bash tools/gpu_check.sh r03_e tests c4 prof pmc:FETCH_SIZE pmc:WRITE_SIZE 2>&1 | tail -60
F=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_step_total.py $F $W 64 > $OUT/pmc_step_total.json 2> $OUT/pmc_step_total.err; head -c 600 $OUT/pmc_step_total.json; cat $OUT/pmc_step_total.err
cp $F $OUT/pmc_step_FETCH_SIZE.csv; cp $W $OUT/pmc_step_WRITE_SIZE.csv; rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
