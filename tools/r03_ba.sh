#!/bin/bash
# session ba: bf16 gradient out of the FAN head - FAN / workflow tests, step A/B against float32 (NIMG_STORE_F32-free: old build n/a)
OUT=gpurun_out/r03_ba; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "fan or FAN or head or workflow or channel or graph" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config'].get('launch_mode_warmup_ms_per_step'))"
}
for rep in 1 2 3; do one c4; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $PWD/$OUT/prof -o c4 --output-format csv -- python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 5 --warmup 2 --no-graph > /dev/null 2>>$OUT/err.txt
python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if '<1, 1,' in r['Name'] or 'gap_bwd' in r['Name']: print(r['Name'][23:110], r['Calls'], float(r['AverageNs'])/1e3)
" $(find $OUT/prof -name "*kernel_stats.csv")
rm -rf $OUT/prof
