#!/bin/bash
# session at: the manipulations of a step on the side streams (NIMG_MANIP_STREAMS=1) - workflow tests with the switch on, step A/B
OUT=gpurun_out/r03_at; mkdir -p $OUT
NIMG_MANIP_STREAMS=1 timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "workflow or manipulation or channel or graph or captured or step" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 30 --warmup 8 $3 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'], d['config'].get('launch_mode_warmup_ms_per_step'))"
}
for rep in 1 2 3; do
  one c4 serial
  NIMG_MANIP_STREAMS=1 one c4 beside
done
one c5 serial; NIMG_MANIP_STREAMS=1 one c5 beside
tail -3 $OUT/err.txt
