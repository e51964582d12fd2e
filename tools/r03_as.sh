#!/bin/bash
# session as: parameter gradients over 1 / 2 / 3 side streams (NIMG_SIDE_STREAMS) - model / workflow / graph tests, then step A/B
OUT=gpurun_out/r03_as; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 30 --warmup 8 $3 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', 'streams $2', d['value'], d['ms_per_step'], d['config'].get('launch_mode_warmup_ms_per_step'))"
}
for rep in 1 2; do
  for n in 1 2 3; do NIMG_SIDE_STREAMS=$n one c4 $n; done
done
for n in 1 2 3; do NIMG_SIDE_STREAMS=$n one c3 $n; NIMG_SIDE_STREAMS=$n one c5 $n; done
tail -3 $OUT/err.txt
