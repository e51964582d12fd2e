#!/bin/bash
# session am: 64-column form of the slab reduction - weight-gradient tests, then configs 3 / 5 / 4
OUT=gpurun_out/r03_am; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "wgrad or weight_grad or dcn or DCN or codec or fan or FAN or unet or backward" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do one c3; one c5; one c4; done
tail -3 $OUT/err.txt
