#!/bin/bash
# session av: codec odds and ends (3-channel d2s2 kernel, bf16 copy out of d512, bf16 e1 gradient) - tests, configs 3 / 5
OUT=gpurun_out/r03_av; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "depth_to_space or residual or dcn or DCN or codec or compression or full_channel or d2s or stride2 or strided" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do one c3; one c5; done
tail -3 $OUT/err.txt
