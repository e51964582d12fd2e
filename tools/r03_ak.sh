#!/bin/bash
# session ak: compile-time exponent in the soft-codebook kernels, wave-parallel finalize kernels - tests, configs 3 / 5
OUT=gpurun_out/r03_ak; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "latent or codebook or entropy or dcn or DCN or codec or compression or full_channel or loss" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  one c3 now
  one c5 now
done
tail -3 $OUT/err.txt
