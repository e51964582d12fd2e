#!/bin/bash
# session bd: e1 and the last block sum stored as bf16 space-to-depth images, e2 / elat forward + weight gradient on the 3x3 kernels - tests, configs 3 / 5
OUT=gpurun_out/r03_bd; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "space_to_depth or depth_to_space or residual or dcn or DCN or codec or compression or full_channel or d2s or stride2 or strided" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do NIMG_NO_S2D_CHAIN=1 one c3 before; one c3 chain; NIMG_NO_S2D_CHAIN=1 one c5 before; one c5 chain; done
tail -3 $OUT/err.txt
