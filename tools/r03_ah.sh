#!/bin/bash
# session ah: bf16 copies of the DCN residual stream - tests, then A/B of configs 3 and 5 (NIMG_NO_BF16_COPY=1 = before)
OUT=gpurun_out/r03_ah; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "residual or dcn or DCN or codec or compression or full_channel or config5 or c5 or c3" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  NIMG_NO_BF16_COPY=1 one c3 f32stream
  one c3 bf16copy
  NIMG_NO_BF16_COPY=1 one c5 f32stream
  one c5 bf16copy
done
tail -3 $OUT/err.txt
