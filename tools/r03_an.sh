#!/bin/bash
# session an: codec epilogue fusions - tests, then configs 3 / 5 (NIMG_NO_D2S_OUT=1: depth_to_space / space_to_depth as separate passes)
OUT=gpurun_out/r03_an; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "depth_to_space or residual or dcn or DCN or codec or compression or full_channel or d2s" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  NIMG_NO_D2S_OUT=1 one c3 separate
  one c3 epilogue
  NIMG_NO_D2S_OUT=1 one c5 separate
  one c5 epilogue
done
tail -3 $OUT/err.txt
