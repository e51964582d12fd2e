#!/bin/bash
# session bg: windowed soft-codebook kernels - latent / codec tests, kernel times, configs 3 / 5 (NIMG_LATENT_NO_WINDOW=1 = before)
OUT=gpurun_out/r03_bg; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "latent or codebook or entropy or dcn or DCN or codec or compression or full_channel" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do NIMG_LATENT_NO_WINDOW=1 one c3 full; one c3 window; done
NIMG_LATENT_NO_WINDOW=1 one c5 full; one c5 window
tail -3 $OUT/err.txt
