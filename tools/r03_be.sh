#!/bin/bash
# session be: repeatability of config 5 with the space-to-depth chain (two late runs of session bd were 20 - 30 % slow)
OUT=gpurun_out/r03_be; mkdir -p $OUT
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; print('$1', '$2', d['value'], d['ms_per_step'], c.get('hip_graph'), c.get('launch_mode_warmup_ms_per_step'))"
}
for rep in 1 2 3 4; do one c5 chain; done
NIMG_NO_S2D_CHAIN=1 one c5 before
one c5 chain
