#!/bin/bash
# session ay: workgroups per all-taps 3x3 weight gradient (NIMG_WGRAD3_BLOCKS: 256 = one per CU, fewer = fewer slabs to reduce;
# with three side streams several weight gradients run beside each other anyway)
OUT=gpurun_out/r03_ay; mkdir -p $OUT
one() {
  python bench.py --workload $1 --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', 'blocks $2', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  for nb in 256 128 64; do NIMG_WGRAD3_BLOCKS=$nb one c3 $nb; done
done
for nb in 256 128 64; do NIMG_WGRAD3_BLOCKS=$nb one c5 $nb; NIMG_WGRAD3_BLOCKS=$nb one c4 $nb; done
