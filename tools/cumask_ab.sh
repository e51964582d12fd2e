#!/bin/bash
# CU-partitioned side streams (NIMG_SIDE_CUS=n: the parameter-gradient streams run on n CUs, the launch stream keeps all 256):
# C4 step, eager launches (a captured graph replays on the launch stream and loses the masks), alternating on one box.
cd "$(dirname "$0")/.."
run() {
  r=$(env "$@" python bench.py --steps 60 --warmup 15 --no-graph --no-side-workloads --no-cpu-baseline --no-parity-mode --no-dp1-nccl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.0f patches/s  path=%s' % (d['ms_per_step'], d['value'], d.get('launch_path')))")
  echo "$*  $r"
}
for rnd in 1 2; do
  run BASE=1
  for n in 224 192 160 128; do
    run NIMG_SIDE_CUS=$n NIMG_WGRAD5_ALLTAPS_BLOCKS=$n
    run NIMG_SIDE_CUS=$n
  done
done
