#!/bin/bash
# quick C4 step timing (no side legs): tools/quick_c4.sh [ENV=VALUE ...]   -> one line per variant, alternating twice
cd "$(dirname "$0")/.."
for rnd in 1 2; do
  for v in "$@" "BASE=1"; do
    r=$(env $v python bench.py --steps 60 --warmup 15 --no-side-workloads --no-cpu-baseline --no-parity-mode --no-dp1-nccl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.0f patches/s  path=%s  ring %.4f ms' % (d['ms_per_step'], d['value'], d.get('launch_path'), d['roofline'].get('ms_per_launch', 0) if isinstance(d.get('roofline'),dict) else 0))")
    echo "$v  $r"
  done
done
