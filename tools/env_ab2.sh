#!/bin/bash
# same-box A/B of environment switches on the C4 step, full line kept: tools/env_ab2.sh <reps> "A=1" "B=2 C=3" ...
R=$1; shift
for i in $(seq 1 $R); do
  for v in "$@"; do
    env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-dp1-nccl --no-side-workloads 2>/dev/null > /tmp/ab.json
    python - "$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().split('\n')[-1])
print('[%-44s] %.0f patches/s  %.3f ms  probe %s  path %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('launch_mode_warmup_ms_per_step'), d.get('launch_path')))
PY
  done
done
