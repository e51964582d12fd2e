#!/bin/bash
# same-box A/B of environment switches on one bench workload: tools/wl_ab.sh <workload> <reps> "A=1" "B=2" ...
W=$1; R=$2; shift 2
for i in $(seq 1 $R); do
  for v in "$@"; do
    env $v python bench.py --workload $W --steps 60 --warmup 10 --no-cpu-baseline --no-parity-mode 2>/dev/null > /tmp/wl.json
    python - "$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/wl.json').read().strip().split('\n')[-1])
print('[%-36s] %.0f patches/s  %.3f ms  probe %s  path %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('launch_mode_warmup_ms_per_step'), d.get('launch_path')))
PY
  done
done
