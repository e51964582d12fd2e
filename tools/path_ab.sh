#!/bin/bash
# graph replay vs eager launches of the C4 step, forced, alternating, block times kept (diagnostic)
for i in 1 2 3; do
  for v in "--force-graph" "--no-graph"; do
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-dp1-nccl --no-side-workloads $v 2>/dev/null > /tmp/p.json
    python - "$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/p.json').read().strip().split('\n')[-1])
print('[%-14s] %.0f patches/s  %.3f ms  blocks %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['config']['block_ms_per_step']))
PY
  done
done
