#!/bin/bash
# graph replay vs eager launches of the C4 step, alternating on one box
OUT=gpurun_out/r03_ae; mkdir -p $OUT
for i in 1 2 3; do
  for v in "" "--no-graph"; do
    echo "== step graph=[$v]"; python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-side-workloads $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['value'], d['ms_per_step'], 'blocks', c.get('block_ms_per_step'), 'graph', c.get('hip_graph'), 'host_cpu', c.get('host_cpu_ms_per_step'))"
  done
done | tee $OUT/graph_vs_eager.txt
