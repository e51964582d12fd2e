#!/bin/bash
OUT=gpurun_out/r03_af; mkdir -p $OUT
for i in 1 2; do
  for v in "" "--no-graph"; do
    echo "== c5 graph=[$v]"; python bench.py --workload c5 --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['value'], d['ms_per_step'], 'graph', c.get('hip_graph'), 'host_cpu', c.get('host_cpu_ms_per_step'))"
  done
done | tee $OUT/c5_graph_vs_eager.txt
