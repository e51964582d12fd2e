#!/bin/bash
# session s: is the c3 step time stable from process to process? (session r: 7.7 ms, then 52 / 37 / 52 ms)
OUT=gpurun_out/r03_s; mkdir -p $OUT
for i in 1 2 3 4; do
  echo "== c3 run $i"; python bench.py --workload c3 --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode 2>$OUT/c3_$i.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
print(d['value'], d['ms_per_step'], 'blocks', c.get('block_ms_per_step'), 'host_cpu', c.get('host_cpu_ms_per_step'), 'graph', c.get('hip_graph'))"
  tail -3 $OUT/c3_$i.err
done | tee $OUT/c3_runs.txt
