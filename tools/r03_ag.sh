#!/bin/bash
OUT=gpurun_out/r03_ag; mkdir -p $OUT
python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads 2>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['value'], d['ms_per_step'], 'graph', c.get('hip_graph'), 'modes', c.get('launch_mode_warmup_ms_per_step'), 'host_cpu', c.get('host_cpu_ms_per_step'), 'eager', c.get('eager_ms_per_step'))"
tail -3 $OUT/err.txt
