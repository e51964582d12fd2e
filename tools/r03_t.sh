#!/bin/bash
# session t: residual adds fused into the 3x3 kernels of the codec; c3 / c5 / c4 of this build, full GPU suite
OUT=gpurun_out/r03_t; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for W in c3 c5; do echo "== $W"; python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['value'], d['ms_per_step'], 'blocks', c.get('block_ms_per_step'), 'graph', c.get('hip_graph'))"; done | tee $OUT/c3c5.txt
echo "== c4"; python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-side-workloads 2>/dev/null | head -c 200 | tee $OUT/c4.txt; echo
