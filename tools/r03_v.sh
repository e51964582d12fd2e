#!/bin/bash
# session v: bf16 storage of the tensors inside the codec's residual blocks (bit-neutral): tests, c3 / c5
OUT=gpurun_out/r03_v; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "dcn or compression or twitter or codec" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_k.log
for W in c3 c5; do echo "== $W"; python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['value'], d['ms_per_step'], 'blocks', c.get('block_ms_per_step'), 'graph', c.get('hip_graph'))"; done | tee $OUT/c3c5.txt
