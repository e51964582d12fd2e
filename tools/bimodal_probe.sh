#!/bin/bash
# The C4 step comes in two per-process modes ~3.5 % apart on one box (tools/rep_c4.sh).  Kernel stats of N separate processes, to see
# which kernels differ:  tools/bimodal_probe.sh [N]  -> gpurun_out/bimodal/run<i>_stats.csv + bench ms per run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bimodal
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for i in $(seq 1 ${1:-6}); do
  rm -rf $OUT/raw
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o c4 -- python $ROOT/bench.py --steps 20 --warmup 5 --no-graph \
      --no-side-workloads --no-cpu-baseline --no-parity-mode --no-dp1-nccl > $OUT/run$i.log 2>&1
  find $OUT/raw -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/run${i}_stats.csv
  rm -rf $OUT/raw
  tail -1 $OUT/run$i.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('run $i: %.3f ms' % d['ms_per_step'])
except Exception as e: print('run $i: ?', e)"
done
