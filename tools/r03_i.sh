#!/bin/bash
# round 3, session i: re-run of the adjusted tests; A/B of the 32-channel-tile threshold of the 3x3 / 1x1 layers on the C4 step
OUT=gpurun_out/r03_i; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -k "dcn_pretraining_harness or trainable_jpeg" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_k.log
for v in 384 1024 2048 8192 384 1024; do
  echo "== NIMG_TN32_BELOW=$v" | tee -a $OUT/tn32_ab.txt
  NIMG_TN32_BELOW=$v timeout 200 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config']['block_ms_per_step'])" | tee -a $OUT/tn32_ab.txt
done
