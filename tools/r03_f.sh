#!/bin/bash
# round 3, session f: 32x32x16 split-accumulator variant of the all-taps weight gradient (tests, timing, SQ counters); the
# tests touched since session e
OUT=gpurun_out/r03_f; mkdir -p $OUT
NIMG_WGRAD5_M32=1 timeout 600 python -m pytest tests -m gpu -q -x -k "wgrad5 or unpool_folded" > $OUT/pytest_m32.log 2>&1; echo "pytest m32 rc=$?"; tail -5 $OUT/pytest_m32.log
for v in "A=1" "NIMG_WGRAD5_M32=1" "NIMG_NO_WGRAD5_ALLTAPS=1" "NIMG_WGRAD5_M32=1" "A=1"; do
  echo "== $v" | tee -a $OUT/wgrad5_time.txt; env $v timeout 120 python tools/wgrad5_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wgrad5_time.txt
done
PMC_EXTRA="m32:NIMG_WGRAD5_M32=1" bash tools/pmc_wgrad5.sh > $OUT/pmc_wgrad5.txt 2>&1; grep -E "^m32|^new" $OUT/pmc_wgrad5.txt
cp gpurun_out/pmc_wgrad5/summary.json $OUT/pmc_wgrad5_summary.json
timeout 900 python -m pytest tests -m gpu -q -k "trainable or bit_neutral or torch_library or captured or add_n or fan_forward or front" > $OUT/pytest_k.log 2>&1; echo "pytest k rc=$?"; tail -8 $OUT/pytest_k.log
