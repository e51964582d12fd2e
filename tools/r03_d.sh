#!/bin/bash
# round 3, session d: all-taps weight gradient after the fragment swap + scheduling control (tests, A/B, SQ counters), NIP
# pre-training curves at smaller learning rates
OUT=gpurun_out/r03_d; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "wgrad5 or unpool_folded" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_k.log
for v in "NIMG_NO_WGRAD5_ALLTAPS=1" "A=1" "NIMG_WGRAD5_SCHED=1" "NIMG_WGRAD5_KX3L=1" "NIMG_WGRAD5_TH8=1" "NIMG_NO_WGRAD5_ALLTAPS=1" "A=1"; do
  echo "== $v" | tee -a $OUT/wgrad5_time.txt; env $v timeout 120 python tools/wgrad5_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wgrad5_time.txt
done
bash tools/pmc_wgrad5.sh > $OUT/pmc_wgrad5.txt 2>&1; tail -12 $OUT/pmc_wgrad5.txt
cp gpurun_out/pmc_wgrad5/summary.json $OUT/pmc_wgrad5_summary.json
timeout 300 python tools/nip_diag.py 1500 3e-4 f32,bf16 > $OUT/nip_diag_3e-4.log 2>&1; grep -v amdgpu.ids $OUT/nip_diag_3e-4.log
timeout 300 python tools/nip_diag.py 1500 1e-4 f32,bf16 > $OUT/nip_diag_1e-4.log 2>&1; grep -v amdgpu.ids $OUT/nip_diag_1e-4.log
