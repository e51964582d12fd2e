#!/bin/bash
# round 3, third GPU session: new 5x5 weight-gradient kernel (tests + A/B timing), NIP pre-training curves of the modes
OUT=gpurun_out/r03_c; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "wgrad5 or unpool_folded or captured_step or dominant_conv5" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_k.log
for v in "NIMG_NO_WGRAD5_ALLTAPS=1" "A=1" "NIMG_WGRAD5_KX3L=1" "NIMG_WGRAD5_TH8=1" "NIMG_WGRAD5_ALLTAPS_BLOCKS=512" "NIMG_NO_WGRAD5_ALLTAPS=1" "A=1"; do
  echo "== $v" | tee -a $OUT/wgrad5_time.txt; env $v timeout 120 python tools/wgrad5_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wgrad5_time.txt
done
timeout 400 python tools/nip_diag.py 400 1e-3 > $OUT/nip_diag.log 2>&1; echo "nip_diag rc=$?"; cat $OUT/nip_diag.log | grep -v amdgpu.ids
