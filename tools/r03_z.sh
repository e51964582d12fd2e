#!/bin/bash
# session z: structured-sparsity (v_smfmac) form of the 5x5 weight gradient: parity test, then stand-alone A/B
OUT=gpurun_out/r03_z; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "wgrad5" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_k.log
for v in "NIMG_NO_WGRAD5_SPARSE=1" "NIMG_X=1" "NIMG_NO_WGRAD5_SPARSE=1" "NIMG_X=1"; do
  echo "== $v"; env $v python tools/wgrad5_time.py 20 2>&1 | grep -v amdgpu.ids
done | tee $OUT/wgrad5_ab.txt
