#!/bin/bash
# diagnostic builds of the sparse weight gradient: no global fetch after the first tile / no tap pipeline (results wrong, timings not)
OUT=gpurun_out/r03_ab; mkdir -p $OUT
for v in "NIMG_X=1" "NIMG_LIBPATH=neural-imaging_amd/libnimg_b.so" "NIMG_LIBPATH=neural-imaging_amd/libnimg_c.so"; do
  echo "== $v"; env $v python tools/wgrad5_time.py 20 2>&1 | grep -v amdgpu.ids
done | tee $OUT/diag.txt
