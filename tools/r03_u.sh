#!/bin/bash
# session u: the codec's dominant 3x3 layer at several batch sizes (tail quantisation?), tile-width threshold, bf16 storage
OUT=gpurun_out/r03_u; mkdir -p $OUT
for v in "NIMG_X=1" "NIMG_TN32_BELOW=100000" "DCN_STORE=bf16" "DCN_STORE=bf16 NIMG_NO_WGRAD3_ALLTAPS=1"; do
  echo "== $v"; env $v python tools/dcn_conv_time.py 20 2>&1 | grep -v amdgpu.ids
done | tee $OUT/dcn_conv_time.txt
