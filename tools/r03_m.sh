#!/bin/bash
OUT=gpurun_out/r03_m; mkdir -p $OUT
python tools/unet_fwd_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/unet_fwd_time.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "conv or unet or workflow_bf16 or model_families" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_k.log
