#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
python tools/c5_grad_diag.py 2>&1 | grep -v amdgpu.ids | tee $OUT/diag.txt
