#!/bin/bash
OUT=gpurun_out/r03_ac; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "dgrad5 or unpool_folded" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_k.log
python tools/dgrad5_time.py 20 2>&1 | grep -v amdgpu.ids | tee $OUT/dgrad5_time.txt
