#!/bin/bash
# round 3, session h: 3x3 all-taps weight gradient (tests + A/B timing), DCN harness with / without the fast power
OUT=gpurun_out/r03_h; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "wgrad3" > $OUT/pytest_wgrad3.log 2>&1; echo "pytest wgrad3 rc=$?"; tail -5 $OUT/pytest_wgrad3.log
timeout 300 python tools/wgrad3_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/wgrad3_time.txt
timeout 600 python -m pytest tests -m gpu -q -k "dcn_pretraining_harness or trainable_jpeg" > $OUT/pytest_dcn.log 2>&1; echo "pytest dcn rc=$?"; tail -4 $OUT/pytest_dcn.log
NIMG_LATENT_GENERIC_POW=1 timeout 600 python -m pytest tests -m gpu -q -k "dcn_pretraining_harness" > $OUT/pytest_dcn_generic.log 2>&1; echo "pytest dcn generic pow rc=$?"; tail -4 $OUT/pytest_dcn_generic.log; grep -h "AssertionError: \[" $OUT/pytest_dcn*.log | cut -c1-200
