#!/bin/bash
# round 3, session g: new tests, c3 / c5 after the latent-kernel changes, default bench line with the side workloads
OUT=gpurun_out/r03_g; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -k "trainable or augmentation or jpeg_process or latent or dcn or learned_codec or torch_library or bit_neutral" > $OUT/pytest_k.log 2>&1; echo "pytest k rc=$?"; tail -8 $OUT/pytest_k.log
bash tools/gpu_check.sh r03_g c3 c5 c4 2>&1 | tail -30
