#!/bin/bash
OUT=gpurun_out/r03_w; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -k "smooth_variant or bit_neutral or full_channel" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_k.log
