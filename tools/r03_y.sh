#!/bin/bash
OUT=gpurun_out/r03_y; mkdir -p $OUT
./tools/probe/smfmac_probe 2>&1 | tee $OUT/smfmac_probe.txt
