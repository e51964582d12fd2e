#!/bin/bash
# timing sweep of the front-end kernel variants (gpurun ... 'bash tools/front_sweep.sh')
mkdir -p gpurun_out/front
for v in 0 1; do echo "== cconv variant $v"; NIMG_CCONV_VARIANT=$v python tools/front_time.py 20 2>&1 | grep -i "cconv3 fwd\|cconv3 dgrad main"; done | tee gpurun_out/front/sweep.txt
python tools/front_time.py 30 2>&1 | tee -a gpurun_out/front/sweep.txt
