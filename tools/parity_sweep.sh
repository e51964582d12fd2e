#!/bin/bash
# bf16-only sweep of the joint phase's learning rate / recipe / batch order: how often does a class die (tools/parity_rootcause.py)?
mkdir -p gpurun_out/r06
for lr in 1e-4 5e-5 3e-5; do
  timeout 900 python tools/parity_rootcause.py --recipes base,warm:300 --seeds 12,13,14,15 --modes bf16 --steps 1000 --lr $lr \
      --log-every 100 --out gpurun_out/r06/parity_sweep_lr$lr.json > gpurun_out/r06/parity_sweep_lr$lr.log 2>&1
done
