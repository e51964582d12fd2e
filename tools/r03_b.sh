#!/bin/bash
# round 3, second GPU session: both compute modes trained from one initialisation on the same batches (does float32 spike too?),
# a smaller joint learning rate, another batch order
OUT=gpurun_out/r03_b; mkdir -p $OUT
timeout 600 python tools/train_parity.py --modes bf16,f32 --pretrain 1500 --pretrain-lr 1e-3 --steps 2000 --lr 1e-4 --log-every 125 --out $OUT/tp_both.json > $OUT/tp_both.log 2>&1; echo "tp_both rc=$?"
timeout 300 python tools/train_parity.py --modes bf16 --pretrain 1500 --pretrain-lr 1e-3 --steps 3000 --lr 5e-5 --log-every 250 --out $OUT/tp_lr5e5.json > $OUT/tp_lr5e5.log 2>&1; echo "tp_lr rc=$?"
timeout 300 python tools/train_parity.py --modes bf16 --pretrain 1500 --pretrain-lr 1e-3 --steps 3000 --lr 1e-4 --seed 77 --log-every 250 --out $OUT/tp_seed77.json > $OUT/tp_seed77.log 2>&1; echo "tp_seed rc=$?"
grep -h "parity\|agreement\|delta" -A3 $OUT/tp_both.log | tail -40
