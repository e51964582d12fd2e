#!/bin/bash
# round 3, first GPU session: does the C4 channel learn on scene_images (recipe search, bf16 only), per-dispatch trace of one
# C4 step, kernel stats of configs 3 and 5
OUT=gpurun_out/r03_a; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 400 python tools/train_parity.py --modes bf16 --pretrain 1500 --pretrain-lr 1e-3 --steps 3000 --lr 1e-4 --log-every 250 --out $OUT/tp_a.json > $OUT/tp_a.log 2>&1; echo "tp_a rc=$?"
timeout 300 python tools/train_parity.py --modes bf16 --pretrain 1500 --pretrain-lr 1e-3 --steps 2000 --lr 5e-4 --log-every 250 --out $OUT/tp_b.json > $OUT/tp_b.log 2>&1; echo "tp_b rc=$?"
(cd /tmp && NIMG_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace -o c4 -- python $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/trace.log 2>&1)
find $OUT/trace -name '*kernel_trace.csv' | head -1 | xargs -I{} cp {} $OUT/c4_kernel_trace.csv; rm -rf $OUT/trace
for W in c3 c5; do
(cd /tmp && NIMG_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$W -o $W -- python $ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/prof_$W.log 2>&1)
find $OUT/prof_$W -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${W}_kernel_stats.csv
find $OUT/prof_$W -name '*kernel_trace.csv' | head -1 | xargs -I{} cp {} $OUT/${W}_kernel_trace.csv; rm -rf $OUT/prof_$W
done
tail -8 $OUT/tp_a.log; tail -5 $OUT/tp_b.log
