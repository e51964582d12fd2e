#!/bin/bash
# session q: separable SSIM kernel; c3 at the reference's batch of 50; kernel stats of c3 / c5 of this build
OUT=gpurun_out/r03_q; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "ssim or dcn or compression or codec or conv1 or constrained" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_k.log
for b in 16 50; do echo "== c3 batch $b"; python bench.py --workload c3 --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | head -c 400; echo; done | tee $OUT/c3_batch.txt
echo "== c5"; python bench.py --workload c5 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | head -c 400 | tee $OUT/c5.txt; echo
for W in c3 c5; do
(cd /tmp && NIMG_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$W -o $W -- python $ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/prof_$W.log 2>&1)
find $OUT/prof_$W -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$W.csv; rm -rf $OUT/prof_$W
echo "== $W kernel stats"; head -22 $OUT/kernel_stats_$W.csv | cut -c1-150
done
