#!/bin/bash
# session r: fast soft-codebook kernels, space-to-depth form of the codec's strided layers: parity tests, c3 / c5 with and without
OUT=gpurun_out/r03_r; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "stride2 or strided or latent or dcn or compression or twitter or codec or workflow" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_k.log
for v in "NIMG_X=1" "NIMG_LATENT_GENERIC=1" "NIMG_NO_S2D_CONV=1" "NIMG_X=1"; do
  echo "== c3 $v"; env $v python bench.py --workload c3 --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | head -c 330; echo
done | tee $OUT/c3_ab.txt
for v in "NIMG_X=1" "NIMG_NO_S2D_CONV=1"; do
  echo "== c5 $v"; env $v python bench.py --workload c5 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | head -c 330; echo
done | tee $OUT/c5_ab.txt
for W in c3 c5; do
(cd /tmp && NIMG_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$W -o $W -- python $ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/prof_$W.log 2>&1)
find $OUT/prof_$W -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$W.csv; rm -rf $OUT/prof_$W
echo "== $W kernel stats"; head -24 $OUT/kernel_stats_$W.csv | cut -c1-150
done
