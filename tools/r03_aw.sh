#!/bin/bash
# session aw / bb / bc / bf / bh: final validation of round 3 - full GPU suite, smoke, bit-identity of the 16-byte resampling form, kernel statistics of
# the C4 step (one stream) and the default bench line
OUT=$PWD/gpurun_out/r03_bh; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python tools/resample_check.py $OUT/vec.pt 2>&1 | tail -1
NIMG_SPARSE_AXIS_SCALAR=1 python tools/resample_check.py $OUT/scalar.pt 2>&1 | tail -1
python - <<'PY'
import torch
a, b = torch.load('gpurun_out/r03_bh/vec.pt'), torch.load('gpurun_out/r03_bh/scalar.pt')
print('resampling: 16-byte row form bit-identical to the per-pixel form:', all(torch.equal(x, y) for x, y in zip(a, b)), len(a))
PY
rm -f $OUT/vec.pt $OUT/scalar.pt
(cd /tmp && NIMG_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c4 -- \
   python $ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-parity-mode --no-side-workloads > $OUT/prof.log 2>&1)
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bf16_b64.csv; rm -rf $OUT/prof
head -12 $OUT/kernel_stats_bf16_b64.csv | cut -c1-150
timeout 900 python bench.py > $OUT/bench_c4.json 2>$OUT/bench_c4.err; head -c 600 $OUT/bench_c4.json; echo; tail -c 900 $OUT/bench_c4.json
