#!/bin/bash
# session ar: 16-byte forms of the separable resampling kernels - tests, bit-identity against the per-pixel form, kernel times, step
OUT=gpurun_out/r03_ar; mkdir -p $OUT
timeout 600 python -m pytest tests -x -q -m gpu -k "manipulation or resample or augment" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
python tools/resample_check.py $OUT/vec.pt 2>&1 | tail -2
NIMG_SPARSE_AXIS_SCALAR=1 python tools/resample_check.py $OUT/scalar.pt 2>&1 | tail -2
python - <<'PY'
import torch
a, b = torch.load('gpurun_out/r03_ar/vec.pt'), torch.load('gpurun_out/r03_ar/scalar.pt')
print('bit-identical to the per-pixel form:', all(torch.equal(x, y) for x, y in zip(a, b)), len(a))
PY
rm -f $OUT/vec.pt $OUT/scalar.pt
one() {
  python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('c4', '$1', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do NIMG_SPARSE_AXIS_SCALAR=1 one scalar; one vector; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $PWD/$OUT/prof -o c4 --output-format csv -- python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 5 --warmup 2 --no-graph > /dev/null 2>>$OUT/err.txt
python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gaussian' in r['Name'] or 'sparse_axis' in r['Name'] or 'sharpen' in r['Name']: print(r['Name'][23:60], r['Calls'], float(r['AverageNs'])/1e3)
" $(find $OUT/prof -name "*kernel_stats.csv")
rm -rf $OUT/prof
