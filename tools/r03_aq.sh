#!/bin/bash
# session aq: wide gaussian kernels - tests, bit-identity against the 16 x 16 form (tools/gauss_check.py), kernel times, step A/B
OUT=gpurun_out/r03_aq; mkdir -p $OUT
timeout 600 python -m pytest tests -x -q -m gpu -k "gaussian or manipulation" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
python tools/gauss_check.py $OUT/wide.pt > /dev/null 2>&1
NIMG_GAUSS_NARROW=1 python tools/gauss_check.py $OUT/narrow.pt > /dev/null 2>&1
python - <<'PY'
import torch
a, b = torch.load('gpurun_out/r03_aq/wide.pt'), torch.load('gpurun_out/r03_aq/narrow.pt')
print('bit-identical to the 16x16 form:', all(torch.equal(x, y) for x, y in zip(a, b)), len(a))
PY
rm -f $OUT/wide.pt $OUT/narrow.pt
one() {
  python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 30 --warmup 8 2>>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('c4', '$1', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do NIMG_GAUSS_NARROW=1 one narrow; one wide; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $PWD/$OUT/prof -o c4 --output-format csv -- python bench.py --no-cpu-baseline --no-parity-mode --no-side-workloads --steps 5 --warmup 2 --no-graph > /dev/null 2>>$OUT/err.txt
python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gaussian' in r['Name'] or 'sparse_axis' in r['Name'] or 'sharpen' in r['Name']: print(r['Name'][23:60], r['Calls'], float(r['AverageNs'])/1e3)
" $(find $OUT/prof -name "*kernel_stats.csv")
rm -rf $OUT/prof
