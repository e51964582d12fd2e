#!/bin/bash
# round 3, session j: same-box A/B on the C4 step: two chunks of loads in flight in the 3x3 kernels (libnimg_pf2.so),
# the all-taps weight gradients on / off
OUT=gpurun_out/r03_j; mkdir -p $OUT
run() { env $1 timeout 200 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config']['block_ms_per_step'])"; }
for v in "A=1" "NIMG_LIBPATH=neural-imaging_amd/libnimg_pf2.so" "NIMG_NO_WGRAD3_ALLTAPS=1" "NIMG_NO_WGRAD5_ALLTAPS=1" "A=1" "NIMG_LIBPATH=neural-imaging_amd/libnimg_pf2.so" "NIMG_NO_WGRAD3_ALLTAPS=1"; do
  echo "== $v" | tee -a $OUT/step_ab.txt; run "$v" | tee -a $OUT/step_ab.txt
done
NIMG_LIBPATH=neural-imaging_amd/libnimg_pf2.so timeout 600 python -m pytest tests -m gpu -q -x -k "conv and not wgrad" > $OUT/pytest_pf2.log 2>&1; echo "pytest pf2 rc=$?"; tail -3 $OUT/pytest_pf2.log
