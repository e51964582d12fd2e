#!/bin/bash
# session aa: sparse (v_smfmac) 5x5 weight gradient in the step: full GPU suite, then step A/B on one box
OUT=gpurun_out/r03_aa; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
for i in 1 2; do
  for v in "NIMG_NO_WGRAD5_SPARSE=1" "NIMG_X=1"; do
    echo "== step $v"; env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-side-workloads 2>/dev/null | head -c 150; echo
  done
done | tee $OUT/step_ab.txt
