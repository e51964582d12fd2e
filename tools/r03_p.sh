#!/bin/bash
# session p: packed-FMA constrained-conv stencil + its weight gradient, lane-local pooling in conv1_pool_fwd: parity tests, then
# same-box A/B (prev = HEAD before, new, b = conv1 with 2 resident workgroups per CU)
OUT=gpurun_out/r03_p; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "cconv or constrained or conv1 or front_end or fan or forensics or workflow" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_k.log
for v in "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_X=1" "NIMG_LIBPATH=neural-imaging_amd/libnimg_b.so" "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_X=1"; do
  echo "== front_time $v"; env $v python tools/front_time.py 20 2>&1 | grep -v "amdgpu.ids"
done | tee $OUT/front_ab.txt
for i in 1 2; do
  for v in "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_X=1" "NIMG_LIBPATH=neural-imaging_amd/libnimg_b.so"; do
    echo "== step $v"; env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-side-workloads 2>/dev/null | head -c 150; echo
  done
done | tee $OUT/step_ab.txt
