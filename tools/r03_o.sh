#!/bin/bash
# session o: new dJPEG kernels (templated rounding, reciprocal division, tables in LDS) + generic manipulations: parity tests, then
# same-box A/B of the builds (prev = before, new = 4/3 workgroups per CU, b = 5/4) and of the ring kernel's operand pipelining
OUT=gpurun_out/r03_o; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "jpeg or manipulations or codec or dcn" > $OUT/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_k.log
for v in "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_X=1" "NIMG_LIBPATH=neural-imaging_amd/libnimg_b.so" "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_X=1"; do
  echo "== front_time $v"; env $v python tools/front_time.py 20 2>&1 | grep -i "djpeg"
done | tee $OUT/djpeg_ab.txt
export FIT_QUICK=1
for v in "NIMG_X=1" "NIMG_RING_PIPE=1" "NIMG_X=1" "NIMG_RING_PIPE=1"; do
  echo "== conv5_fit $v"; env $v python tools/conv5_fit.py 2>&1 | grep "TFLOP"
done | tee $OUT/ring_pipe_ab.txt
for i in 1 2; do
  for v in "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_X=1" "NIMG_RING_PIPE=1"; do
    echo "== step $v"; env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-side-workloads 2>/dev/null | head -c 150; echo
  done
done | tee $OUT/step_ab.txt
