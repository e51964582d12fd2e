#!/bin/bash
# same-box A/B of two library builds on the bench step (diagnostic): libnimg_prev.so vs libnimg.so, alternating
for i in 1 2; do
  for v in "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_X=1"; do
    echo "== $v"; env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode 2>/dev/null | head -c 150; echo
  done
done
