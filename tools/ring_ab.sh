#!/bin/bash
# A/B of the ring kernel's forms on one box (diagnostic): previous build (libnimg_prev.so) vs current, launch variants
export FIT_QUICK=1
for v in "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_RING_WGS=-1" "NIMG_RING_WGS=0" "NIMG_LIBPATH=neural-imaging_amd/libnimg_prev.so" "NIMG_RING_WGS=-1"; do
  echo "== $v"; env $v python tools/conv5_fit.py 2>&1 | grep "TFLOP"
done
