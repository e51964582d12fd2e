#!/bin/bash
# round 3, session l: 3x3 ring kernel - bit-identity tests, per-layer A/B, C4 step A/B
OUT=gpurun_out/r03_l; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "conv3_ring" > $OUT/pytest_ring3.log 2>&1; echo "pytest ring3 rc=$?"; tail -6 $OUT/pytest_ring3.log
timeout 300 python tools/unet_fwd_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/unet_fwd_time.txt
run() { env $1 timeout 200 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config']['block_ms_per_step'], d['config']['loss'])"; }
for v in "A=1" "NIMG_NO_CONV3_RING=1" "NIMG_CONV3_RING_ONLY=128" "A=1" "NIMG_NO_CONV3_RING=1"; do
  echo "== $v" | tee -a $OUT/step_ab.txt; run "$v" | tee -a $OUT/step_ab.txt
done
