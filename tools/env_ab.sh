#!/bin/bash
# same-box A/B of environment switches on the bench step (diagnostic): tools/env_ab.sh "A=1" "B=2" ...
for i in 1 2; do
  for v in "$@"; do
    echo "== $v"; env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode 2>/dev/null | head -c 150; echo
  done
done
