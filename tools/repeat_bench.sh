#!/bin/bash
# the default bench line N times in a row on one box, full JSON kept, rocm-smi clocks / power in between (diagnostic: does a box
# sustain the step rate over consecutive runs?)   tools/repeat_bench.sh <outdir> <n> [bench flags...]
OUT=$1; N=$2; shift 2
mkdir -p $OUT
for i in $(seq 1 $N); do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode "$@" > $OUT/run_$i.json 2> $OUT/run_$i.err
  python - $OUT/run_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('run', sys.argv[1], 'patches/s %.0f  ms %.3f  ring %.3f ms  path %s  host_cpu %s' % (
    d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d.get('launch_path'), d.get('eager_host_cpu_ms_per_step')))
PY
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | head -6
done
