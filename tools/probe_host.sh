# consecutive default-size bench runs on one box with the stall probe (diagnostic): does a run stall at the start of its timed loop?
mkdir -p gpurun_out/$1
grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat; cat /sys/fs/cgroup/cpu.max
for i in 1 2 3 4 5 6 7; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-dp1-nccl --stall-probe > gpurun_out/$1/run_$i.json 2> gpurun_out/$1/run_$i.err
  python - gpurun_out/$1/run_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
c = d['config']
print('run %s: %.0f patches/s  %.3f ms  blocks %s  slowest host steps %s  path %s' % (sys.argv[1][-10:], d['value'], d['ms_per_step'], c['block_ms_per_step'], c.get('stall_probe_slowest_host_steps'), d.get('launch_path')))
PY
  grep -c "most recent call first" gpurun_out/$1/run_$i.err
done
grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat
