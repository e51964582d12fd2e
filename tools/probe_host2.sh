mkdir -p gpurun_out/r04_zb
grep -E "nr_throttled|nr_periods" /sys/fs/cgroup/cpu.stat
python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-parity-mode --no-dp1-nccl --no-graph > gpurun_out/r04_zb/run.json 2> gpurun_out/r04_zb/run.err &
BP=$!
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  sleep 2
  echo "--- t=$((2*i))s  $(grep -E 'nr_throttled' /sys/fs/cgroup/cpu.stat)  load $(cut -d' ' -f1 /proc/loadavg)"
  top -H -b -n 1 -w 200 2>/dev/null | sed -n 7,16p | cut -c1-150
done
wait $BP
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_zb/run.json').read().strip().split('\n')[-1])
print('%.0f patches/s  %.3f ms  ring %.3f host_cpu %.2f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['config'].get('host_cpu_ms_per_step', 0)))
PY
