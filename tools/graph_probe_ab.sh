# does the launch-path probe (graph capture + replays) before the timed loop change the eager step? default vs --no-graph, alternating
mkdir -p gpurun_out/$1
for i in 1 2 3; do
  for v in "" "--no-graph"; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-parity-mode --no-dp1-nccl $v > gpurun_out/$1/run.json 2> gpurun_out/$1/run.err
    python - "$v" gpurun_out/$1/run.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split('\n')[-1])
c = d['config']
print('[%-10s] %.0f patches/s  %.3f ms  blocks %s  probe %s  path %s' % (sys.argv[1], d['value'], d['ms_per_step'], c['block_ms_per_step'], c.get('launch_mode_warmup_ms_per_step'), d.get('launch_path')))
PY
  done
done
