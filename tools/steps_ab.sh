# step rate against the number of timed steps and the run-ahead bound (diagnostic), one box
mkdir -p gpurun_out/$1
for v in "--steps 200" "--steps 1500" "--steps 1500 --run-ahead 0" "--steps 200 --run-ahead 0" "--steps 1500 --run-ahead 8" "--steps 200"; do
    python bench.py --warmup 10 --no-cpu-baseline --no-parity-mode --no-dp1-nccl --no-side-workloads $v > gpurun_out/$1/run.json 2> gpurun_out/$1/run.err
    python - "$v" gpurun_out/$1/run.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split('\n')[-1])
c = d['config']
print('[%-28s] %.0f patches/s  %.3f ms  blocks %s  ring %.3f  path %s' % (sys.argv[1], d['value'], d['ms_per_step'], c['block_ms_per_step'], d['roofline']['ms_per_launch'], d.get('launch_path')))
PY
done
