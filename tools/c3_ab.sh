# config-3 step under environment switches (diagnostic), alternating: tools/c3_ab.sh "A=1" "B=1 C=2" ...
for i in 1 2; do
  for v in "$@"; do
    env $v python bench.py --workload c3 --steps 60 --warmup 10 --no-cpu-baseline --no-parity-mode 2>/dev/null > /tmp/c3.json
    python - "$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/c3.json').read().strip().split('\n')[-1])
print('[%-44s] %.0f patches/s  %.3f ms  path %s' % (sys.argv[1], d['value'], d['ms_per_step'], d.get('launch_path')))
PY
  done
done
