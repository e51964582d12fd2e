cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python bench.py --steps 60 --warmup 15 --no-side-workloads --no-cpu-baseline --no-parity-mode --no-dp1-nccl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('%.3f ms  path=%s  probe=%s  blocks=%s  ring=%.4f  clk=%s W=%s' % (d['ms_per_step'], d.get('launch_path'), c.get('launch_mode_warmup_ms_per_step'), c.get('block_ms_per_step'), d['roofline']['ms_per_launch'], round(d['roofline']['power']['gfx_clock_mhz_mean']), round(d['roofline']['power']['socket_power_w_mean'])))"; done
