cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do for v in NIMG_NO_FORK_GROUPS=1 BASE=1; do
r=$(env $v python bench.py --steps 60 --warmup 15 --no-side-workloads --no-cpu-baseline --no-parity-mode --no-dp1-nccl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f %s' % (d['ms_per_step'], d.get('launch_path')))")
echo "$v $r"; done; done
