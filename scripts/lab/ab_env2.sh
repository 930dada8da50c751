# A/B of one environment knob with per-kernel times: bash scripts/lab/ab_env2.sh NAME "v1 v2 ..." [bench args]
NAME=$1; VALS=$2; shift 2
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $VALS; do env $NAME=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scale-probe "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$NAME=$v', '%.4g obs/s'%d['value'], '%.4f ms'%d['ms_per_step'], 'iters', d['config']['lm_iterations_per_step'], {n:round(x['avg_us'],1) for n,x in k.items()})"; done; done
