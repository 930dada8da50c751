# usage: bash scripts/lab/ab_env_generic.sh VAR "v1 v2 ..." [kernel substring]      (GPU box; hipEvent averages of bench.py at config 2)
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; K=${3:-schur_gram}
for v in $VALS $VALS; do echo "$VAR=$v"; env $VAR=$v python bench.py --steps 10 --warmup 2 --no-side-paths --no-scale-probe --no-cpu-baseline 2>/dev/null | tail -1 | K=$K python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('  ms/step', round(d['ms_per_step'],4), 'value %.4e'%d['value'])
for k,v in (d.get('kernels') or {}).items():
    if os.environ['K'] in k: print('  ', k, v)
"; done
