# usage (GPU box): bash scripts/r06/ab_mode.sh "<bench.py flags>" "ENV=.." ...  -- bench.py with the given flags under each environment; prints ms/step and the kernel averages
cd $GRAFT_REPO_ROOT
FLAGS="$1"; shift
for e in "$@"; do
  echo "== [$e]"
  env $e python3 bench.py --steps 20 --warmup 3 --no-side-paths --no-scale-probe --no-cpu-baseline --detail /tmp/d.json $FLAGS 2>/tmp/err.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  ms/step %.4f  value %.4e  us/iter(kernels) %s' % (d['ms_per_step'], d['value'], d.get('kernel_us_per_lm_iteration')))
print('  ', ' '.join('%s=%.1f' % (k.replace('k_',''), v) for k, v in (d.get('kernel_avg_us') or {}).items()))
" || tail -3 /tmp/err.txt
done
