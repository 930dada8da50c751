# A/B of the point-kernel variants on the GPU box: per-kernel hipEvent averages next to the step time
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "3 1" "6 1" "8 1" "3 2" "3 3" "6 2" "6 3"; do set -- $v
SSFM_PL_UNROLL=$1 SSFM_BS_GROUP=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scale-probe 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('PL=$1 BS=$2', '%.4g obs/s'%d['value'], '%.4f ms'%d['ms_per_step'], 'lin %.1f backsub %.1f us'%(k['k_point_lin']['avg_us'], k['k_point_backsub']['avg_us']), 'iters', d['config']['lm_iterations_per_step'])"
done; done
