cd $GRAFT_REPO_ROOT
for v in 2 1 2 1; do echo "LPP=$v"; SSFM_BACKSUB_LPP=$v python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', d['ms_per_step'], 'value %.4e'%d['value'])
pk=d.get('roofline_per_kernel') or {}
for k,v in pk.items():
    if 'backsub' in k or 'point_lin' in k: print(' ', k, v.get('avg_us'))
"; done
