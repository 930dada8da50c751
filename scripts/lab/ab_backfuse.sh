cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do echo "SSFM_BACK_FUSE=$v"; SSFM_BACK_FUSE=$v python bench.py --steps 10 --warmup 2 --no-side-paths --no-scale-probe 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), 'value %.4e'%d['value'], 'parity', d['parity_vs_oracle']['max_rel_camera'], d['parity_vs_oracle']['iterations_gpu'])
pk=d.get('kernels') or {}
for k,v in pk.items():
    if 'band_back' in k or 'band_chol' in k: print(' ', k, v)
"; done
