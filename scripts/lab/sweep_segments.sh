# configs[4]-size solve over forced segment counts per component (GPU box): bash scripts/lab/sweep_segments.sh "4 6 8 10 14"
cd $GRAFT_REPO_ROOT
for P in $1; do echo "== SSFM_BAND_SEGMENTS=$P"; CHECK=0 SSFM_BAND_SEGMENTS=$P python scripts/dbg_scale.py 2>&1 | grep -E "band_segments|obs/s|k_band|k_sub" ; done
echo "== default"; CHECK=1 python scripts/dbg_scale.py 2>&1 | grep -E "band_segments|obs/s|k_band|k_sub|rel cam"
