#!/bin/bash
# Segment-count sweep of the substructured band solve (SSFM_BAND_SEGMENTS) on the 2000 / 4000-node pose graphs and the configs[4]-size BA.  (GPU box)
#   bash scripts/sweep_segments_r04.sh "0 8 10 12 14 16"      (0 = the planner's choice)
cd $GRAFT_REPO_ROOT
for P in ${1:-0 12 16 20 24 28 32 40 48}; do
  echo "== SSFM_BAND_SEGMENTS=$P (0 = planner's choice)"
  if [ $P -eq 0 ]; then unset SSFM_BAND_SEGMENTS; else export SSFM_BAND_SEGMENTS=$P; fi
  python scripts/prof_rot.py 2000 4000 2>&1 | grep "n=" | cut -c1-90
  CHECK=0 python scripts/dev/scale.py 2>&1 | grep -E "band_segments|obs/s" | sed -e "s/.*'band_half_width'/'band_half_width'/" | cut -c1-200
done
