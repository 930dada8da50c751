#!/bin/bash
# GPU box: the reduced solve at the configs[4] size by the number of band segments per component (default: the planner's choice).  Usage: bash scripts/gpu_segments_sweep.sh
cd $GRAFT_REPO_ROOT
for P in default 4 6 8 12 16 24; do
  if [ "$P" = default ]; then unset SSFM_BAND_SEGMENTS; else export SSFM_BAND_SEGMENTS=$P; fi
  CHECK=0 timeout 200 python scripts/dev/scale.py 2>&1 | grep -E "band_segments|k_band_chol_v2|k_sub_sep_chain|k_sub_spike|k_sub_sep_assemble|k_band_back|k_sub_apply|obs/s" | tr '\n' ' ' | sed "s/^/P=$P /" | cut -c1-900; echo
done
