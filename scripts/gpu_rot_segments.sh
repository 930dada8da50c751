#!/bin/bash
# GPU box: optimize_rotations at 300 nodes with the band cut into P segments (default: the planner's choice)
cd $GRAFT_REPO_ROOT
for P in default 1 2 3 4 6 8; do
  if [ "$P" = default ]; then timeout 120 python3 scripts/prof_rot.py 300 2>&1 | grep "n=300" | sed "s/^/P=default /";
  else SSFM_BAND_SEGMENTS=$P timeout 120 python3 scripts/prof_rot.py 300 2>&1 | grep "n=300" | sed "s/^/P=$P /"; fi
done
SSFM_PLAN_TIMING=1 timeout 120 python3 scripts/prof_rot.py 300 2>&1 | grep -v amdgpu | tail -12
