#!/bin/bash
# GPU box: pose-graph solves under rocprofv3 (kernel stats).  Usage: bash scripts/gpu_prof_rot.sh <tag> [n]
TAG=${1:-r03}; N=${2:-300}
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 120 python3 $GRAFT_REPO_ROOT/scripts/prof_rot.py 300 2000 4000 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_rot -o rot -- python3 $GRAFT_REPO_ROOT/scripts/prof_rot.py $N > $OUT/prof_${TAG}_rot.log 2>&1
F=$(find $OUT/prof_${TAG}_rot -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then head -24 "$F"; cp "$F" $OUT/${TAG}_rot${N}_rocprofv3_kernel_stats.csv; else tail -5 $OUT/prof_${TAG}_rot.log; fi
