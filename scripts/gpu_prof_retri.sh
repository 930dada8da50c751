#!/bin/bash
# GPU box: Retriangulate under rocprofv3 (kernel stats).  Usage: bash scripts/gpu_prof_retri.sh <tag>
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
CHECK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_retri -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 2 > $OUT/prof_${TAG}_retri.log 2>&1
F=$(find $OUT/prof_${TAG}_retri -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then head -8 "$F"; cp "$F" $OUT/${TAG}_retri_rocprofv3_kernel_stats.csv; else tail -5 $OUT/prof_${TAG}_retri.log; fi
