#!/bin/bash
# rocprofv3 kernel stats of optimize_rotations at N nodes (default 4000) -> gpurun_out/<tag>_rot<N>_rocprofv3_kernel_stats.csv.  Usage (on the GPU box): bash scripts/prof_rot_rocprof.sh <tag> [N]
TAG=${1:-r04k}; N=${2:-4000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rot_${TAG} -o rot -- python3 scripts/prof_rot.py $N > gpurun_out/rot_${TAG}.log 2>&1
F=$(find gpurun_out/rot_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/${TAG}_rot${N}_rocprofv3_kernel_stats.csv && head -16 "$F" | cut -c1-160
tail -2 gpurun_out/rot_${TAG}.log
