#!/bin/bash
# quick look at the pairwise RANSAC: parity tests, end-to-end rate, kernel time.  Usage: bash scripts/gpu_pairwise_quick.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 python -m pytest tests/test_ransac_trace_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python scripts/bench_pairwise.py 400000 100000 1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pw_${TAG} -o pw -- python3 scripts/bench_pairwise.py 200000 100000 1 > $OUT/pw_${TAG}.log 2>&1
head -4 $(find $OUT/pw_${TAG} -name "*kernel_stats.csv" | head -1) | cut -c1-200
