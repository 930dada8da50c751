#!/bin/bash
# rocprofv3 kernel stats + PMC of the pose-graph / Retriangulate / focal-search / pairwise-RANSAC kernels.  Usage: bash scripts/gpu_profile_misc.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/misc_${TAG} -o misc -- python3 scripts/prof_misc_workload.py > $OUT/misc_${TAG}.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/misc_pmc_${TAG} -o misc -- python3 scripts/prof_misc_workload.py > $OUT/misc_pmc_${TAG}.log 2>&1
head -30 $(find $OUT/misc_${TAG} -name "*kernel_stats.csv" | head -1)
tail -2 $OUT/misc_${TAG}.log $OUT/misc_pmc_${TAG}.log
