#!/bin/bash
# rocprofv3 kernel stats of the pairwise RANSAC (both modes), the pose-graph solvers, Retriangulate and the focal search -> gpurun_out/prof_misc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_misc
cd $R
python scripts/bench_pairwise.py 1999000 100000 1 > gpurun_out/prof_misc/pairwise_full.json 2> gpurun_out/prof_misc/pairwise_full.err
python scripts/bench_pairwise.py 200000 100000 0 > gpurun_out/prof_misc/pairwise_fixed.json 2>> gpurun_out/prof_misc/pairwise_full.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_misc/trace -o pw -- python3 scripts/bench_pairwise.py 200000 100000 1 > gpurun_out/prof_misc/pw_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_misc/fixed -o pw -- python3 scripts/bench_pairwise.py 100000 100000 0 > gpurun_out/prof_misc/pw_fixed.log 2>&1
find gpurun_out/prof_misc -name "*kernel_stats.csv" | head
