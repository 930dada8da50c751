#!/bin/bash
# Retriangulate only: kernel stats + the VALU / wait counters (the last two commands of scripts/gpu_final_r05.sh pmc), for a kernel change after the round's PMC pass.
TAG=${1:-r05ze}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out
first_csv() { find "$1" -name "$2" 2>/dev/null | head -1; }
cd /tmp && export TMPDIR=/tmp
CHECK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_retri_${TAG} -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 1 > $OUT/prof_retri_${TAG}.log 2>&1
F=$(first_csv $OUT/prof_retri_${TAG} "*kernel_stats.csv"); [ -n "$F" ] && { cp "$F" $OUT/${TAG}_retri_rocprofv3_kernel_stats.csv; head -3 "$F" | cut -c1-60,230-330; }
CHECK=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_retri_${TAG} -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 1 > $OUT/pmc_retri_${TAG}.log 2>&1
F=$(first_csv $OUT/pmc_retri_${TAG} "*counter_collection.csv"); [ -n "$F" ] && cp "$F" $OUT/${TAG}_retri_counter_collection.csv
tail -1 $OUT/pmc_retri_${TAG}.log
