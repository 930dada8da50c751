cd /tmp && export TMPDIR=/tmp
export CHECK=0 SSFM_DETERMINISTIC=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_det -o det -- python3 $GRAFT_REPO_ROOT/scripts/dev/ring.py config2 > $GRAFT_REPO_ROOT/gpurun_out/prof_det.log 2>&1
F=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_det -name "*kernel_stats.csv" | head -1); head -16 $F | cut -c1-60,200-400
T=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_det -name "*kernel_trace.csv" | head -1); cp $T $GRAFT_REPO_ROOT/gpurun_out/det_kernel_trace.csv; wc -l $T
