cd /tmp && export TMPDIR=/tmp
CHECK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_retri_ao -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 3 > /dev/null 2>&1
F=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_retri_ao -name "*kernel_stats.csv" | head -1); head -4 $F | cut -c1-50,230-330
