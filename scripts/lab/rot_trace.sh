# kernel timeline of one optimize_rotations call at 300 cameras (GPU box)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/rot_trace
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/rot_trace -o rot -- python3 $GRAFT_REPO_ROOT/scripts/dev/rot_time.py > $OUT/rot_trace.log 2>&1
tail -3 $OUT/rot_trace.log
