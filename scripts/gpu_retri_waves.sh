#!/bin/bash
# GPU box: Retriangulate trace kernel at 1 / 2 / 3 waves per SIMD (SSFM_RETRI_WAVES), kernel time from rocprofv3.
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in 1 2 3; do
  CHECK=0 SSFM_RETRI_WAVES=$w timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_retri_w$w -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 2 > $OUT/prof_retri_w$w.log 2>&1
  F=$(find $OUT/prof_retri_w$w -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "waves $w:"; if [ -n "$F" ]; then grep "k_retriangulate_trace" "$F" | cut -d, -f1-4 | cut -c1-60,300-; grep "^trace" $OUT/prof_retri_w$w.log; else tail -3 $OUT/prof_retri_w$w.log; fi
done
