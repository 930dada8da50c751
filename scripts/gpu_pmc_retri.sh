#!/bin/bash
# GPU box: PMC counters of the Retriangulate trace kernel.
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
CHECK=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_retri -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 1 > $OUT/pmc_retri.log 2>&1
CHECK=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_retri2 -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 1 > $OUT/pmc_retri2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
for d in ("pmc_retri", "pmc_retri2"):
    fs = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "retriangulate_trace" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: sum(v) / len(v) for k, v in acc.items()})
PY
tail -2 $OUT/pmc_retri2.log
