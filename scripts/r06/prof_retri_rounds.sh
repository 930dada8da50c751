# rocprofv3 kernel trace of Retriangulate (trace mode) launch by launch:  bash scripts/r06/prof_retri_rounds.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rt -o rt -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 2 > /tmp/prof_rt.log 2>&1
F=$(find /tmp/prof_rt -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "retriangulate_trace" in r["Kernel_Name"]]
for r in rows[-8:]:
    print(r["Kernel_Name"][:40], "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
