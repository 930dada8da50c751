# rocprofv3 kernel stats of bench.py at config 2 with SSFM_DETERMINISTIC=1 (GPU box):  bash scripts/r06/prof_det.sh
cd /tmp && export TMPDIR=/tmp
export SSFM_DETERMINISTIC=${1:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale-probe --no-side-paths --steps 5 --warmup 1 --detail /tmp/d.json > /tmp/prof_det.log 2>&1
F=$(find /tmp/prof_det -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("ssfm::", "")
    print(f"{n:40s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.2f} us")
PY
