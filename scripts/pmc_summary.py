"""Per-kernel averages of the rocprofv3 --pmc passes written by scripts/gpu_bench_profile.sh, and the HBM-traffic figure
bench.py reports as roofline.traffic.  Usage: python scripts/pmc_summary.py <tag>   (reads gpurun_out/pmc{1,2,3}_<tag>)
FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE counts 64-B requests for 128-B fetches, hence the x2
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated there and is reported as is."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for i in (1, 2, 3, 4):
    f = os.path.join(ROOT, "gpurun_out", f"pmc{i}_{tag}", "ba_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for row in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0].replace("ssfm::", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items() if k.startswith("k_")}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_per_kernel_avg.json"), "w"), indent=1, sort_keys=True)
traffic = {"_note": "(2*FETCH_SIZE + WRITE_SIZE) KB -> bytes per launch, separate --pmc passes (profiles/%s_pmc_per_kernel_avg.json); "
                    "gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied; WRITE_SIZE uncalibrated" % tag}
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        traffic[re.sub(r"<.*>", "", k)] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
try:
    import subprocess, datetime
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "unknown"
sys.path.insert(0, ROOT)
from bench import kernel_source_digest
traffic["_meta"] = {"tag": tag, "kernel_source_sha16": kernel_source_digest(), "commit_at_collection": commit, "collected": datetime.date.today().isoformat(),
                    "command": "bench.py --no-cpu-baseline --no-scale-probe --no-side-paths --steps 1 --warmup 0 under rocprofv3 --kernel-trace --pmc <one counter group per pass>"}
traffic["_mfma_mops_f64"] = {re.sub(r"<.*>", "", k): d["SQ_INSTS_VALU_MFMA_MOPS_F64"] for k, d in out.items() if d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) > 0}   # per launch; one MOP = 512 flop
traffic["_valu_wave_instructions"] = {re.sub(r"<.*>", "", k): d["SQ_INSTS_VALU"] for k, d in out.items() if "SQ_INSTS_VALU" in d}   # per launch, all waves
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for k in sorted(out):
    d = out[k]
    print(f"{k:28s} fetch {2*d.get('FETCH_SIZE',0)/1024:8.2f} MB  write {d.get('WRITE_SIZE',0)/1024:8.2f} MB  "
          f"wave_cycles {d.get('SQ_WAVE_CYCLES',0):12.0f}  wait_any {d.get('SQ_WAIT_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1):5.2f}  valu {d.get('SQ_INSTS_VALU',0):10.0f}")
