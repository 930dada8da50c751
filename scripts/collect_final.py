"""Copies the summaries of scripts/gpu_final.sh from gpurun_out/ into profiles/ (tracked) and derives the JSON files bench.py reads:
  profiles/<tag>_rocprofv3_kernel_stats.csv, <tag>_scale_rocprofv3_kernel_stats.csv, <tag>_retri_rocprofv3_kernel_stats.csv,
  profiles/<tag>_pmc_per_kernel_avg.json, <tag>_pmc_traffic.json (scripts/pmc_summary.py), profiles/r05_pmc_retriangulate.json, profiles/<tag>_bench*.json
Usage: python scripts/collect_final.py <tag>"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
for name in (f"{tag}_rocprofv3_kernel_stats.csv", f"{tag}_scale_rocprofv3_kernel_stats.csv", f"{tag}_retri_rocprofv3_kernel_stats.csv",
             f"{tag}_ragged14_rocprofv3_kernel_stats.csv", f"{tag}_ragged8_rocprofv3_kernel_stats.csv", f"{tag}_rot4000_rocprofv3_kernel_stats.csv"):
    if os.path.exists(os.path.join(G, name)): shutil.copy(os.path.join(G, name), os.path.join(P, name)); print("copied", name)
for v in ("", "_2ranks_host", "_focalfree", "_spherical"):
    src = os.path.join(G, f"bench_{tag}{v}.json")
    if os.path.exists(src) and os.path.getsize(src) > 0: shutil.copy(src, os.path.join(P, f"{tag}_bench{v}.json")); print("copied", os.path.basename(src))
    src = os.path.join(G, f"bench_{tag}{v}_detail.json")
    if os.path.exists(src) and os.path.getsize(src) > 0: shutil.copy(src, os.path.join(P, f"{tag}_bench{v}_detail.json")); print("copied", os.path.basename(src))
if os.path.exists(os.path.join(G, f"pmc1_{tag}", "ba_counter_collection.csv")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), tag])
rc = os.path.join(G, f"{tag}_retri_counter_collection.csv"); rs = os.path.join(G, f"{tag}_retri_rocprofv3_kernel_stats.csv")
if os.path.exists(rc) and os.path.exists(rs):
    acc = {}
    for r in csv.DictReader(open(rc)):
        if "retriangulate_trace" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    us = None
    for r in csv.DictReader(open(rs)):
        if "retriangulate_trace" in r["Name"]: us = float(r["AverageNs"]) / 1e3
    out = {"kernel": "k_retriangulate_trace", "avg_launch_us_rocprof": us, "counters_per_launch": avg, "source": f"profiles/{tag}_retri_* (rocprofv3 --pmc pass of scripts/gpu_final.sh), not measured in the bench run"}
    if us and "SQ_INSTS_VALU" in avg:
        out["valu_wave_instructions_per_launch"] = avg["SQ_INSTS_VALU"]
        out["valu_issue_frac"] = avg["SQ_INSTS_VALU"] / (256 * us * 1e-6 * 2.4e9)      # one wave64 VALU op per SIMD per 4 cycles: 256 CUs x 4 SIMDs x clk / 4
        if "SQ_WAVE_CYCLES" in avg and "SQ_WAIT_ANY" in avg: out["wait_any_frac_of_wave_cycles"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
        if "SQ_THREAD_CYCLES_VALU" in avg and "SQ_ACTIVE_INST_VALU" in avg and avg["SQ_ACTIVE_INST_VALU"] > 0: out["active_lanes_of_64"] = avg["SQ_THREAD_CYCLES_VALU"] / avg["SQ_ACTIVE_INST_VALU"]
    json.dump(out, open(os.path.join(P, f"{tag[:3]}_pmc_retriangulate.json"), "w"), indent=1, sort_keys=True); print(out)
    shutil.copy(rc, os.path.join(P, f"{tag}_retri_counter_collection.csv"))
