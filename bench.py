#!/usr/bin/env python3
"""bench.py -- BA observations/second on the BASELINE.json configs[1] workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path: one `SfM::Optimize()`-equivalent solve (device LM loop: linearise +
Schur assembly + reduced solve + back-substitution + candidate cost, every iteration until Ceres' termination
tests fire) of the 300-camera / 100k-point / 600k-observation synthetic circle, general BA (6-dof cameras,
camera 0 fixed, focal fixed -- the calibrated configuration of configs[1]).  The flattened problem is resident in
HBM before the timed region (ssfm_ba_create); each step restores the initial parameters on the device
(ssfm_ba_reset) and runs ssfm_ba_run.  value = M * n_LM / t  with n_LM = linearisations over the K steps
(SURVEY.md 8d), whole job, max time over ranks.

N > 1: points are sharded over ranks (cameras replicated), one RCCL all-reduce of the partial reduced system per
LM iteration.  Default ("strong"): BASELINE.json's metric is THIS 300-camera problem at 1/2/4/8 GPUs, so the problem stays
fixed and is sharded N ways; the reduced solve is replicated (about half of an N = 1 iteration), which caps the speed-up
near 2x -- that is what the line reports.  The same JSON line also carries `configs4_sharded`: the BASELINE
configs[4] problem (4000 cameras / 1.5 M points / 12 M observations, the 8-GPU config) sharded over the same N ranks.
For N > 1 the line also carries `pairwise_sharded`: BASELINE configs[3] (1 999 000 image pairs x 500 correspondences) over the same N ranks.
At N = 1 `roofline_configs4` prices every data-parallel kernel at the configs[4] SIZE (12 M observations on this one GPU: SURVEY 8d's "real HBM test").
`--scaling weak` (opt-in): the circle grows with the job instead -- N x 300 cameras / N x 100k points / N x 600k observations
from the same generator rule (stride round(Nc/75), i.e. 4N rings of 75 cameras), every rank keeps 600k observations.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); the copy bandwidth of the box is MEASURED in every run (hbm_copy_GBs, ssfm_debug_copy_bandwidth)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 x 2.4 GHz)


def _pick_profiles():
    """the committed rocprofv3 summaries this run quotes (scripts/gpu_final.sh + collect_final.py + pmc_summary.py): the PMC file whose recorded kernel-source digest
    equals today's, else the newest by tag (bench.py then prints pmc_stale = true); the kernel-stats file and the Retriangulate counters of the same tag"""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    pick = None
    for c in cands:
        try:
            if (json.load(open(c)).get("_meta") or {}).get("kernel_source_sha16") == kernel_source_digest():
                pick = c
        except Exception:
            pass
    if pick is None and cands:
        pick = cands[-1]
    if pick is None:
        return os.path.join("profiles", "none_pmc_traffic.json"), os.path.join("profiles", "none_rocprofv3_kernel_stats.csv"), None
    tag = os.path.basename(pick).split("_")[0]
    retri = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_retriangulate.json")))
    return os.path.relpath(pick, ROOT), os.path.join("profiles", f"{tag}_rocprofv3_kernel_stats.csv"), (os.path.relpath(retri[-1], ROOT) if retri else None)


def kernel_source_digest():
    """sha256[:16] over the BA kernel sources (the files a PMC profile of bench.py depends on); a committed PMC file carries the digest of the tree it
    profiled in _meta.kernel_source_sha16 and bench.py prints pmc_stale = true when today's differs (the GPU box has no .git to ask)"""
    import hashlib
    d = os.path.join(ROOT, "spherical_sfm_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")) and not f.startswith(("ransac", "lomsac", "retriangulate", "sampson", "rotavg", "line_search")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


PMC_PROFILE, ROCPROF_STATS, PMC_RETRI = _pick_profiles()


def rocprof_avg_us():
    """{kernel name without template arguments: average launch duration in us} from the committed rocprofv3 summary (None if it is not there)"""
    import csv
    import re
    path = os.path.join(ROOT, ROCPROF_STATS)
    if not os.path.exists(path):
        return None
    acc = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void ", "", r["Name"]).split("(")[0].replace("ssfm::", "")
        name = re.sub(r"<.*>", "", name)
        c, t = int(r["Calls"]), float(r["TotalDurationNs"])
        a = acc.setdefault(name, [0, 0.0]); a[0] += c; a[1] += t
    return {k: v[1] / v[0] / 1e3 for k, v in acc.items() if v[0]}


# ---------------------------------------------------------------------------------------------------------------------
# The record.  bench.py computes a large `detail` dictionary (per-kernel tables, side paths, notes); that goes to
# bench_detail.json.  The LAST stdout line is compact_record(detail): numbers and short identifiers only, well under
# 6 KB, strict JSON.  tests/test_bench_record_cpu.py builds it from a committed detail file and checks size and keys.
RECORD_LIMIT = 6000
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def _num(x, sig=6):
    """a float rounded to `sig` significant digits (ints, None, bools and strings pass; NaN / inf become None: strict JSON has neither)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def _pick(d, keys, sig=6):
    """{k: d[k]} for the scalar entries of `keys` present in d, rounded"""
    if not isinstance(d, dict):
        return None
    return {k: _num(d[k], sig) for k in keys if k in d and not isinstance(d[k], (dict, list))}


def compact_record(d):
    """the one-line record of a run from the detail dictionary `d` (see main()): every value a number, bool, null or a short identifier"""
    cfg = d.get("config", {})
    rec = {k: _num(d.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    rec["config"] = {"workload": str(cfg.get("workload", ""))[:200], "camera_dof": cfg.get("camera_dof"), "lm_iterations_per_step": _num(cfg.get("lm_iterations_per_step")),
                     "sharding": cfg.get("sharding"), "comm": str(cfg.get("comm", ""))[:40], "reduced_solver": str(cfg.get("reduced_solver_short", "band-cholesky"))[:40],
                     "pcg_sweeps": cfg.get("pcg_sweeps")}
    r = d.get("roofline") or {}
    rec["roofline"] = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes_per_launch", "algorithmic_flop_per_launch", "executed_frac"))
    pm = d.get("pmc_profile") or {}
    rec["roofline"]["pmc_tag"] = pm.get("tag"); rec["roofline"]["pmc_stale"] = pm.get("stale")
    if d.get("roofline_hbm"):
        rec["roofline_hbm"] = _pick(d["roofline_hbm"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_measured_copy", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us"))
    if d.get("roofline_lm_iteration"):
        rec["roofline_lm_iteration"] = _pick(d["roofline_lm_iteration"], ("bound", "algorithmic_bytes", "avg_ms", "achieved", "peak", "unit", "frac", "frac_of_measured_copy"))
    if d.get("cpu_baseline"):
        rec["cpu_baseline"] = _pick(d["cpu_baseline"], ("value", "unit", "cores", "kind", "solve_s", "end_to_end_s", "host_cpus"))
        rec["cpu_baseline"]["sample"] = str(d["cpu_baseline"].get("sample", ""))[:120]
    if d.get("parity_vs_oracle"):
        rec["parity_vs_oracle"] = _pick(d["parity_vs_oracle"], ("max_rel_camera", "max_rel_point", "iterations_gpu", "iterations_cpu"), 3)
    rec["hbm_copy_GBs"] = _num(d.get("hbm_copy_GBs"), 4)
    if d.get("kernels"):
        n_it = max(1.0, float(cfg.get("lm_iterations_per_step") or 1.0))
        n_lm_prof = d.get("lm_iterations_profiled_step") or n_it
        rec["kernel_us_per_lm_iteration"] = _num(sum(v["launches"] * v["avg_us"] for v in d["kernels"].values()) / n_lm_prof, 4)
        rec["kernel_avg_us"] = {k: _num(v["avg_us"], 4) for k, v in d["kernels"].items()}
    if d.get("end_to_end_optimize"):
        rec["end_to_end_optimize"] = _pick(d["end_to_end_optimize"], ("first_s", "median_s", "best_s", "cpu_s", "speedup", "speedup_warm"), 4)
        w = d["end_to_end_optimize"].get("warm")
        if isinstance(w, dict):
            rec["end_to_end_optimize"]["warm_s"] = _num(w.get("gpu_s"), 4)
    if d.get("scaling_model"):
        rec["scaling_model"] = {k: _num(v, 4) for k, v in d["scaling_model"].items() if not isinstance(v, (dict, list, str))}
    # side paths: at most ten scalars
    side = {}
    sp = d.get("side_paths") or {}
    def g(path, sig=4):
        o = sp
        for k in path:
            if not isinstance(o, dict) or k not in o:
                return None
            o = o[k]
        return _num(o, sig) if not isinstance(o, (dict, list)) else None
    for name, path in (("pairwise_lomsac_pairs_per_s", ("pairwise_lomsac", "value")), ("rotation_averaging_ms", ("rotation_averaging", "value")),
                       ("retriangulate_ms", ("retriangulate", "value")), ("deterministic_overhead", ("deterministic_accumulation", "overhead")),
                       ("default_repeats_identical", ("deterministic_accumulation", "default_repeats_identical")),
                       ("irregular_3_to_14_obs_per_s", ("ba_irregular", "tracks_3_to_14", "value")), ("irregular_3_to_14_grouped", ("ba_irregular", "tracks_3_to_14", "grouped_fraction_of_observations")),
                       ("irregular_3_to_8_obs_per_s", ("ba_irregular", "tracks_3_to_8", "value")),
                       ("irregular_1000_cameras_obs_per_s", ("ba_irregular", "tracks_3_to_8_1000_cameras", "value")), ("pipeline_configs2_gpu_ms", ("pipeline_configs2", "gpu_ms_total"))):
        v = g(path)
        if v is not None:
            side[name] = v
    if side:
        rec["side_paths"] = side
    if d.get("scale_probe_configs4_size_one_gpu"):
        rec["scale_probe_configs4_size_one_gpu"] = _pick(d["scale_probe_configs4_size_one_gpu"], ("value", "unit", "ms_per_solve", "lm_iterations"), 4)
        li = (d.get("roofline_configs4") or {}).get("lm_iteration")
        if li:
            rec["scale_probe_configs4_size_one_gpu"]["frac_hbm_lm_iteration"] = _num(li.get("frac_hbm"), 4)
    for k, keys in (("timing_without_collective", ("ms_per_lm_iteration", "ms_per_lm_iteration_with_collectives")),
                    ("configs4_sharded", ("value", "unit", "ms_per_solve", "lm_iterations", "n_gpus")), ("pairwise_sharded", ("value", "unit", "seconds", "n_gpus"))):
        if d.get(k):
            rec[k] = _pick(d[k], keys, 4)
    rec["detail"] = d.get("detail_path")
    return rec


def record_line(d):
    """compact_record(d) as the single stdout line; raises if it is not a record the driver can keep (size, strict JSON, required keys)"""
    rec = compact_record(d)
    line = json.dumps(rec, allow_nan=False, separators=(", ", ": "))
    missing = [k for k in REQUIRED_KEYS if rec.get(k) is None and k != "vs_baseline"]
    if missing or "\n" in line or len(line) >= RECORD_LIMIT:
        raise ValueError(f"bench record unusable: {len(line)} chars, missing {missing}")
    return line


def _jsonable(o):
    """detail dictionary with NaN / inf replaced by None (strict JSON) and numpy scalars unwrapped"""
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, float):
        return o if (o == o and o not in (float("inf"), float("-inf"))) else None
    if hasattr(o, "item") and not isinstance(o, (str, bytes)):
        try:
            return _jsonable(o.item())
        except Exception:
            return str(o)
    return o


def pair_kernel_flops(pairs):
    """k_schur_pairs2, per (observation of camera c, observation of camera c2) pair of one point: two re-linearisations
    (reprojection + analytic 2x6 camera / 2x3 point blocks, ~150 flop each) and the DCxDC block update through the 2x2 core
    Jc_i^T (Jp_i Vs Jp_j^T) Jc_j (126 multiply-adds = 252 flop at DC = 6)."""
    return pairs * (2 * 150.0 + 252.0)


PAIR_USEFUL_FLOP = 252.0       # the DCxDC block update through the 2x2 core: 126 multiply-adds at DC = 6
PAIR_RELIN_FLOP = 2 * 150.0    # what the kernel spends on top: both observations of a pair are re-linearised (each observation K - 1 times per iteration)


def kernel_rooflines(kern, M, nP, nnzb, Nc, dc, pairs, n_lm, world=1, focal_free=False):
    """Per-kernel HBM fractions of the data-parallel kernels of one LM iteration from their hipEvent durations (SURVEY 8d algorithmic bytes, split per
    pass as DESIGN.md 4 states them), and the pair kernel against the FP64 vector peak by USEFUL flops next to the modelled total."""
    per = {
        # lane per point: pixels 16 + camera ids 8 per observation; X 24 + Jacobi scales 24 read, record PS 96 + g_p 24 written per point
        "k_point_lin": 24.0 * M + 168.0 * nP,
        # diagonal blocks / J_c^T r: pixels + ids per observation, X + PS per point, diagonal S blocks + rhs written
        "k_cam_sums2": 24.0 * M + 120.0 * nP + Nc * (dc * dc + dc) * 8.0,
        "k_schur_pairs2": pair_kernel_bytes(M, nP, nnzb, Nc, dc),
        "k_schur_gram": gram_kernel_bytes(M, nP, nnzb, Nc, dc, focal_free),
        # back substitution + candidate + both costs in one sweep: pixels + ids per observation; X, PS, g_p read, candidate X written per point
        "k_point_backsub": 24.0 * M + 168.0 * nP,
        # the same pass for grouped points (lane = observation): pixels only (the observations of a group are consecutive, no ids); X, scales, g_p, V^-1 read, candidate X written
        "k_gram_backsub": 16.0 * M + 144.0 * nP,
    }
    out = {}
    for k, b in per.items():
        us = kern.get(k, {}).get("avg_us")
        if us is None or not us == us:
            continue
        b = b / world
        out[k] = {"avg_us": us, "launches_per_lm_iteration": kern[k]["launches"] / max(1, n_lm), "algorithmic_bytes_per_launch": b,
                  "achieved_GBs": b / (us * 1e-6) / 1e9, "frac_hbm": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    us = kern.get("k_schur_pairs2", {}).get("avg_us")
    if us is not None and us == us:
        p = pairs / world
        out["k_schur_pairs2"]["fp64_vector"] = {
            "pairs_per_launch": p, "useful_flop_per_pair": PAIR_USEFUL_FLOP, "modelled_flop_per_pair": PAIR_USEFUL_FLOP + PAIR_RELIN_FLOP,
            "useful_TFLOPs": p * PAIR_USEFUL_FLOP / (us * 1e-6) / 1e12, "useful_frac": p * PAIR_USEFUL_FLOP / (us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "modelled_TFLOPs": p * (PAIR_USEFUL_FLOP + PAIR_RELIN_FLOP) / (us * 1e-6) / 1e12,
            "modelled_frac": p * (PAIR_USEFUL_FLOP + PAIR_RELIN_FLOP) / (us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "note": "flop counts are a hand model of the kernel's arithmetic (DESIGN.md 4), not a counter; durations are measured"}
    us = kern.get("k_schur_gram", {}).get("avg_us")
    if us is not None and us == us:
        ex, useful = gram_kernel_flops(M / world, nP / world, pairs / world, dc)
        out["k_schur_gram"]["fp64"] = {
            "executed_TFLOPs": ex / (us * 1e-6) / 1e12, "executed_frac": ex / (us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "useful_TFLOPs": useful / (us * 1e-6) / 1e12, "useful_frac": useful / (us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "note": "FP64 matrix and vector instructions share one peak on gfx950 (78.6 TFLOP/s) and, measured, one pipe; flop counts are a hand model "
                    "(DESIGN.md 4), durations are measured.  When some points are not grouped the pair kernel runs too and these figures overcount."}
    return out


def pass_rooflines(kern, M, nP, world, gram, copy_gbs):
    """SURVEY 8d's B_alg = 72 B/obs + 240 B/pt split over the logical passes of one LM iteration and the kernels that run them -- the parts ADD UP to B_alg:
         A    linearise + assemble     obs 16 + ids 8 per observation; X 24 read, V^-1 + g_p 72 written per point          = 24 M +  96 nP
         C+D  back-substitute + cost   obs 24 twice per observation; V^-1 + g_p 72 read, dp 24 written, X_new 24 read,
                                       24 committed per point                                                               = 48 M + 144 nP
       durations: hipEvent averages of this run (NaN when a kernel did not run)"""
    a_k = ["k_point_lin", "k_schur_gram"] if gram else ["k_point_lin", "k_cam_sums2", "k_schur_pairs2"]
    c_k = [k for k in ("k_point_backsub", "k_gram_backsub") if kern.get(k, {}).get("launches", 0)]
    def us_of(names):
        t = 0.0
        for k in names:
            u = kern.get(k, {}).get("avg_us"); n = kern.get(k, {}).get("launches", 0)
            if n and u == u: t += u
        return t
    out = {}
    for name, ks, b in (("A_linearise_assemble", a_k, 24.0 * M + 96.0 * nP), ("CD_backsubstitute_candidate_cost", c_k, 48.0 * M + 144.0 * nP)):
        us = us_of(ks); b = b / world
        g = b / (us * 1e-6) / 1e9 if us > 0 else None
        out[name] = {"kernels": ks, "algorithmic_bytes": b, "sum_of_avg_us": us, "achieved_GBs": g, "frac_hbm": (g / HBM_PEAK_GBS) if g else None,
                     "frac_of_measured_copy": (g / copy_gbs) if (g and copy_gbs) else None}
    out["sum_of_algorithmic_bytes"] = sum(v["algorithmic_bytes"] for v in out.values() if isinstance(v, dict))
    out["B_alg"] = (72.0 * M + 240.0 * nP) / world
    return out


SHARDED_KERNELS = ("k_point_lin", "k_cam_sums2", "k_schur_pairs2", "k_schur_gram", "k_point_backsub", "k_gram_backsub")


def scaling_model(world, n_lm_per_step):
    """DESIGN.md 6's strong-scaling model, evaluated from the committed N = 1 bench line of the same workload (the newest profiles/r??*_bench.json): the point-major
    kernels shard with 1 / N, the reduced solve and the small per-camera kernels are replicated, and every LM iteration pays two all-reduces (the reduced system and
    the candidate-cost scalars, ~20 us each over xGMI -- an assumption until a multi-GPU box has been measured).  Printed next to the measured value so that the first
    real curve can be read against it; it is NOT a measurement."""
    import glob
    import re
    # round 6: the committed record lines are compact; the kernel table lives in the detail file next to them (<tag>_bench_detail.json); older rounds: the line itself
    cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_bench_detail.json")) if re.search(r"r\d\d[a-z]?_bench_detail\.json$", f))
    if not cands:
        cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_bench.json")) if re.search(r"r\d\d[a-z]?_bench\.json$", f))
    if not cands:
        return {"note": "no committed N = 1 bench line under profiles/"}
    try:
        base = json.loads(open(cands[-1]).read().strip().splitlines()[-1])
        kern = base["kernels"]; it1 = base["steps"] and base["ms_per_step"] * 1e3 / max(1.0, base["config"]["lm_iterations_per_step"])
    except Exception as e:
        return {"note": f"could not read {cands[-1]}: {e}"}
    nit = max(1.0, base["config"]["lm_iterations_per_step"])                   # the kernel table of a bench line is ONE profiled step
    per_it = lambda k: kern[k]["avg_us"] * kern[k]["launches"] / nit if k in kern and kern[k].get("launches") else 0.0
    lm_kernels = [k for k in kern if kern[k].get("launches", 0) >= 0.9 * nit]
    sharded = sum(per_it(k) for k in lm_kernels if k in SHARDED_KERNELS)
    replicated = sum(per_it(k) for k in lm_kernels if k not in SHARDED_KERNELS)
    gaps = max(0.0, it1 - sharded - replicated)
    coll = 0.0 if world == 1 else 2 * 20.0
    itn = sharded / world + replicated + gaps + coll
    return {"source": os.path.relpath(cands[-1], ROOT), "n1_iteration_us": it1, "sharded_kernels_us": sharded, "replicated_kernels_us": replicated, "dispatch_gaps_us": gaps,
            "assumed_collectives_us": coll, "predicted_iteration_us": itn, "predicted_speedup_vs_n1": it1 / itn if itn > 0 else None,
            "predicted_value_obs_per_s": base["value"] * it1 / itn if itn > 0 else None,
            "note": "model (DESIGN.md 6), not a measurement: sharded / N + replicated + gaps + two assumed 20 us all-reduces per LM iteration"}


def with_copy_fraction(per_kernel, copy_gbs, rocprof=None):
    """adds frac_of_measured_copy (achieved / the device copy bandwidth measured in the same run) next to frac_hbm (achieved / nominal 8 TB/s), and -- when the committed
    rocprofv3 summary of this command is there -- the same fractions from ITS average launch durations (the event brackets of this run include a few us of dispatch
    hand-over per launch: VERDICT r3 #12)"""
    for k, v in per_kernel.items():
        if copy_gbs and v.get("achieved_GBs") is not None:
            v["frac_of_measured_copy"] = v["achieved_GBs"] / copy_gbs
        us = (rocprof or {}).get(k)
        if us:
            g = v["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9
            v["rocprof"] = {"avg_us": us, "achieved_GBs": g, "frac_hbm": g / HBM_PEAK_GBS, "frac_of_measured_copy": (g / copy_gbs) if copy_gbs else None, "source": ROCPROF_STATS}
    return per_kernel


def gram_kernel_bytes(M, nP, nnzb, Nc, dc, focal_free):
    """k_schur_gram (the Schur assembly of points that share their camera list: off-diagonal AND diagonal blocks, right-hand side, focal border):
    every observation's pixel pair once (16 B; no index: the observations of a group are consecutive), every point's X + record PS once
    (24 + 72 B, + 24 with a free focal), every block of S written once, the camera-side vectors (rhs, J_c^T r, diag U, focal border)."""
    return 16.0 * M + (96.0 + (24.0 if focal_free else 0.0)) * nP + nnzb * dc * dc * 8.0 + Nc * 4 * dc * 8.0


GRAM_TILE_FLOP = 2 * 16 * 16 * 4                # one v_mfma_f64_16x16x4_f64
GRAM_OBS_VALU_FLOP = 150.0 + 102.0 + 132.0      # per observation, once: linearisation, half product Y = Jc^T (Jp L) (30 + 72), camera-side sums (Jc^T Jc 84 + two Jc^T v 48)


def gram_kernel_flops(M, nP, pairs, dc):
    """(executed, useful) FP64 flop of one k_schur_gram launch (hand model).  Executed: NT (NT + 1) / 2 tiles x 6 k-steps per 8 points on the
    matrix pipe (6 tiles at DC = 6) + the VALU work per observation.  Useful: the same VALU work (every observation is linearised once now) + the
    entries of the Gram matrix the algorithm asks for: Y_a Y_b^T per (observation, observation) pair (DC x DC x 3 multiply-adds) and the upper
    triangle of Y_a Y_a^T per observation."""
    rows = dc * int(round(M / max(nP, 1)))              # Gram rows of a point with the average number of observations (the synthetic workloads have one K)
    # 16x16 products per k-step: one, three or six by the number of 16-row tiles in use; up to four rows beyond the last full tile (18 or 36 rows: six cameras)
    # go through ceil(rows / 16) 4x4x4 instructions (4 blocks x 128 flop) instead of one more row of tiles
    full, tail = rows // 16, rows % 16
    if full >= 1 and 1 <= tail <= 4:
        tile_flop = (full * (full + 1) // 2) * GRAM_TILE_FLOP + ((rows + 15) // 16) * 512
    else:
        nt = (rows + 15) // 16
        tile_flop = (nt * (nt + 1) // 2) * GRAM_TILE_FLOP
    executed = (nP / 8.0) * 6 * tile_flop + M * GRAM_OBS_VALU_FLOP
    useful = pairs * dc * dc * 3 * 2 + M * (dc * (dc + 1) / 2) * 3 * 2 + M * GRAM_OBS_VALU_FLOP
    return executed, useful


def algorithmic_bytes(M, nP, nnzb, dc, focal_free):
    """per LM iteration 72 B/observation + 240 B/point (+ the S term, reported separately)."""
    per_iter = 72.0 * M + 240.0 * nP
    # pass A = Schur assembly (k_cam_sums2 + k_schur_pairs2): obs 16 + ids 8 per observation; X 24 + V^-1 48 + g 24 (+ Wf 24) per point;
    # S row blocks written
    schur = 24.0 * M + (96.0 + (24.0 if focal_free else 0.0)) * nP + nnzb * dc * dc * 8.0
    return per_iter, schur


def pair_kernel_bytes(M, nP, nnzb, Nc, dc):
    """k_schur_pairs2 alone: every observation's pixel pair once (16 B), every point's X + scaled V^-1 once (24 + 48 B), the
    off-diagonal blocks of S written once.  The pair index lists are an implementation artefact and are not counted."""
    return 16.0 * M + 72.0 * nP + max(nnzb - Nc, 0) * dc * dc * 8.0


def side_paths(ctx):
    """The two other paths of SURVEY.md 8 next to the headline, each with its CPU port timed on a bounded sample and a parity figure
    (side measurements: they never enter `value`):
      * batched pairwise LO-MSAC (estimate_pairwise, examples/spherical_sfm_tools.cpp:300-420) in the reference-trace mode: 16384 pairs x 500
        correspondences, 30 % outliers -- host buffers in, results out (PCIe inclusive);
      * optimize_rotations (src/rotation_averaging.cpp:52-103) on the 300-camera graph of SURVEY 8d."""
    import numpy as np
    from spherical_sfm_amd import ba, synth, ransac, rotavg
    from oracle import oracle as O
    res = {}
    F = 1000.0; THR = (2 / F) ** 2; NC = 500; POOL = 256; P = 16384
    probs = [synth.make_relative_pose_problem(NC, seed=1000 + k, noise=1 / F, outlier_frac=0.3, rotation_deg=1 + (k % 60)) for k in range(POOL)]
    ptr = (np.arange(P + 1, dtype=np.int64) * NC).astype(np.int32)
    # the data estimate_pairwise holds: per-frame feature rays (frame k = the u-rays and the v-rays of pool problem k) + per-pair match lists
    feat_ptr = (np.arange(POOL + 1, dtype=np.int64) * 2 * NC).astype(np.int32)
    feat_rays = np.ascontiguousarray(np.concatenate([np.concatenate([q[0], q[1]]) for q in probs]))
    fr = (np.arange(P) % POOL).astype(np.int32); m0 = np.tile(np.arange(NC, dtype=np.int32), P); m1 = m0 + NC
    run = lambda: ransac.estimate_indexed(ctx, feat_ptr, feat_rays, fr, fr, ptr, m0, m1, THR, min_num_inliers=20)
    run()                                                                               # warm-up: module load, pinned staging buffers
    t = time.perf_counter(); o = run(); dt = time.perf_counter() - t
    k_ms = ransac.last_kernel_ms(ctx)                                                   # device time of the kernels of that call (hipEvent brackets per slab)
    # SURVEY 8d prices this leg in FP64 flop/s of Sampson scoring: every iteration scores its four candidate models against all correspondences of the pair
    # (include/RansacLib/ransac.h:295-303, src/spherical_estimator.cpp:67-78), 48 flop per score (3x3 matrix-vector product 15, E^T v's two components 10, v . Eu 5,
    # two squared norms 7, square + quotient via reciprocal ~11); local optimisation, least squares and the minimal solver are NOT counted (algorithmic flop of the scoring only)
    scores = float(o["iterations"].astype(np.float64).sum()) * 4.0 * NC
    sampson_tflops = scores * 48.0 / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None
    ns = 24; tc = time.perf_counter(); worst = 0.0; same_its = 0
    for k in range(ns):
        r = O.lomsac_pair(probs[k][0], probs[k][1], THR, min_num_inliers=20)
        worst = max(worst, float(np.linalg.norm(O.so3ln(r["R"] @ o["R"][k].T)))); same_its += int(r["iterations"] == int(o["iterations"][k]))
    tc = (time.perf_counter() - tc) / ns
    res["pairwise_lomsac"] = {"workload": f"{P} pairs x {NC} correspondences, 30% outliers, reference-trace LO-MSAC (std::mt19937 streams replayed on the device), ssfm_ransac_batch_indexed: "
                                          f"feature rays of {POOL} frames + match lists from host memory",
                              "value": P / dt, "unit": "pairs/s", "includes_pcie": True, "mean_iterations": float(o["iterations"].mean()),
                              "kernel_ms": k_ms, "kernel_pairs_per_s": (P / (k_ms * 1e-3)) if k_ms > 0 else None,
                              "roofline": {"bound": "fp64 vector", "achieved": sampson_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": (sampson_tflops / FP64_VECTOR_PEAK_TFLOPS) if sampson_tflops else None,
                                           "algorithmic_flop": scores * 48.0, "scores": scores, "flop_per_score": 48,
                                           "note": "algorithmic flop = Sampson scores of the RANSAC iterations only (iterations x 4 models x correspondences x 48), divided by the device time of "
                                                   "ALL kernels of the call (sampling, minimal solver, scoring, local optimisation, final least squares, decomposition); rays sit in LDS, "
                                                   "so the bound is the FP64 vector pipe, not HBM (SURVEY 8d)"},
                              "cpu_baseline": {"value": 1.0 / tc, "unit": "pairs/s", "cores": 1, "kind": "port", "sample": f"the first {ns} pairs"},
                              "parity_vs_oracle": {"max_rotation_error_rad": worst, "pairs_with_identical_iteration_count": same_its, "pairs_checked": ns}}
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(300, 8)
    rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel)
    t = time.perf_counter(); Rg, _, sg = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel); dt = time.perf_counter() - t
    t = time.perf_counter(); Rc, _, sc = O.optimize_rotations(R0, i0, i1, Rrel); tc = time.perf_counter() - t
    err = max(float(np.linalg.norm(O.so3ln(Rg[k] @ Rc[k].T))) for k in range(len(Rg)))
    res["rotation_averaging"] = {"workload": f"optimize_rotations, 300 cameras, {len(i0)} edges, SoftLOne(0.03)", "value": 1e3 * dt, "unit": "ms per call", "higher_is_better": False,
                                 "iterations": sg.get("iterations"), "cpu_baseline": {"value": 1e3 * tc, "unit": "ms per call", "cores": 1, "kind": "port", "sample": "the same graph"},
                                 "parity_vs_oracle": {"max_rotation_error_rad": err, "iterations_cpu": sc.get("iterations")}}
    # ---- SfM::Retriangulate (src/sfm.cpp:156-192) on its own: the trace-replay kernel at the config-2 size, the oracle on 16 threads beside it.  The kernel is one lane
    # per point and latency bound (1563 waves for 1024 SIMDs); its VALU-issue fraction comes from a committed PMC pass like the headline kernel's.
    pr = synth.make_circle(300, 100000, 6, rot_noise_deg=0.0, pixel_noise=0.5)
    ba.retriangulate(ctx, pr)
    t = time.perf_counter(); Xg, ning = ba.retriangulate(ctx, pr); dt = time.perf_counter() - t
    t = time.perf_counter(); Xe, _ = ba.retriangulate(ctx, pr, mode=ba.RETRI_MODE_ENUMERATE); dte = time.perf_counter() - t
    t = time.perf_counter(); Xo, nino = O.retriangulate(pr, 16); tc = time.perf_counter() - t
    nz = Xo.any(1)
    pmc_r = {}
    try:
        pmc_r = json.load(open(os.path.join(ROOT, PMC_RETRI)))
    except Exception:
        pass
    res["retriangulate"] = {
        "workload": "300 cameras x 100 000 points x 6 observations, per-point LO-MSAC of the reference replayed draw for draw (ssfm_retriangulate, trace mode)",
        "value": 1e3 * dt, "unit": "ms per call (host buffers in, points out)", "higher_is_better": False, "points_per_s": 100000 / dt,
        "enumerate_mode_ms": 1e3 * dte,
        "cpu_baseline": {"value": 1e3 * tc, "unit": "ms per call", "cores": 16, "kind": "port", "sample": "the same 100 000 points"},
        "parity_vs_oracle": {"identical_zero_sets": bool(np.array_equal(nz, Xg.any(1))), "identical_inlier_counts": bool(np.array_equal(ning, nino)),
                             "max_rel_point": float((np.linalg.norm(Xg - Xo, axis=1)[nz] / np.linalg.norm(Xo[nz], axis=1)).max())},
        "valu_issue": pmc_r or {"note": "no committed PMC pass found (profiles/r05_pmc_retriangulate.json)"}}
    # ---- order-independent accumulation (csrc/det_acc.h): the metric's problem with SSFM_DETERMINISTIC=1 -- repeated solves bit for bit, and what it costs
    try:
        prob = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
        out = {}
        for mode in ("0", "1"):
            os.environ["SSFM_DETERMINISTIC"] = mode
            adj = ba.BundleAdjuster(ctx, prob)
            adj.run(); runs = []; best = 1e9
            for _ in range(20):
                adj.reset(); t = time.perf_counter(); sa = adj.run(); best = min(best, time.perf_counter() - t)
                cg, pg, _f = adj.download(); runs.append((np.copy(cg), np.copy(pg), sa["final_cost"]))
            adj.close()
            out[mode] = {"ms_per_solve": 1e3 * best, "lm_iterations": sa["iterations"],
                         "repeats_identical_to_the_first": sum(1 for q in runs[1:] if np.array_equal(q[0], runs[0][0]) and np.array_equal(q[1], runs[0][1]) and q[2] == runs[0][2]),
                         "repeats": len(runs) - 1, "first": runs[0]}
        d0, d1 = out["0"].pop("first"), out["1"].pop("first")
        res["deterministic_accumulation"] = {
            "workload": "the metric's problem (300 cameras x 100 000 points x 600 000 observations), resident handle, 20 solves per mode",
            "default_fp64_atomics": out["0"], "SSFM_DETERMINISTIC=1": out["1"], "overhead": out["1"]["ms_per_solve"] / out["0"]["ms_per_solve"] - 1.0,
            "max_rel_camera_between_modes": float(np.abs(d0[0] - d1[0]).max() / np.abs(d0[0]).max()),
            "note": "round 6: every point of this problem sits in a signature group, so the mode stores per-task partial blocks with plain stores and k_finalize_gather folds them in task order (no atomics for the reduced system, no decode launches); long accumulators (integer atomics) for the scalar block; off by default"}
    finally:
        os.environ.pop("SSFM_DETERMINISTIC", None)
    # ---- irregular structure: 300 cameras, 600k observations, RAGGED tracks of 3..14 (and 3..8) consecutive frames, point ids in build_sfm's
    # order (examples/spherical_sfm_tools.cpp:862-955; synth.make_ragged_circle).  Reports what the same LM loop does when the synthetic circle's regularity is gone:
    # grouped fraction (planner: signature sort + cost model), obs/s, where the time goes, parity and the CPU port beside it.
    res["ba_irregular"] = {}
    for ncam_i, nobs_i, max_len in ((300, 600000, 14), (300, 600000, 8), (1000, 3000000, 8)):
        prob = synth.make_ragged_circle(ncam_i, nobs_i, 3, max_len)
        info, _, _, _ = ba.plan(prob)
        adj = ba.BundleAdjuster(ctx, prob)
        adj.reset(); adj.run()
        t = time.perf_counter(); n = 0
        for _ in range(3):
            adj.reset(); sa = adj.run(); n += sa["num_linearizations"]
        dt = time.perf_counter() - t
        adj.set_profiling(True); adj.reset(); sp_ = adj.run(); kt = adj.kernel_times(); adj.set_profiling(False)
        cg, pg, _ = [np.copy(a) if hasattr(a, "copy") else a for a in adj.download()]
        adj.close()
        t = time.perf_counter(); oc, op, of, os_ = O.ba_solve(prob); tcpu = time.perf_counter() - t
        tot = sum(v["total_ms"] for v in kt.values()) or 1.0
        res["ba_irregular"][f"tracks_3_to_{max_len}" + ("" if ncam_i == 300 else f"_{ncam_i}_cameras")] = {
            "workload": f"{ncam_i} cameras x {len(prob.points)} points x {len(prob.obs_cam)} observations, tracks of 3..{max_len} consecutive frames, general BA, focal fixed",
            "value": len(prob.obs_cam) * n / dt, "unit": "obs/s", "ms_per_lm_iteration": 1e3 * dt / n, "lm_iterations": sa["iterations"],
            "grouped_fraction_of_observations": info["num_observations_grouped"] / max(1, info["num_observations_used"]),
            "band_half_width": info["band_half_width"], "band_segments": info["band_segments"], "band_separators": info["band_separators"],
            "share_of_gpu_time": {k: v["total_ms"] / tot for k, v in kt.items() if v["launches"]},
            "avg_us": {k: 1e3 * v["total_ms"] / v["launches"] for k, v in kt.items() if v["launches"]},
            "cpu_baseline": {"value": os_["num_residual_blocks"] * os_["num_linear_solves"] / (os_["t_total_s"] - os_["t_flatten_s"]), "unit": "obs/s", "cores": os_["threads_used"], "kind": "port",
                             "sample": "the same problem, one solve", "wall_s": tcpu},
            "parity_vs_oracle": {"max_rel_camera": float(np.abs(cg - oc).max() / np.abs(oc).max()),
                                 "max_rel_point": float((np.linalg.norm(pg - op, axis=1) / np.linalg.norm(op, axis=1)).max()), "iterations_cpu": os_["iterations"]},
            "note": "one connected ring of cameras laid out in its own circular order (round 5: half-width = longest track - 1, separators by cyclic reduction, DESIGN.md 4b); "
                    "signature groups (k_schur_gram_any: all track lengths in one launch) when the planner's cost model expects them to beat the pair lists -- at 600k observations "
                    "they do not (70 us against 43 + 20 us), at 3 M they do (160 against 162 + 62 us)"}
    # ---- the drivers' whole stage sequence at BASELINE configs[2] size (500 frames / 170 000 points / 1.02 M observations, shared focal free, -generalba):
    # Optimize -> Retriangulate -> Optimize -> unfix t -> Optimize -> Normalize -> Retriangulate -> Optimize -> Normalize (examples/run_spherical_sfm_uncalib.cpp:176-222)
    # through the C++ mirror (spherical_sfm_amd/demo_circle = shim/demo_circle.cpp, its own process and context), wall per stage as the driver sees it (flatten +
    # upload + kernels + download), and every stage replayed on the CPU port from the state the GPU left before it (oracle/pipeline_chain.py): time + parity.
    # Retriangulate and the retriangulate-only figures below answer VERDICT r3 #9 (it dominates the sequence).
    exe = os.path.join(ROOT, "spherical_sfm_amd", "demo_circle")
    if os.path.exists(exe):
        import subprocess
        import tempfile
        from oracle import pipeline_chain as PC
        with tempfile.TemporaryDirectory() as td:
            dump = os.path.join(td, "dump.bin")
            t = time.perf_counter()
            pr = subprocess.run([exe, "170000", dump, "500", "6", "7", "1", "0"], capture_output=True, text=True, timeout=600)
            wall = time.perf_counter() - t
            if pr.returncode == 0:
                ms = [float(x) for x in [l for l in pr.stdout.splitlines() if l.startswith("STAGE_MS")][0].split()[1:]]
                d = PC.read_dump(dump)
                names = {0: "Optimize", 1: "Retriangulate", 2: "Normalize", 3: "unfix translations"}
                stages = []; general = False
                for k, st in enumerate(d["stages"]):
                    if st["kind"] == PC.UNFIX:
                        general = True; continue
                    t = time.perf_counter(); want, info = PC.run_stage(O, d, d["states"][k], st["kind"], general, False); tcpu = time.perf_counter() - t
                    diff = PC.compare_states(d["states"][k + 1], want)
                    stages.append({"stage": names[st["kind"]] + (" (general)" if general and st["kind"] == 0 else " (spherical)" if st["kind"] == 0 else ""),
                                   "gpu_ms": ms[k], "cpu_port_ms": 1e3 * tcpu, "lm_iterations": st["iterations"] if st["kind"] == 0 else None,
                                   "iterations_cpu": info.get("iterations") if st["kind"] == 0 else None,
                                   "max_rel_camera": diff["cam"], "max_rel_point": diff["pt_max"], "rel_focal": diff["focal"], "zero_set_difference": diff["zero_diff"]})
                res["pipeline_configs2"] = {
                    "workload": "BASELINE configs[2] size: 500 cameras x 170 000 points x 1 020 000 observations, K = 6, stride 7, shared focal free, spherical then general BA "
                                "(demo_circle through the sphericalsfm::SfM mirror; the image / matching front end is out of scope)",
                    "stages": stages, "gpu_ms_total": sum(x["gpu_ms"] for x in stages), "cpu_port_ms_total": sum(x["cpu_port_ms"] for x in stages),
                    "cpu_cores": 16, "process_wall_s": wall,
                    "note": "gpu_ms = host wall of the shim call (map flatten + plan + upload + kernels + download); cpu_port_ms = the oracle on the same input state, 16 threads; "
                            "both exclude building the synthetic scene"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cameras", type=int, default=300)
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--obs-per-point", type=int, default=6)
    ap.add_argument("--mode", choices=["general", "spherical"], default="general")
    ap.add_argument("--focal-free", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--no-scale-probe", action="store_true", help="skip the configs[4]-sized single-GPU side measurement")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1 only: strong (default) = the BASELINE configs[1] problem sharded N ways; weak = N x (cameras, points) per job")
    ap.add_argument("--comm", choices=["rccl", "host"], default="rccl",
                    help="N > 1: rccl (one GPU per rank, RCCL over xGMI) or host (ranks share the visible GPUs, reductions staged through gloo: "
                         "exercises the sharded path on a 1-GPU box; not a performance configuration)")
    ap.add_argument("--no-side-paths", action="store_true", help="skip the pairwise-RANSAC and rotation-averaging side measurements (N = 1)")
    ap.add_argument("--no-configs4", action="store_true", help="N > 1: skip the configs[4] problem sharded over the N ranks")
    ap.add_argument("--no-collective-probe", action="store_true", help="N > 1: skip the timing probe that re-runs the sharded solve with its reductions switched off")
    ap.add_argument("--no-pairwise", action="store_true", help="N > 1: skip BASELINE configs[3] (exhaustive pairwise RANSAC) sharded over the N ranks")
    ap.add_argument("--pairwise-pairs", type=int, default=1999000, help="N > 1: image pairs of the pairwise_sharded leg (default: the 2000-frame exhaustive circle)")
    ap.add_argument("--detail", default=None, help="where the full detail dictionary goes (default: bench_detail.json next to bench.py)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from spherical_sfm_amd import ba, synth, _lib

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    host_comm = world > 1 and args.comm == "host"
    if host_comm:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_comm:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    stream = torch.cuda.Stream()
    ctx = ba.Context(local_rank, stream=stream.cuda_stream)
    if host_comm:
        def hook(arr, op):
            t = torch.from_numpy(arr)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        ctx.comm_init_host(world, rank, hook)
    elif world > 1:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(ba.Context.unique_id()), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, 0)
        ctx.comm_init(bytes(uid.cpu().tolist()), world, rank)

    spherical = args.mode == "spherical"
    if world > 1 and args.scaling == "weak":
        args.cameras *= world; args.points *= world
    prob = synth.make_circle(args.cameras, args.points, args.obs_per_point, spherical=spherical, focal_fixed=not args.focal_free)
    adj = ba.BundleAdjuster(ctx, prob)

    def one_step():
        adj.reset()
        return adj.run()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        s = one_step()
    barrier()
    t0 = time.perf_counter()
    n_lm = 0; phase = {"t_kernel_linearize_ms": 0.0, "t_kernel_schur_ms": 0.0, "t_kernel_pcg_ms": 0.0, "t_kernel_update_ms": 0.0}
    for _ in range(args.steps):
        s = one_step()
        n_lm += s["num_linearizations"]
        for k in phase:
            phase[k] += s[k]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_comm else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    M = int(s["num_residual_blocks_global"])
    value = M * n_lm / elapsed

    # ---- per-kernel durations: one extra step with every launch bracketed by hipEvents on the solver's stream
    adj.set_profiling(True)
    sp = one_step()
    ktimes = adj.kernel_times()
    adj.set_profiling(False)
    phase = {k: sp[k] for k in phase}; n_lm_prof = sp["num_linearizations"]     # per-phase device times are only recorded with profiling on
    cams_gpu, pts_gpu, f_gpu = [np.copy(a) if hasattr(a, "copy") else a for a in adj.download()]

    # measured device copy bandwidth (the HBM fractions are quoted against it AND against the nominal 8 TB/s): 512 MB float4 copy, beyond the Infinity Cache
    copy_gbs = None
    if rank == 0:
        import ctypes as _C
        g = _C.c_double(0.0)
        if _lib.lib().ssfm_debug_copy_bandwidth(ctx._p, 512 << 20, 10, _C.byref(g)) == 0 and g.value > 0:
            copy_gbs = float(g.value)

    def vs_copy(gbs):
        return (gbs / copy_gbs) if (gbs is not None and copy_gbs) else None

    out = None
    if rank == 0:
        dc = s["camera_dof"]
        # (observation, observation) pairs of the off-diagonal Schur blocks: k (k - 1) / 2 per used point with k observations
        kk = np.bincount(prob.obs_pt, minlength=args.points).astype(np.float64); kk = kk[(kk >= 3)]
        pairs = float((kk * (kk - 1) / 2).sum())
        # structure sizes for the byte model
        nnzb = s["reduced_blocks"]
        per_iter_bytes, schur_bytes = algorithmic_bytes(M, args.points, nnzb, dc, args.focal_free)
        kern = {k: {"launches": v["launches"], "avg_us": 1e3 * v["total_ms"] / max(1, v["launches"])} for k, v in ktimes.items()}
        # largest data-parallel kernel (k_band_chol_v2 is longer but is a sequential dependency chain): the Schur assembly -- k_schur_gram when the points
        # share camera lists (they do in this workload: every point of a camera window has the same K cameras), k_schur_pairs2 (+ k_cam_sums2) otherwise
        gram = kern.get("k_schur_gram", {}).get("launches", 0) > 0
        dom = "k_schur_gram" if gram else "k_schur_pairs2"
        dom_us = kern.get(dom, {}).get("avg_us", float("nan"))
        dom_bytes = (gram_kernel_bytes(M, args.points, nnzb, args.cameras, dc, args.focal_free) if gram else pair_kernel_bytes(M, args.points, nnzb, args.cameras, dc)) / world
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us == dom_us else None
        asm_us = dom_us + (kern.get("k_cam_sums2", {}).get("avg_us", 0.0) if gram else kern.get("k_cam_sums2", {}).get("avg_us", float("nan")))
        if gram:
            ex_flop, useful_flop = gram_kernel_flops(M / world, args.points / world, pairs / world, dc)
            rc = {"bound": "fp64 (matrix + vector instructions, one peak and one pipe on gfx950)", "kernel": dom, "pairs_per_launch": pairs / world, "observations_per_launch": M / world,
                  "flop_source": "modelled (hand count, DESIGN.md 4): executed = the 16x16x4 products of the row tiles in use (+ three 4x4x4 instructions for the 4-row tail at 36 rows) x 6 k-steps per 8 points on the matrix pipe + 384 flop per observation on the VALU; "
                                 "useful = the Gram entries the algorithm asks for (DC x DC x 3 multiply-adds per pair, the upper triangle per observation) + the same VALU work",
                  "achieved": (ex_flop / (dom_us * 1e-6) / 1e12) if dom_us == dom_us else None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": (ex_flop / (dom_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if dom_us == dom_us else None,
                  "useful_achieved": (useful_flop / (dom_us * 1e-6) / 1e12) if dom_us == dom_us else None,
                  "useful_frac": (useful_flop / (dom_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if dom_us == dom_us else None}
        else:
            rc = {"bound": "fp64-vector", "kernel": dom, "pairs_per_launch": pairs / world, "flop_per_pair": pair_kernel_flops(1),
                  "flop_source": "modelled (hand count of the kernel's arithmetic: 2 re-linearisations x 150 + 252 for the block update), not a counter",
                  "achieved": (pair_kernel_flops(pairs / world) / (dom_us * 1e-6) / 1e12) if dom_us == dom_us else None,
                  "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": (pair_kernel_flops(pairs / world) / (dom_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if dom_us == dom_us else None,
                  # by USEFUL work: each observation is re-linearised K - 1 times per iteration; only the 252-flop block update is work the algorithm asks for
                  "useful_flop_per_pair": PAIR_USEFUL_FLOP,
                  "useful_achieved": (pairs / world * PAIR_USEFUL_FLOP / (dom_us * 1e-6) / 1e12) if dom_us == dom_us else None,
                  "useful_frac": (pairs / world * PAIR_USEFUL_FLOP / (dom_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if dom_us == dom_us else None}
        iter_ms = 1e3 * elapsed / max(1, n_lm)          # wall time of the timed loop per LM iteration (host round trips included)
        traffic = None; valu = None; mops = None; pmc_meta = None
        tp = os.path.join(ROOT, PMC_PROFILE)
        if os.path.exists(tp):
            try:
                pm = json.load(open(tp)); traffic = pm.get(dom); valu = pm.get("_valu_wave_instructions", {}).get(dom); mops = pm.get("_mfma_mops_f64", {}).get(dom)
                # which run the quoted counters come from: the tag of the file and what its collector recorded about the tree it profiled (a stale file shows here)
                meta = pm.get("_meta") if isinstance(pm.get("_meta"), dict) else {}
                pmc_meta = {"file": PMC_PROFILE, "tag": os.path.basename(PMC_PROFILE).split("_")[0], "recorded": meta,
                            "kernel_source_sha16_now": kernel_source_digest(), "stale": meta.get("kernel_source_sha16") != kernel_source_digest()}
            except Exception:
                traffic = None
        out = {
            "metric": "BA obs/sec (Jac+Schur+PCG)", "value": value, "unit": "obs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.cameras} cams x {args.points} pts x {M} obs synthetic circle (BASELINE configs[1]" + (f" x {world}" if world > 1 and args.scaling == "weak" else "") + "), "
                                   f"{args.mode} BA, focal {'free' if args.focal_free else 'fixed'}, CauchyLoss(1.0), Ceres-default LM",
                       "reduced_solver": "exact block-banded Cholesky in Cuthill-McKee order (direct, like the reference's SPARSE_SCHUR); the PCG of the metric's name is a "
                                         f"refinement that did not run: {s.get('pcg_iterations_total', 0)} sweeps in the last step",
                       "reduced_solver_short": "band-cholesky (direct)", "pcg_sweeps": int(s.get("pcg_iterations_total", 0)),
                       "camera_dof": dc, "lm_iterations_per_step": n_lm / args.steps, "sharding": f"points/{world}", "comm": ("none" if world == 1 else ("host-staged gloo (test configuration)" if host_comm else "rccl"))},
            # The dominant kernel against the roof that bounds it.  k_schur_gram is a matrix-core kernel: ALGORITHMIC flop per launch (what the Schur assembly asks for
            # whatever the implementation: per observation one linearisation 150 + the half product 102 + the camera-side sums 132 + the upper triangle of its
            # diagonal Gram block 126; per observation pair the DC x DC x 3 multiply-adds of its off-diagonal block 216 -- DESIGN.md 4) / measured launch duration,
            # against the dense FP64 MFMA peak (which on gfx950 is also the FP64 vector peak, and one pipe).  The pair kernel (ungrouped problems) stays on the HBM line.
            "roofline": ({"bound": "mfma", "kernel": dom, "achieved": rc["useful_achieved"], "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rc["useful_frac"],
                          "traffic": traffic, "dtype_peak": "FP64 dense MFMA 78.6 TFLOP/s = the FP64 vector peak (256 CUs x 4 SIMDs x 16 FMA lanes x 2 x 2.4 GHz; a v_mfma_f64_16x16x4 issues every 64 cycles: measured, profiles/r03_notes.md)", "algorithmic_flop_per_launch": useful_flop,
                          "executed_flop_per_launch": ex_flop, "executed_frac": rc["frac"], "avg_launch_us": dom_us,
                          # cross-check of the model's matrix-pipe part against a counter: SQ_INSTS_VALU_MFMA_MOPS_F64 (512 flop per MOP) of the committed PMC profile
                          "executed_mfma_flop_model": ex_flop - M / world * GRAM_OBS_VALU_FLOP, "executed_mfma_flop_pmc": (mops * 512.0) if mops else None,
                          "traffic_source": f"committed profile {PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command, "
                                            "gfx950 x2 FETCH_SIZE correction applied; HBM bytes per launch) -- NOT measured in this run",
                          "traffic_breakdown": "fetched 22.1 MB = the algorithmic reads (observations 9.6 + point records 9.6 + points 2.4 MB: read once); written 12.3 MB of which the "
                                               "algorithmic part is the 0.6 MB of S -- the rest is what WRITE_SIZE counts for 0.97 M fp64 atomic adds of the tasks' blocks (8 B and more per "
                                               "atomic; profiles/r05z_pmc_per_kernel_avg.json), not re-reads"} if gram else
                         {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                          "traffic_source": f"committed profile {PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command, "
                                            "gfx950 x2 FETCH_SIZE correction applied) -- NOT measured in this run",
                          "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": dom_us}),
            # the same kernel on the HBM line (SURVEY 8d's figure of merit): algorithmic bytes per launch / duration
            "pmc_profile": pmc_meta,
            "scaling_model": scaling_model(world, n_lm / max(1, args.steps)) if world > 1 else None,
            "hbm_copy_GBs": copy_gbs,       # measured in this run: float4 device copy of 512 MB (read + write bytes per second)
            "roofline_hbm": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                             "frac_of_measured_copy": vs_copy(achieved),
                             "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": dom_us},
            "roofline_compute": rc,
            # the per-kernel byte counts below are each kernel's OWN minimum traffic (its inputs and outputs once); three of them read the observations again, so they
            # do not add up to SURVEY 8d's B_alg -- roofline_passes does: its two passes split B_alg = 72 B/obs + 240 B/pt exactly
            "roofline_per_kernel_note": "algorithmic_bytes_per_launch = per-kernel minimum (the kernel's own inputs and outputs once); the kernels of one LM iteration re-read the "
                                        "observations, so these do NOT sum to B_alg; see roofline_passes for a split that does",
            "roofline_passes": pass_rooflines(kern, M, args.points, world, gram, copy_gbs),
            "roofline_per_kernel": with_copy_fraction(kernel_rooflines(kern, M, args.points, nnzb, args.cameras, dc, pairs, n_lm_prof, world, args.focal_free), copy_gbs,
                                                      rocprof_avg_us() if (world == 1 and not spherical and not args.focal_free and args.cameras == 300 and args.points == 100000) else None),
            # the dominant kernel is bound by instruction issue, not by HBM: VALU wave-instructions per launch (PMC SQ_INSTS_VALU of the
            # committed profile) against what 256 CUs x 4 SIMDs can issue in the measured launch time (one wave64 VALU op per SIMD per 4 cycles)
            "valu_issue": {"kernel": dom, "wave_instructions_per_launch": valu, "clock_ghz": 2.4, "source": f"committed profile {PMC_PROFILE} (SQ_INSTS_VALU), not measured in this run",
                           "frac": (valu / (256 * dom_us * 1e-6 * 2.4e9) if valu and dom_us == dom_us else None)},
            "roofline_schur_assembly": {"bound": "hbm", "kernels": [dom] if gram else ["k_cam_sums2", dom], "algorithmic_bytes": schur_bytes / world,
                                        "avg_us": asm_us, "achieved": (schur_bytes / world) / (asm_us * 1e-6) / 1e9 if asm_us == asm_us else None,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": (schur_bytes / world) / (asm_us * 1e-6) / 1e9 / HBM_PEAK_GBS if asm_us == asm_us else None},
            "reduced_factorisation": {"name": "k_band_chol_v2", "launches_per_lm_iteration": kern.get("k_band_chol_v2", {}).get("launches", 0) / max(1, n_lm_prof),
                                      "avg_us": kern.get("k_band_chol_v2", {}).get("avg_us"),
                                      "note": "block-banded Cholesky of the reduced camera system (ring halves, then separators): chains of dependent steps, latency bound; no roofline applies"},
            "roofline_lm_iteration": {"bound": "hbm", "algorithmic_bytes": per_iter_bytes, "avg_ms": iter_ms,
                                      "achieved": per_iter_bytes / (iter_ms * 1e-3) / 1e9 if iter_ms > 0 else None,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": per_iter_bytes / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if iter_ms > 0 else None,
                                      "frac_of_measured_copy": vs_copy(per_iter_bytes / (iter_ms * 1e-3) / 1e9) if iter_ms > 0 else None},
            "kernels": kern, "lm_iterations_profiled_step": n_lm_prof,
            "phases_ms_per_lm_iteration_profiled_step": {k: v / max(1, n_lm_prof) for k, v in phase.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O   # checker + timed CPU baseline ("port"): a proxy for the Ceres path
            best = None
            for _ in range(args.cpu_reps):
                oc, op, of, os_ = O.ba_solve(prob)
                if best is None or os_["t_total_s"] < best["t_total_s"]:
                    best = os_
            cpu_obs_s = best["num_residual_blocks"] * best["num_linear_solves"] / (best["t_total_s"] - best["t_flatten_s"])
            err_c = float(np.abs(cams_gpu - oc).max() / np.abs(oc).max())
            err_p = float((np.linalg.norm(pts_gpu - op, axis=1) / np.linalg.norm(op, axis=1)).max())
            out["cpu_baseline"] = {"value": cpu_obs_s, "unit": "obs/s", "cores": best["threads_used"], "kind": "port",
                                   "sample": f"the full step workload (one Optimize-equivalent solve, {best['num_linear_solves']} LM iterations), "
                                             f"best of {args.cpu_reps}; LM loop only, flatten excluded on both sides",
                                   "host_cpus": os.cpu_count(), "solve_s": best["t_total_s"] - best["t_flatten_s"],
                                   "end_to_end_s": best["t_total_s"]}
            out["parity_vs_oracle"] = {"max_rel_camera": err_c, "max_rel_point": err_p, "iterations_gpu": sp["iterations"],
                                       "iterations_cpu": best["iterations"]}
            out["speedup_vs_cpu_lm_loop"] = value / cpu_obs_s
            # end to end, what a drop-in SfM::Optimize() call costs: ssfm_ba_solve = host planning + allocation/H2D + LM + D2H,
            # against the CPU port's flatten + solve (the reference additionally probes Np x Nc map entries, src/sfm.cpp:240-263)
            def e2e_call(p_=None):
                _, _, _, se = ba.optimize(ctx, prob if p_ is None else p_)
                return {"gpu_s": se["t_flatten_s"] + se["t_upload_s"] + se["t_solve_s"] + se["t_download_s"], "gpu_plan_s": se["t_flatten_s"],
                        "gpu_alloc_upload_s": se["t_upload_s"], "gpu_solve_s": se["t_solve_s"], "gpu_download_s": se["t_download_s"]}
            # cold = first call on a structure (plans, allocates, uploads the index lists): best of three such calls -- the problem itself and two copies with one
            # point held fixed (a different structure hash, the same work); a single sample right behind the 16-thread CPU baseline read 7 ... 17 ms
            import dataclasses
            colds = [e2e_call()]
            for k_ in (5, 6):
                pf_ = prob.pt_fixed.copy(); pf_[k_] = 1
                colds.append(e2e_call(dataclasses.replace(prob, pt_fixed=pf_)))
            srt = sorted(colds, key=lambda d: d["gpu_s"])
            e2e = dict(srt[len(srt) // 2]); e2e["cold_samples_s"] = [c_["gpu_s"] for c_ in colds]      # the MEDIAN sample is the quoted one (gpu_s and its parts)
            e2e["first_s"] = colds[0]["gpu_s"]; e2e["median_s"] = srt[len(srt) // 2]["gpu_s"]; e2e["best_s"] = srt[0]["gpu_s"]
            e2e_call()                                              # back on the original structure (cold once more), then:
            warm = min((e2e_call() for _ in range(3)), key=lambda d: d["gpu_s"])   # same structure again (the drivers' pattern): plan cache hit
            e2e["warm"] = warm
            e2e["cpu_s"] = best["t_total_s"]; e2e["speedup"] = best["t_total_s"] / e2e["gpu_s"]
            nres, t_ref_flat = O.reference_style_flatten(prob)
            e2e["cpu_reference_style_build_loop_s"] = t_ref_flat             # what SfM::Optimize() spends before Ceres starts
            e2e["speedup_incl_reference_build_loop"] = (best["t_total_s"] - best["t_flatten_s"] + t_ref_flat) / e2e["gpu_s"]
            e2e["speedup_warm"] = best["t_total_s"] / warm["gpu_s"]
            out["end_to_end_optimize"] = e2e
        if world == 1 and not args.no_side_paths and not args.no_cpu_baseline:
            out["side_paths"] = side_paths(ctx)
        if world == 1 and not args.no_scale_probe and not spherical and not args.focal_free and args.cameras == 300:
            # side measurement, not the headline: the BASELINE configs[4] SIZE (4000 cameras / 1.5 M points / 12 M observations, two
            # rings of 2000 cameras) on this one GPU -- the long components go through the substructured factorisation
            adj.close()
            big = synth.make_circle(4000, 1500000, 8, spherical=False, focal_fixed=True)
            adj = ba.BundleAdjuster(ctx, big)
            adj.run(); torch.cuda.synchronize()
            tb = time.perf_counter(); nb = 0
            for _ in range(2):
                adj.reset(); sb = adj.run(); nb += sb["num_linearizations"]
            torch.cuda.synchronize(); tb = time.perf_counter() - tb
            out["scale_probe_configs4_size_one_gpu"] = {
                "workload": "4000 cams x 1500000 pts x 12000000 obs, general BA, focal fixed", "value": sb["num_residual_blocks"] * nb / tb, "unit": "obs/s",
                "ms_per_solve": 1e3 * tb / 2, "lm_iterations": sb["num_linearizations"], "band_half_width": sb["band_half_width"],
                "factorisation_workgroups": sb["band_segments"], "separators": sb["band_separators"]}
            # "treat config 5 as the real HBM test" -- the working set (1.2 GB touched per iteration) does not fit the 256 MB Infinity Cache.
            # One more solve with every launch bracketed by hipEvents on the solver's stream: per-kernel HBM fractions and the pair kernel by useful flops.
            adj.set_profiling(True); adj.reset(); sq = adj.run(); kb = adj.kernel_times(); adj.set_profiling(False)
            kernb = {k: {"launches": v["launches"], "avg_us": 1e3 * v["total_ms"] / max(1, v["launches"])} for k, v in kb.items()}
            Mb = int(sq["num_residual_blocks"]); nlb = sq["num_linearizations"]
            kkb = np.bincount(big.obs_pt, minlength=1500000).astype(np.float64); kkb = kkb[kkb >= 3]
            per_iter_b, _ = algorithmic_bytes(Mb, 1500000, sq["reduced_blocks"], sq["camera_dof"], False)
            iter_ms_b = 1e3 * tb / max(1, nb)
            gpu_ms = sum(v["total_ms"] for v in kb.values()) / max(1, nlb)
            out["roofline_configs4"] = {
                "workload": "BASELINE configs[4] SIZE on one GPU: 4000 cams x 1500000 pts x 12000000 obs (8 observations per point), general BA, focal fixed",
                "lm_iteration": {"algorithmic_bytes": per_iter_b, "avg_ms_wall": iter_ms_b, "achieved_GBs": per_iter_b / (iter_ms_b * 1e-3) / 1e9,
                                 "frac_hbm": per_iter_b / (iter_ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_of_measured_copy": vs_copy(per_iter_b / (iter_ms_b * 1e-3) / 1e9),
                                 "kernel_ms_per_iteration_profiled": gpu_ms},
                "kernels": with_copy_fraction(kernel_rooflines(kernb, Mb, 1500000, sq["reduced_blocks"], 4000, sq["camera_dof"], float((kkb * (kkb - 1) / 2).sum()), nlb), copy_gbs),
                "all_kernels_avg_us": {k: v["avg_us"] for k, v in kernb.items()},
                "share_of_gpu_time": {k: v["total_ms"] / max(1e-12, sum(w["total_ms"] for w in kb.values())) for k, v in kb.items()}}
    if world > 1 and not args.no_collective_probe:
        # What do the collectives cost?  The same sharded solve with the reductions switched off (ssfm_debug_timing_skip_collectives: every rank iterates on its own
        # shard, results meaningless, kernels and sizes unchanged); per LM iteration, rank 0's count.  Reported next to the real figure, never as `value`.
        _lib.lib().ssfm_debug_timing_skip_collectives(ctx._p, 1)
        adj.reset(); adj.run(); barrier()
        tq = time.perf_counter(); nq = 0
        for _ in range(max(2, args.steps // 2)):
            adj.reset(); sq = adj.run(); nq += sq["num_linearizations"]
        torch.cuda.synchronize(); tq = time.perf_counter() - tq
        _lib.lib().ssfm_debug_timing_skip_collectives(ctx._p, 0)
        barrier()
        if rank == 0:
            with_ms = 1e3 * elapsed / max(1, n_lm); without_ms = 1e3 * tq / max(1, nq)
            out["timing_without_collective"] = {"ms_per_lm_iteration": without_ms, "ms_per_lm_iteration_with_collectives": with_ms,
                                                "collective_cost_ms_per_lm_iteration_estimate": with_ms - without_ms, "lm_iterations_timed_rank0": nq,
                                                "note": "reductions skipped: every rank solved its own shard; a timing probe, its results are not a solution"}
    if world > 1 and not args.no_configs4 and not spherical and not args.focal_free and args.cameras == 300 and args.scaling == "strong":
        # BASELINE configs[4] (the 8-GPU config): 4000 cameras / 1.5 M points / 12 M observations sharded over the N ranks of this job
        adj.close()
        big = synth.make_circle(4000, 1500000, 8, spherical=False, focal_fixed=True)
        adj = ba.BundleAdjuster(ctx, big)
        adj.run(); barrier()
        tb = time.perf_counter(); nb = 0
        for _ in range(2):
            adj.reset(); sb = adj.run(); nb += sb["num_linearizations"]
        barrier(); tb = time.perf_counter() - tb
        t = torch.tensor([tb], dtype=torch.float64, device="cpu" if host_comm else "cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); tb = float(t.item())
        if rank == 0:
            out["configs4_sharded"] = {
                "workload": "BASELINE configs[4]: 4000 cams x 1500000 pts x 12000000 obs, general BA, focal fixed, points sharded over the ranks, one all-reduce per LM iteration",
                "n_gpus": world, "scaling": "strong", "value": sb["num_residual_blocks_global"] * nb / tb, "unit": "obs/s", "ms_per_solve": 1e3 * tb / 2,
                "lm_iterations": sb["num_linearizations"], "observations_this_rank": sb["num_residual_blocks"], "band_half_width": sb["band_half_width"],
                "factorisation_workgroups": sb["band_segments"], "separators": sb["band_separators"]}
    if world > 1 and not args.no_pairwise:
        # BASELINE configs[3] ("2000-frame circle, exhaustive pairwise spherical_estimator RANSAC batched across 4 x MI355X"): the 1 999 000 image
        # pairs x 500 correspondences over the N ranks of this job through ssfm_ransac_batch_indexed_sharded -- every rank passes the same pair list,
        # estimates pairs r, r + N, ... and one all-reduce per call of 100 000 pairs hands every rank every result (no other exchange).
        from spherical_sfm_amd import ransac
        Fp = 1000.0; NCp = 500; POOL = 2000; TOTAL = args.pairwise_pairs; PER = 100000
        pool = [synth.make_relative_pose_problem(NCp, seed=1000 + k, noise=1 / Fp, outlier_frac=0.3, rotation_deg=1 + (k % 60)) for k in range(POOL)]
        feat_ptr = (np.arange(POOL + 1, dtype=np.int64) * 2 * NCp).astype(np.int32)
        feat_rays = np.ascontiguousarray(np.concatenate([np.concatenate([q[0], q[1]]) for q in pool]))
        pptr = (np.arange(PER + 1, dtype=np.int64) * NCp).astype(np.int32); pm0 = np.tile(np.arange(NCp, dtype=np.int32), PER); pm1 = pm0 + NCp
        def pw_call(first, n):
            fr = ((first + np.arange(n)) % POOL).astype(np.int32)
            return ransac.estimate_indexed(ctx, feat_ptr, feat_rays, fr, fr, pptr[:n + 1], pm0[:n * NCp], pm1[:n * NCp], (2 / Fp) ** 2, sharded=True, min_num_inliers=20)
        pw_call(0, 2000 * world)                                                  # warm-up: module load, pinned staging buffers, the communicator
        barrier(); tp = time.perf_counter(); done = 0; acc = 0
        while done < TOTAL:
            n = min(PER, TOTAL - done); o = pw_call(done, n); acc += int((o["num_inliers"] > 20).sum()); done += n
        barrier(); tp = time.perf_counter() - tp
        t = torch.tensor([tp], dtype=torch.float64, device="cpu" if host_comm else "cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); tp = float(t.item())
        if rank == 0:
            out["pairwise_sharded"] = {
                "workload": f"BASELINE configs[3]: {TOTAL} image pairs x {NCp} correspondences (30 % outliers, 1 px noise at f = 1000), reference-trace LO-MSAC, "
                            f"ssfm_ransac_batch_indexed_sharded: pairs round robin over the ranks, host buffers in, every result on every rank (one all-reduce per {PER} pairs)",
                "n_gpus": world, "scaling": "strong", "value": TOTAL / tp, "unit": "pairs/s", "seconds": tp, "accepted_pairs": acc, "includes_pcie": True}
    if rank == 0:
        detail = _jsonable(out)
        dpath = args.detail or os.path.join(ROOT, "bench_detail.json" if world == 1 else f"bench_detail_n{world}.json")
        try:
            with open(dpath, "w") as fh:
                json.dump(detail, fh, allow_nan=False)
            detail["detail_path"] = os.path.relpath(dpath, ROOT)
        except OSError as e:
            print(f"bench.py: could not write {dpath}: {e}", file=sys.stderr)
        sys.stdout.flush()
        print(record_line(detail), flush=True)
    adj.close(); ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
