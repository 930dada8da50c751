"""Irregular-structure study: 300 cameras, ~600k observations, ragged tracks of 3..14 consecutive frames with point ids in build_sfm order
(spherical_sfm_amd/synth.py: make_ragged_circle).  Prints per-kernel times of one solve for a few planner settings; run on the GPU box.
  python scripts/prof_irregular.py [max_len=14] [check=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherical_sfm_amd import ba, synth  # noqa: E402

max_len = int(sys.argv[1]) if len(sys.argv) > 1 else 14
check = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
ctx = ba.Context(0)
prob = synth.make_ragged_circle(300, 600000, 3, max_len)
M = len(prob.obs_cam)
ref = None
for sort, min_run in [("0", 32), ("1", 32), ("1", 16), ("1", 8)]:
    os.environ["SSFM_GRAM_SORT"] = sort; os.environ["SSFM_GRAM_MIN_RUN"] = str(min_run)
    info, _, _, _ = ba.plan(prob)
    adj = ba.BundleAdjuster(ctx, prob)
    adj.reset(); s = adj.run()
    t = time.perf_counter(); n = 0
    for _ in range(3):
        adj.reset(); s = adj.run(); n += s["num_linearizations"]
    dt = time.perf_counter() - t
    adj.set_profiling(True); adj.reset(); sp = adj.run(); kt = adj.kernel_times(); adj.set_profiling(False)
    cams, pts, f = [np.copy(a) if hasattr(a, "copy") else a for a in adj.download()]
    if ref is None: ref = (cams, pts)
    print(f"sort={sort} min_run={min_run}: grouped {info['num_points_grouped']}/{info['num_points_used']} points, {info['num_observations_grouped']}/{info['num_observations_used']} obs, "
          f"band {info['band_half_width']}, segments {info['band_segments']}, separators {info['band_separators']}; iterations {s['iterations']} termination {s['termination']}, "
          f"{1e3 * dt / n:.3f} ms per LM iteration, {M * n / dt:.3e} obs/s; max cam diff vs first setting {np.abs(cams - ref[0]).max():.2e}")
    print("   " + ", ".join(f"{k} {1e3 * v['total_ms'] / max(1, v['launches']):.1f}us x{v['launches'] / max(1, sp['num_linearizations']):.1f}" for k, v in kt.items() if v["launches"]))
    adj.close()
if check:
    from oracle import oracle as O
    t = time.perf_counter(); oc, op, of, os_ = O.ba_solve(prob); tc = time.perf_counter() - t
    print(f"oracle: iterations {os_['iterations']}, {tc:.2f} s; max rel cam {np.abs(ref[0] - oc).max() / np.abs(oc).max():.2e} pt {(np.linalg.norm(ref[1] - op, axis=1) / np.linalg.norm(op, axis=1)).max():.2e}")
