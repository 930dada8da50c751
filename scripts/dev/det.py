"""SSFM_DETERMINISTIC=1 (csrc/det_acc.h): repeated solves bit for bit?  python scripts/dev/det.py [reps]   (the env var is read per handle)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spherical_sfm_amd import ba, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = ba.Context(0)
cases = [("config2", synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)),
         ("config2 focal free", synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=False)),
         ("spherical", synth.make_circle(300, 100000, 6, spherical=True, focal_fixed=False)),
         ("ragged 3..8 (pair lists)", synth.make_ragged_circle(300, 600000, 3, 8)),
         ("ragged 3..14 focal free", synth.make_ragged_circle(300, 600000, 3, 14, focal_fixed=False))]
for name, p in cases:
    out = {}
    for det in ("0", "1"):
        os.environ["SSFM_DETERMINISTIC"] = det
        res = []; best = 1e9
        for r in range(reps):
            t = time.time(); c, x, f, s = ba.optimize(ctx, p); best = min(best, time.time() - t)
            res.append((c.copy(), x.copy(), f, s["final_cost"], s["iterations"]))
        same = sum(1 for q in res[1:] if np.array_equal(q[0], res[0][0]) and np.array_equal(q[1], res[0][1]) and np.array_equal(np.asarray(q[2]), np.asarray(res[0][2]), equal_nan=True) and q[3] == res[0][3])
        if det == "1" and same < reps - 1:
            for q in res[1:]: print("      differs:", "cams" if not np.array_equal(q[0], res[0][0]) else "", "pts" if not np.array_equal(q[1], res[0][1]) else "", "f" if not np.array_equal(np.asarray(q[2]), np.asarray(res[0][2]), equal_nan=True) else "", "cost" if q[3] != res[0][3] else "", "its", q[4])
        adj = ba.BundleAdjuster(ctx, p); adj.run(); tb = 1e9
        for _ in range(5): adj.reset(); t = time.time(); s2 = adj.run(); tb = min(tb, time.time() - t)
        adj.close()
        out[det] = res[0]
        print(f"[{name}] SSFM_DETERMINISTIC={det}: {same} of {reps - 1} repeats identical to the first (cameras, points, focal, final cost); iterations {res[0][4]}; "
              f"resident solve {1e3 * tb:.3f} ms = {1e6 * tb / max(s2['num_linearizations'], 1):.1f} us / iteration", flush=True)
    a, b = out["0"], out["1"]
    print(f"    deterministic against default: cameras {np.abs(a[0] - b[0]).max() / np.abs(a[0]).max():.2e}, points {np.abs(a[1] - b[1]).max() / np.abs(a[1]).max():.2e}, cost {abs(a[3] - b[3]) / a[3]:.2e}, iterations {a[4]} / {b[4]}", flush=True)
