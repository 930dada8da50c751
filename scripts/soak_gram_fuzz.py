"""Soak of the signature-group path: random problems with runs of every length class, loose points, constant points / cameras, all losses -- the solve with groups
(k_schur_gram, k_gram_backsub forced on / off at random) against the same solve with SSFM_GRAM=0.  usage: python scripts/soak_gram_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from spherical_sfm_amd import ba
import test_gram_groups_gpu as G

os.environ["SSFM_NO_PLAN_CACHE"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = ba.Context(0)
rng = np.random.default_rng(2026)
worst = 0.0; bad = 0
for case in range(n):
    spherical = bool(rng.integers(0, 2)); focal_fixed = bool(rng.integers(0, 2)); loss = int(rng.integers(0, 3))
    if rng.random() < 0.5:
        p = G.multi_k_problem(1000 + case, spherical, focal_fixed, block=int(rng.choice([32, 40, 48, 100])))
    else:
        p = G.mixed_problem(1000 + case, int(rng.integers(3, 9)), spherical, focal_fixed, loose_frac=float(rng.choice([0.0, 0.2, 0.5])), per_cam=int(rng.choice([33, 45, 70])))
    os.environ["SSFM_GRAM_KMIN"] = str(int(rng.integers(2, 5))); os.environ["SSFM_GRAM_BACKSUB"] = str(int(rng.integers(0, 2))); os.environ["SSFM_GRAM_PTS"] = str(int(rng.choice([8, 24, 64, 96, 192])))
    os.environ.pop("SSFM_GRAM", None)
    c1, x1, f1, s1 = ba.optimize(ctx, p, loss_type=loss, loss_scale=1.5)
    os.environ["SSFM_GRAM"] = "0"
    c0, x0, f0, s0 = ba.optimize(ctx, p, loss_type=loss, loss_scale=1.5)
    e = max(np.abs(c1 - c0).max() / np.abs(c0).max(), np.abs(x1 - x0).max() / np.abs(x0).max(), abs(f1 - f0) / f0)
    ok = s1["iterations"] == s0["iterations"] and s1["termination"] == s0["termination"] and e <= 1e-7
    worst = max(worst, e); bad += 0 if ok else 1
    if not ok:
        print(f"case {case}: MISMATCH iterations {s1['iterations']} / {s0['iterations']} rel {e:.2e} spherical {spherical} focal_fixed {focal_fixed} loss {loss} env "
              f"{os.environ['SSFM_GRAM_KMIN']} {os.environ['SSFM_GRAM_BACKSUB']} {os.environ['SSFM_GRAM_PTS']}", flush=True)
print(f"{n} cases, {bad} mismatches, worst relative difference {worst:.2e}")
ctx.close()
