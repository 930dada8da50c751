"""debug: which pairs of the reference-trace LO-MSAC differ from the oracle, and where (tests/test_ransac_trace_gpu.py thresholds)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.spatial.transform import Rotation
from spherical_sfm_amd import ba, synth, ransac
from oracle import oracle as O
THR = (2 / 600) ** 2
def frob_err(E, Es):
    a = E / np.linalg.norm(E); b = Es / np.linalg.norm(Es); return min(np.linalg.norm(a - b), np.linalg.norm(a + b))
def pairs(n_pairs, n_corr, outlier_frac, noise, seed0=100):
    return [synth.make_relative_pose_problem(n_corr, seed=seed0 + k, noise=noise, outlier_frac=outlier_frac, rotation_deg=5 + (k % 30)) for k in range(n_pairs)]
ctx = ba.Context(0)
for name, probs, kw, okw in [
    ("estimate_pairwise options", pairs(192, 150, 0.3, 1 / 600) + pairs(64, 500, 0.45, 1 / 600, seed0=300), dict(min_num_inliers=20), dict(min_num_inliers=20)),
    ("LO steps 10 / lsq 4", pairs(96, 200, 0.35, 1 / 600, seed0=500), dict(num_lo_steps=10, num_lsq_iterations=4, final_least_squares=0, min_num_inliers=20),
     dict(num_lo_steps=10, num_lsq_iterations=4, final_least_squares=False, min_num_inliers=20))]:
    out = ransac.estimate_pairs(ctx, [(p[0], p[1]) for p in probs], THR, **kw)
    nbad = 0
    for k, (u, v, R, E, inl) in enumerate(probs):
        o = O.lomsac_pair(u, v, THR, **okw)
        same_trace = out["iterations"][k] == o["iterations"] and out["lo_runs"][k] == o["lo_runs"]
        same_mask = (out["inliers"][k] == o["inliers"]).all()
        fe = frob_err(out["E"][k], o["E"])
        if not (same_trace and same_mask and fe <= 1e-9):
            nbad += 1
            print(f"  pair {k}: it {out['iterations'][k]}/{o['iterations']} lo {out['lo_runs'][k]}/{o['lo_runs']} inl {out['num_inliers'][k]}/{o['num_inliers']} "
                  f"mask diff {(out['inliers'][k] != o['inliers']).sum()} score {out['scores'][k]:.15e}/{o['score']:.15e} E err {fe:.2e}")
    print(name, ": differing pairs", nbad, "of", len(probs))
ctx.close()
