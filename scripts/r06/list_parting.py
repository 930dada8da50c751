"""which runs with in-loop local optimisation leave the reference RansacLib's (tests/test_reference_pins_gpu.py asserts this list): python scripts/r06/list_parting.py (GPU box)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_reference_pins_gpu as T
from spherical_sfm_amd import ba, ransac
ctx = ba.Context(0)
g = np.load(os.path.join(T.GOLD, "ref_ransaclib.npz")); m = T._fixture_module()
ptr = g["pair_ptr"]; bad = []
for k in range(len(g["pair_seed"])):
    kw = dict(m.PAIR_CASES[g["pair_case"][k]][4])
    dev = {T._KW.get(a, a): (int(b) if isinstance(b, (bool, np.bool_)) else b) for a, b in kw.items()}
    dev.setdefault("final_least_squares", 1); dev.setdefault("num_lo_steps", 0); dev.setdefault("num_lsq_iterations", 0)
    u = g["pair_u"][ptr[k]:ptr[k + 1]]; v = g["pair_v"][ptr[k]:ptr[k + 1]]
    out = ransac.estimate_pairs(ctx, [(u, v)], T.THR, seed=int(g["pair_seed"][k]), min_num_inliers=0, **dev)
    same = (out["iterations"][0] == g["pair_iterations"][k] and out["lo_runs"][0] == g["pair_lo_runs"][k]
            and out["num_inliers"][0] == g["pair_num_inliers"][k] and np.array_equal(out["inliers"][0], g["pair_mask"][ptr[k]:ptr[k + 1]]))
    close = (not same) or g["pair_num_inliers"][k] == 0 or len(u) <= 5 or T._sign_dist(out["E"][0], g["pair_E"][k]) <= 1e-8
    if dev["num_lo_steps"] > 0 and not (same and close):
        bad.append((k, int(g["pair_case"][k]), len(u), int(out["iterations"][0]), int(g["pair_iterations"][k]), int(out["num_inliers"][0]), int(g["pair_num_inliers"][k]), bool(same)))
print("PARTING", [b[0] for b in bad]); 
for b in bad: print(b)
print("n lo cases", sum(1 for k in range(len(g["pair_seed"])) if dict(m.PAIR_CASES[g["pair_case"][k]][4]).get("num_lo_steps", 0) > 0))
