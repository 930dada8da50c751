import os, sys, importlib.util
sys.path.insert(0, '/root/repo')
import numpy as np
from spherical_sfm_amd import ba, ransac
from oracle import oracle as O
GOLD = '/root/repo/tests/golden'
spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_reference_fixtures.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
g = np.load(os.path.join(GOLD, "ref_ransaclib.npz")); ptr = g["pair_ptr"]
KW = dict(use_poly="use_poly_solver", max_iterations="max_num_iterations", min_iterations="min_num_iterations")
ctx = ba.Context(0)
for k in range(len(g["pair_seed"])):
    u = g["pair_u"][ptr[k]:ptr[k + 1]]; v = g["pair_v"][ptr[k]:ptr[k + 1]]
    if len(u) > 5: continue
    kw = dict(m.PAIR_CASES[g["pair_case"][k]][4])
    dev = {KW.get(a, a): (int(b) if isinstance(b, (bool, np.bool_)) else b) for a, b in kw.items()}
    dev.setdefault("final_least_squares", 1); dev.setdefault("num_lo_steps", 0); dev.setdefault("num_lsq_iterations", 0)
    out = ransac.estimate_pairs(ctx, [(u, v)], (2 / 1000.0) ** 2 if not hasattr(m, 'THR') else m.THR, seed=int(g["pair_seed"][k]), min_num_inliers=0, **dev)
    mask = g["pair_mask"][ptr[k]:ptr[k + 1]].astype(bool)
    sd = sum(O.sampson(out["E"][0], u[i], v[i]) for i in range(len(u)) if mask[i]); sr = sum(O.sampson(g["pair_E"][k], u[i], v[i]) for i in range(len(u)) if mask[i])
    print(k, len(u), int(mask.sum()), "dev %.3e ref %.3e" % (sd, sr), "dE %.2e" % min(np.abs(out["E"][0] - g["pair_E"][k]).max(), np.abs(out["E"][0] + g["pair_E"][k]).max()))
