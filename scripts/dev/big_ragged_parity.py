import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spherical_sfm_amd import ba, synth
from oracle import oracle as O
p = synth.make_ragged_circle(1000, 3000000, 3, 8)
oc, op, of, os_ = O.ba_solve(p)
ctx = ba.Context(0)
for env in ({}, {"SSFM_GRAM": "0"}, {"SSFM_GRAM_ANY": "0"}, {"SSFM_RING": "0"}, {"SSFM_DETERMINISTIC": "1"}):
    for k, v in env.items(): os.environ[k] = v
    c, x, f, s = ba.optimize(ctx, p)
    for k in env: del os.environ[k]
    e = np.linalg.norm(x - op, axis=1) / np.linalg.norm(op, axis=1)
    print(env, "its", s["iterations"], os_["iterations"], "cam", np.abs(c - oc).max() / np.abs(oc).max(), "pt max", e.max(), "pt 99.9%", np.quantile(e, 0.999), "median", np.median(e), "cost", abs(s["final_cost"] - os_["final_cost"]) / os_["final_cost"], flush=True)
