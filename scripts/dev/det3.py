import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spherical_sfm_amd import ba, synth
mode = sys.argv[1]
ctx = ba.Context(0)
p = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
if mode == "after0":
    os.environ["SSFM_DETERMINISTIC"] = "0"
    for r in range(3): ba.optimize(ctx, p)
os.environ["SSFM_DETERMINISTIC"] = "1"
for cap in (5, 8, 14, 50):
    res = []
    for r in range(6):
        c, x, f, s = ba.optimize(ctx, p, max_num_iterations=cap)
        res.append((c.copy(), x.copy(), s["final_cost"], s["iterations"]))
    print(f"{mode} cap {cap}: its {res[0][3]} cams equal {[bool(np.array_equal(q[0], res[0][0])) for q in res[1:]]} cost equal {[q[2] == res[0][2] for q in res[1:]]} max cam diff {max(np.abs(q[0]-res[0][0]).max() for q in res[1:]):.2e}", flush=True)
