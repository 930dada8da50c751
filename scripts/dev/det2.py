import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spherical_sfm_amd import ba, synth
os.environ["SSFM_DETERMINISTIC"] = "1"
ctx = ba.Context(0)
p = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
for cap in (1, 2, 3, 5):
    res = []
    for r in range(6):
        c, x, f, s = ba.optimize(ctx, p, max_num_iterations=cap)
        res.append((c.copy(), x.copy(), s["final_cost"]))
    print(f"cap {cap}: cams equal {[bool(np.array_equal(q[0], res[0][0])) for q in res[1:]]} pts equal {[bool(np.array_equal(q[1], res[0][1])) for q in res[1:]]} cost equal {[q[2] == res[0][2] for q in res[1:]]} max cam diff {max(np.abs(q[0]-res[0][0]).max() for q in res[1:]):.2e}", flush=True)
# the same handle, reset + run
adj = ba.BundleAdjuster(ctx, p)
outs = []
for r in range(6):
    adj.reset(); s = adj.run(); c, x, f = adj.download(); outs.append((c.copy(), x.copy(), s["final_cost"]))
print("resident handle:", [bool(np.array_equal(q[0], outs[0][0]) and np.array_equal(q[1], outs[0][1])) for q in outs[1:]], [q[2] == outs[0][2] for q in outs[1:]])
