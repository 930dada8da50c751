import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from spherical_sfm_amd import ba, synth
ctx = ba.Context(0)
p = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
for env in ("0", "1"):
    os.environ["SSFM_DETERMINISTIC"] = env
    c, pts, f, s = ba.optimize(ctx, p, verbose=0)
    print("det", env, "iterations", s["iterations"], "termination", s["termination"], "cost", s["initial_cost"], s["final_cost"], "lin", s["num_linearizations"])
