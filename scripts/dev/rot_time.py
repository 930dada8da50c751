import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from spherical_sfm_amd import ba, synth, rotavg
ctx = ba.Context(0)
for n in (300, 2000):
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8)
    rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel)
    for k in range(3):
        t = time.perf_counter(); Rg, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel); dt = time.perf_counter() - t
        print(n, "call %.3f ms" % (1e3 * dt), {k2: round(1e3 * v, 3) for k2, v in s.items() if k2.startswith("t_") and k2.endswith("_s")}, "iters", s["iterations"], "lin", s["num_linearizations"])
