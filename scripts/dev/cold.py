import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, dataclasses
from spherical_sfm_amd import ba, synth
p = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
ctx = ba.Context(0)
ba.optimize(ctx, p)
for i in range(3):
    tf = p.trans_fixed.copy(); tf[5 + i] = 1                     # a changed mask: new structure
    q = dataclasses.replace(p, trans_fixed=tf)
    t = time.perf_counter(); c, pts, f, s = ba.optimize(ctx, q); dt = time.perf_counter() - t
    print("cold call %d: python %.2f ms ; plan %.2f upload %.2f solve %.2f download %.2f" % (i, 1e3 * dt, 1e3 * s["t_flatten_s"], 1e3 * s["t_upload_s"], 1e3 * s["t_solve_s"], 1e3 * s["t_download_s"]), flush=True)
