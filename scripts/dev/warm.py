import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from spherical_sfm_amd import ba, synth
p = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
ctx = ba.Context(0)
for i in range(4):
    t = time.perf_counter(); c, pts, f, s = ba.optimize(ctx, p); dt = time.perf_counter() - t
    print("call %d: python %.2f ms ; plan %.2f upload %.2f solve %.2f download %.2f" % (i, 1e3 * dt, 1e3 * s["t_flatten_s"], 1e3 * s["t_upload_s"], 1e3 * s["t_solve_s"], 1e3 * s["t_download_s"]), flush=True)
