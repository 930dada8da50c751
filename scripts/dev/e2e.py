"""End-to-end Optimize() timing on the GPU box: ssfm_ba_solve = flatten + upload + LM + download, per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa
from spherical_sfm_amd import ba, synth
p = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
ctx = ba.Context(0)
for it in range(4):
    t = time.time(); cams, pts, f, s = ba.optimize(ctx, p); dt = time.time() - t
    print("ssfm_ba_solve wall %.1f ms | flatten %.1f upload %.1f solve %.1f download %.1f ms | its %d" % (
        dt * 1e3, s["t_flatten_s"] * 1e3, s["t_upload_s"] * 1e3, s["t_solve_s"] * 1e3, s["t_download_s"] * 1e3, s["iterations"]))
