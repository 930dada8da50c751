"""k_gram_backsub against k_point_backsub by problem size and observations per point.  usage: [SSFM_GRAM_BACKSUB=0] python scripts/prof_gram_backsub.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spherical_sfm_amd import ba, synth
ctx = ba.Context(0)
for nc, npts, K in [(300, 100000, 6), (300, 100000, 8), (300, 100000, 4), (4000, 1500000, 6), (4000, 1500000, 8)]:
    p = synth.make_circle(nc, npts, K, spherical=False, focal_fixed=True, check_in_frame=False)
    adj = ba.BundleAdjuster(ctx, p); adj.run(); adj.reset(); adj.set_profiling(True); s = adj.run(); kt = adj.kernel_times(); adj.close()
    show = {k: round(1e3 * v["total_ms"] / max(1, v["launches"]), 1) for k, v in kt.items() if k in ("k_point_backsub", "k_gram_backsub", "k_schur_gram", "k_point_lin")}
    print(f"cams {nc} pts {npts} K {K}: iterations {s['iterations']} {show}", flush=True)
ctx.close()
