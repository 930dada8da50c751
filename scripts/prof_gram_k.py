"""Schur assembly by observations per point: k_schur_gram against k_schur_pairs2 + k_cam_sums2.  usage: [SSFM_GRAM=0] python scripts/prof_gram_k.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spherical_sfm_amd import ba, synth
ctx = ba.Context(0)
for nc, npts in [(300, 100000), (4000, 1500000)]:
    for K in (3, 4, 5, 6, 7):
        p = synth.make_circle(nc, npts, K, spherical=False, focal_fixed=True, check_in_frame=False)
        adj = ba.BundleAdjuster(ctx, p); adj.run(); adj.reset(); adj.set_profiling(True); s = adj.run(); kt = adj.kernel_times(); adj.close()
        show = {k: round(1e3 * v["total_ms"] / max(1, v["launches"]), 1) for k, v in kt.items() if k in ("k_schur_pairs2", "k_cam_sums2", "k_schur_gram")}
        print(f"cams {nc} pts {npts} K {K}: assembly {sum(show.values()):.1f} us {show} solve {1e3 * s['t_solve_s']:.2f} ms / {s['iterations']} iterations", flush=True)
ctx.close()
