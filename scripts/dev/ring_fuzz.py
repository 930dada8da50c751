"""Random ring-shaped problems through the planner's fold / ring decision and the ring kernels, against the oracle: python scripts/dev/ring_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spherical_sfm_amd import ba, synth
from oracle import oracle as O
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = ba.Context(0)
bad = 0
for it in range(ncase):
    sph = bool(rng.integers(0, 2)); ff = bool(rng.integers(0, 2)); det = bool(rng.integers(0, 3) == 0)
    if rng.integers(0, 2):
        nc = int(rng.integers(90, 720)); ml = int(rng.integers(3, 13)); nobs = int(rng.integers(40, 120)) * nc
        try: p = synth.make_ragged_circle(nc, nobs, 3, ml, spherical=sph, focal_fixed=ff, seed=int(rng.integers(1, 10**6)))
        except AssertionError: continue
        name = f"ragged nc={nc} obs={nobs} len<={ml}"
    else:
        nc = int(rng.integers(64, 900)); K = int(rng.integers(3, 9)); npts = int(rng.integers(8, 30)) * nc
        p = synth.make_circle(nc, npts, K, spherical=sph, focal_fixed=ff, check_in_frame=False, seed=int(rng.integers(1, 10**6)))
        name = f"circle nc={nc} pts={npts} K={K}"
    os.environ["SSFM_DETERMINISTIC"] = "1" if det else "0"
    info = ba.plan(p)[0]
    c, x, f, s = ba.optimize(ctx, p)
    oc, ox, of, os_ = O.ba_solve(p)
    used = np.linalg.norm(ox, axis=1) > 0
    ec = np.abs(c - oc).max() / np.abs(oc).max(); ep = (np.linalg.norm(x[used] - ox[used], axis=1) / np.linalg.norm(ox[used], axis=1)).max() if used.any() else 0.0
    ok = s["termination"] == os_["termination"] and abs(s["iterations"] - os_["iterations"]) <= 1 and ec <= 1e-5 and ep <= 1e-3 and s["pcg_iterations_total"] == 0
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"{name} sph={int(sph)} ff={int(ff)} det={int(det)} b={info['band_half_width']} segs={info['band_segments']} seps={info['band_separators']} "
          f"its {s['iterations']}/{os_['iterations']} term {s['termination']}/{os_['termination']} cam {ec:.1e} pt {ep:.1e} pcg {s['pcg_iterations_total']}", flush=True)
print("FUZZ", "FAILED" if bad else "OK", bad)
