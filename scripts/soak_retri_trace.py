"""Soak of the reference-trace Retriangulate: random problem shapes / noise levels / outlier fractions / track lengths, device vs oracle --
iteration counts, LO runs, inlier flags and zeroing decisions must be IDENTICAL, points <= 1e-9.  usage: python scripts/soak_retri_trace.py [cases]"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spherical_sfm_amd import ba, synth
from oracle import oracle as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = ba.Context(0)
rng = np.random.default_rng(2024)
tot = 0; bad_cases = 0; worst = 0.0
for k in range(cases):
    Nc = int(rng.choice([24, 60, 90, 200])); K = int(rng.choice([3, 4, 5, 6, 8, 12, 20])); Np = int(rng.integers(300, 3000))
    noise = float(rng.choice([0.0, 0.2, 0.5, 1.5, 4.0])); frac = float(rng.choice([0.0, 0.1, 0.3, 0.6]))
    K = min(K, Nc // 4)
    prob = synth.make_circle(Nc, Np, K, rot_noise_deg=float(rng.choice([0.0, 0.5])), pixel_noise=noise, seed=int(rng.integers(1, 10**6)), check_in_frame=False, xy_range=0.25)
    if frac > 0:
        synth.corrupt_observations(prob, frac, seed=int(rng.integers(1, 10**6)))
    if k % 3 == 0:                                                     # ragged: drop a random tail of every track (some below 3 observations)
        keep_n = rng.integers(1, K + 1, Np)
        order = np.argsort(prob.obs_pt, kind="stable"); start = np.searchsorted(prob.obs_pt[order], np.arange(Np))
        rank = np.zeros(len(prob.obs_pt), int); rank[order] = np.arange(len(order)) - start[prob.obs_pt[order]]
        sel = rank < keep_n[prob.obs_pt]
        prob = dataclasses.replace(prob, obs_xy=prob.obs_xy[sel].copy(), obs_cam=prob.obs_cam[sel], obs_pt=prob.obs_pt[sel])
    if k % 5 == 0:                                                     # some cameras turned around: points behind them (DBL_MAX errors)
        cams = prob.cameras.copy(); cams[::7, 3:] += [0.0, np.pi, 0.0]; prob = dataclasses.replace(prob, cameras=cams)
    Xo, no, ito, loo, flo = O.retriangulate_ex(prob, 16)
    Xg, ng, itg, log_, flg = ba.retriangulate_ex(ctx, prob)
    same = np.array_equal(itg, ito) and np.array_equal(log_, loo) and np.array_equal(ng, no) and np.array_equal(flg, flo) and np.array_equal(Xg.any(1), Xo.any(1))
    nz = Xo.any(1)
    rel = (np.linalg.norm(Xg - Xo, axis=1)[nz] / np.linalg.norm(Xo[nz], axis=1)).max() if nz.any() else 0.0
    tot += Np; worst = max(worst, rel)
    if not same or rel > 1e-9:
        bad_cases += 1
        print(f"case {k}: Nc {Nc} Np {Np} K {K} noise {noise} outliers {frac}: trace identical {same}, differing points: it {(itg != ito).sum()} lo {(log_ != loo).sum()} nin {(ng != no).sum()}, max rel {rel:.2e}")
print(f"{cases} cases, {tot} points: {bad_cases} cases with any difference; worst point difference {worst:.2e}; max iterations seen {int(ito.max())}")
ctx.close()
