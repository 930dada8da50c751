"""Randomised soak of the substructured band solver (GPU box): half-widths 1..19, both block sizes, 1..3 components, 2..12 segments per component -- every
combination of partial 16-tiles, one / two LDS triangles, one- and two-sided separator chains -- against numpy's dense solve.   python scripts/soak_band_solver.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import _band_ref as R
from spherical_sfm_amd import ba
ctx = ba.Context(0)
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0; bad = 0
for case in range(ncase):
    dc = int(rng.choice([3, 6])); b = int(rng.integers(1, 20 if dc == 6 else 39)); P = int(rng.integers(2, 13))
    if b * dc > 114: b = 114 // dc
    ncomp = int(rng.integers(1, 4))
    rows = [int(P * (b + 1) + (P - 1) * b + rng.integers(0, 60)) for _ in range(ncomp)]
    os.environ["SSFM_BAND_SEGMENTS"] = str(P)
    band, A, cp, rhs = R.random_band_system(rows, b, dc, seed=1000 + case)
    X, info = ba.band_solve_probe(ctx, dc, cp, band, rhs)
    xr = np.linalg.solve(A, rhs.T).T
    err = np.abs(X - xr).max() / np.abs(xr).max()
    worst = max(worst, err)
    if info["failed"] != 0 or not (err <= 1e-11):
        bad += 1; print("BAD case", case, dict(dc=dc, b=b, P=P, rows=rows), info, err, flush=True)
print("cases", ncase, "bad", bad, "worst relative error %.2e" % worst)
