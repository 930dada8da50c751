"""Soak of the reduced-system solver (GPU box): random block-band systems through ssfm_band_solve_probe -- block size 3 / 6, half-width 1..19 (6x6 blocks: ..30, the packed window of round 4), 1..4 components of
random length, forced segment counts 1..12 (one-sided, two-sided chains, twisted components) -- against numpy's dense solve.  usage: python scripts/soak_band.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa
import _band_ref as R
from spherical_sfm_amd import ba
ctx = ba.Context(0)
rng = np.random.default_rng(2026)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
worst = 0.0
for k in range(cases):
    dc = int(rng.choice([3, 6])); b = int(rng.integers(1, 31 if dc == 6 else 20)); P = int(rng.integers(1, 13))      # round 4: 6x6 blocks up to half-width 30 (packed window)
    ncomp = int(rng.integers(1, 5))
    rows = [int(rng.integers(max(2, b // 2), 60 + 40 * P)) for _ in range(ncomp)]
    os.environ["SSFM_BAND_SEGMENTS"] = str(P)
    band, A, cp, rhs = R.random_band_system(rows, b, dc, seed=1000 + k)
    X, info = ba.band_solve_probe(ctx, dc, cp, band, rhs)
    xr = np.linalg.solve(A, rhs.T).T
    err = np.abs(X - xr).max() / np.abs(xr).max()
    worst = max(worst, err)
    if info["failed"] or not (err <= 1e-11):
        print("FAIL case", k, dict(dc=dc, b=b, P=P, rows=rows), info, err); sys.exit(1)
print("soak ok:", cases, "cases, worst relative error %.2e" % worst)
