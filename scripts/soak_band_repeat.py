import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import _band_ref as R
from spherical_sfm_amd import ba
ctx = ba.Context(0)
for (dc, b, rows, P) in [(6, 14, [700, 690], 10), (6, 5, [300, 310, 305], 7), (3, 9, [400], 5)]:
    os.environ["SSFM_BAND_SEGMENTS"] = str(P)
    band, A, cp, rhs = R.random_band_system(rows, b, dc, seed=5)
    X0, info = ba.band_solve_probe(ctx, dc, cp, band, rhs)
    xr = np.linalg.solve(A, rhs.T).T
    print(dc, b, rows, P, "err", np.abs(X0 - xr).max() / np.abs(xr).max(), info)
    for rep in range(150):
        X, _ = ba.band_solve_probe(ctx, dc, cp, band, rhs)
        assert np.array_equal(X, X0), rep
print("repeat soak ok")
