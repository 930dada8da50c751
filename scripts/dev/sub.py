"""Debug: the band solve probe against numpy, stage by stage.  usage: python scripts/dbg_sub.py [dc] [b] [rows] [P]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import _band_ref as R
from spherical_sfm_amd import ba

dc = int(sys.argv[1]) if len(sys.argv) > 1 else 6
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
P = int(sys.argv[4]) if len(sys.argv) > 4 else 2
os.environ["SSFM_BAND_SEGMENTS"] = str(P)
band, A, cp, rhs = R.random_band_system([rows, rows + 7], b, dc, seed=1)
ctx = ba.Context(0)
X, info, Z, D, T = ba.band_solve_probe(ctx, dc, cp, band, rhs, dump=True)
xr = np.linalg.solve(A, rhs.T).T
print(info, "solution rel err", np.abs(X - xr).max() / np.abs(xr).max())
segs, seps = R.segment_table(cp, b, P)
print("segs", segs, "seps", seps)
if info["separators"]:
    Zr, Dr, Tr = R.substructure_intermediates(A, rhs, segs, seps, b, dc)
    ns = len(seps)
    print("Z err", np.abs(Z - Zr).max(), "of", np.abs(Zr).max())
    for (r0, r1, re, hl) in segs:
        if hl:
            e = np.abs(Z[:, r0*dc:re*dc] - Zr[:, r0*dc:re*dc]); print("  seg", r0, r1, re, "Z err pivots", e[:, :(r1-r0)*dc].max(), "cont", e[:, (r1-r0)*dc:].max() if re > r1 else 0.0)
    print("D err", [float(np.abs(np.tril(D[s]) - Dr[s]).max()) for s in range(ns)], "of", np.abs(Dr).max())
    print("T err", [float(np.abs(T[s] - Tr[s]).max()) for s in range(ns)], "of", np.abs(Tr).max())
    for (r0, r1, re, hl) in segs:
        e = np.abs(X[:, r0*dc:r1*dc] - xr[:, r0*dc:r1*dc]).max(); print("  seg", r0, r1, "x err", e)
    for (p0, rs) in seps:
        print("  sep", p0, "x err", np.abs(X[:, p0*dc:(p0+b)*dc] - xr[:, p0*dc:(p0+b)*dc]).max())
