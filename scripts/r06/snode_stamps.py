"""phase stamps of k_snode_solve on the config-2 structure (four rings of 75 cameras):  python scripts/r06/snode_stamps.py [nr]   (GPU box)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SSFM_SNODE_STAMPS"] = "1"
import numpy as np
from spherical_sfm_amd import ba
from test_snode_gpu import window_system
ctx = ba.Context(0)
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rp, ci, blk, A, rhs2 = window_system([(75, 5, True)] * 4, 6, seed=1)
Y, info = ba.snode_solve_probe(ctx, 6, rp, ci, blk, rhs2, nr=nr)
xr = np.linalg.solve(A, rhs2.T).T
print(info, np.abs(Y[:nr] - xr[:nr]).max() / np.abs(xr).max())
