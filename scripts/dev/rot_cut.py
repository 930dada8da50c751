import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from spherical_sfm_amd import ba, rotavg, synth
from oracle import oracle as O
R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(1100, 4)
Ro, co, so = O.optimize_rotations(R0.copy(), i0, i1, Rrel)
print("oracle", co, so["iterations"])
for env in ({}, {"SSFM_BAND_SEGMENTS": "1"}, {"SSFM_BAND_MERGE": "0"}, {"SSFM_BAND_MERGE": "0", "SSFM_BAND_SEGMENTS": "1"}):
    for k in ("SSFM_BAND_SEGMENTS", "SSFM_BAND_MERGE"): os.environ.pop(k, None)
    os.environ.update(env)
    ctx = ba.Context(0)
    R, cost, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel)
    print(env, cost, (cost - co) / co, s["iterations"], s["pcg_iterations_total"], s["band_half_width"], "maxdiff", np.abs(R - Ro).max())
