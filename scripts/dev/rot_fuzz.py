"""Random pose graphs (ring of n nodes, each linked to its next `deg` neighbours) through the planner's fold / ring decision: the first iterations against the oracle.
python scripts/dev/rot_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.spatial.transform import Rotation
from spherical_sfm_amd import ba, rotavg, synth
from oracle import oracle as O
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = ba.Context(0); bad = 0
for it in range(ncase):
    n = int(rng.integers(100, 2600)); deg = int(rng.integers(2, 12)); focal = bool(rng.integers(0, 3) == 0); cap = 4
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, deg, noise_deg=0.2, outlier_frac=0.02, seed=int(rng.integers(1, 10**6)))
    O.pose_graph_test_options(cap)
    try:
        if focal: Ro, fo, co, so = O.optimize_rotations_and_focal_length(R0.copy(), i0, i1, Rrel, 800.0, 400.0, 1600.0)
        else: Ro, co, so = O.optimize_rotations(R0.copy(), i0, i1, Rrel)
    finally: O.pose_graph_test_options(0)
    if focal: R, f, c, s = rotavg.optimize_rotations_and_focal_length(ctx, R0, i0, i1, Rrel, 800.0, 400.0, 1600.0, max_num_iterations=cap)
    else: R, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel, max_num_iterations=cap)
    ang = np.linalg.norm(Rotation.from_matrix(np.einsum('nij,nkj->nik', R, Ro)).as_rotvec(), axis=1).max()
    ok = s["iterations"] == so["iterations"] and s["num_successful_steps"] == so["num_successful_steps"] and abs(c - co) <= 1e-6 * co and ang <= 1e-4
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"n={n} deg={deg} focal={int(focal)} its {s['iterations']}/{so['iterations']} ok-steps {s['num_successful_steps']}/{so['num_successful_steps']} dcost {abs(c - co) / co:.1e} angle {ang:.1e}", flush=True)
print("ROT FUZZ", "FAILED" if bad else "OK", bad)
