"""Per-iteration comparison of optimize_rotations_and_focal_length against the oracle (caps 1..K).  usage: python scripts/dev/focal_steps.py n K"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.spatial.transform import Rotation
from oracle import oracle as O
from spherical_sfm_amd import ba, rotavg, synth
n, K = int(sys.argv[1]), int(sys.argv[2])
ctx = ba.Context(0)
R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8, noise_deg=0.2, outlier_frac=0.02)
for k in range(1, K + 1):
    O.pose_graph_test_options(k)
    Ro, fo, co, so = O.optimize_rotations_and_focal_length(R0.copy(), i0, i1, Rrel, 800.0, 400.0, 1600.0)
    O.pose_graph_test_options(0)
    R, f, c, s = rotavg.optimize_rotations_and_focal_length(ctx, R0, i0, i1, Rrel, 800.0, 400.0, 1600.0, max_num_iterations=k)
    ang = np.linalg.norm(Rotation.from_matrix(np.einsum('nij,nkj->nik', R, Ro)).as_rotvec(), axis=1).max()
    print(f"RING={os.environ.get('SSFM_RING', '1')} cap {k}: gpu it {s['iterations']} ok {s['num_successful_steps']} cost {c!r} f {f!r} | oracle it {so['iterations']} ok {so['num_successful_steps']} cost {co!r} f {fo!r} | angle {ang:.2e}", flush=True)
