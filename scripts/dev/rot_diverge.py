"""How fast the device and the oracle part on the 2000 / 4000-node pose graphs: relative cost difference after k LM iterations (both capped at k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.spatial.transform import Rotation
from spherical_sfm_amd import ba, synth, rotavg
from oracle import oracle as O
ctx = ba.Context(0)
for n in (2000, 4000):
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8)
    for k in (5, 8, 10, 12, 15, 20, 30, 40, 50):
        O.pose_graph_test_options(k)
        Ro, co, so = O.optimize_rotations(R0.copy(), i0, i1, Rrel)
        R, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel, max_num_iterations=k)
        ang = np.array([np.linalg.norm(Rotation.from_matrix(a @ b.T).as_rotvec()) for a, b in zip(R, Ro)])
        print(n, k, "rel cost diff %.2e" % (abs(c - co) / co), "max angle %.2e" % ang.max(), "accepted", s["num_successful_steps"], so["num_successful_steps"], "iterations", s["iterations"], so["iterations"], flush=True)
O.pose_graph_test_options(0)
