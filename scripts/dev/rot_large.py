import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.spatial.transform import Rotation
from spherical_sfm_amd import ba, synth, rotavg
from oracle import oracle as O
ctx = ba.Context(0)
for n in (2000, 4000):
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8)
    Ro, co, so = O.optimize_rotations(R0.copy(), i0, i1, Rrel)
    R, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel)
    ang = np.array([np.linalg.norm(Rotation.from_matrix(a @ b.T).as_rotvec()) for a, b in zip(R, Ro)])
    print(n, "cost", c, co, abs(c - co) / co, "iterations", s["iterations"], so["iterations"], "succ", s["num_successful_steps"], so.get("num_successful_steps"), "max angle", ang.max())
