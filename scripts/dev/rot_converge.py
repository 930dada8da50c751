"""Converged-minimum comparison of the large pose graphs (VERDICT r4 #1b): both sides with the iteration cap lifted and tight tolerances.
usage: python scripts/dev/rot_converge.py [n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.spatial.transform import Rotation
from oracle import oracle as O
from spherical_sfm_amd import ba, rotavg, synth

ctx = ba.Context(0)
tol = dict(function_tolerance=float(os.environ.get("FT", 1e-16)), gradient_tolerance=1e-16, parameter_tolerance=float(os.environ.get("PT", 1e-16)))
cap = int(os.environ.get("CAP", 8000))
for n in [int(a) for a in sys.argv[1:]] or [2000, 4000]:
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8)
    t = time.time(); R, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel, max_num_iterations=cap, **tol); tg = time.time() - t
    O.pose_graph_test_options(cap, tol["function_tolerance"], tol["gradient_tolerance"], tol["parameter_tolerance"])
    t = time.time(); Ro, co, so = O.optimize_rotations(R0.copy(), i0, i1, Rrel); to = time.time() - t
    O.pose_graph_test_options(0)
    ang = np.linalg.norm(Rotation.from_matrix(np.einsum('nij,nkj->nik', R, Ro)).as_rotvec(), axis=1)
    print(f"n={n} gpu: it {s['iterations']} (ok {s['num_successful_steps']}, rejected {s['num_unsuccessful_steps']}) term {s['termination']} cost {c!r} {tg:.1f}s | oracle: it {so['iterations']} term {so['termination']} cost {co!r} {to:.1f}s | "
          f"dcost/cost {abs(c - co) / co:.2e} max angle {ang.max():.2e} rad", flush=True)
    # cross starts: is each side's answer a minimum by the OTHER side's rules?  (a flat valley shows as: cost equal to rounding, rotations apart)
    t = time.time(); R2, c2, s2 = rotavg.optimize_rotations(ctx, Ro, i0, i1, Rrel, max_num_iterations=cap, **tol)
    O.pose_graph_test_options(cap, tol["function_tolerance"], tol["gradient_tolerance"], tol["parameter_tolerance"])
    Ro2, co2, so2 = O.optimize_rotations(R.copy(), i0, i1, Rrel)
    O.pose_graph_test_options(0)
    a2 = np.linalg.norm(Rotation.from_matrix(np.einsum('nij,nkj->nik', R2, Ro)).as_rotvec(), axis=1).max()
    a3 = np.linalg.norm(Rotation.from_matrix(np.einsum('nij,nkj->nik', Ro2, R)).as_rotvec(), axis=1).max()
    a4 = np.linalg.norm(Rotation.from_matrix(np.einsum('nij,nkj->nik', R2, Ro2)).as_rotvec(), axis=1).max()
    print(f"    gpu from the oracle's answer: it {s2['iterations']} cost {c2!r} moved {a2:.2e} rad | oracle from the gpu's answer: it {so2['iterations']} cost {co2!r} moved {a3:.2e} rad | "
          f"the two restarted answers apart {a4:.2e} rad", flush=True)
