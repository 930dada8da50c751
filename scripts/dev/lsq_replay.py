"""debug: replay every LeastSquares call of the oracle's LO-MSAC run of one pair through the device probe (same start model, same ray
subset): where does the device fit leave the oracle's, and which Levenberg-Marquardt rule flipped?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spherical_sfm_amd import ba, synth, ransac
from oracle import oracle as O
THR = (2 / 600) ** 2
def frob_err(E, Es):
    a = E / np.linalg.norm(E); b = Es / np.linalg.norm(Es); return min(np.linalg.norm(a - b), np.linalg.norm(a + b))
ctx = ba.Context(0)
tot = 0; bad = 0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    u, v, R, E, inl = synth.make_relative_pose_problem(200, seed=500 + k, noise=1 / 600, outlier_frac=0.35, rotation_deg=5 + (k % 30))
    r, log = O.lsq_log(lambda: O.lomsac_pair(u, v, THR, num_lo_steps=10, num_lsq_iterations=4, final_least_squares=False, min_num_inliers=20))
    lists = [l["sample"] for l in log]; starts = [l["E_in"] for l in log]
    for wave in (True, False):
        Eg, x, it, status, c0, c1 = ransac.sampson_refine_probe_ex(ctx, u, v, lists, starts, wave=wave)
        for i, l in enumerate(log):
            fe = frob_err(Eg[i], l["E_out"]); tot += 1
            if fe > 1e-9 or it[i] != l["iterations"]:
                bad += 1
                o = O.sampson_least_squares_ex(u, v, l["sample"], l["E_in"])
                print(f"pair {k} call {i} wave={wave}: n={len(l['sample'])} iterations dev {it[i]} oracle {l['iterations']} status {status[i]} E err {fe:.2e} "
                      f"cost dev {c1[i]:.6e} oracle {o['final_cost']:.6e} initial {c0[i]:.6e}/{o['initial_cost']:.6e} termination {o['termination']} "
                      f"succ/unsucc {o['successful']}/{o['unsuccessful']} |x - x_o| r {np.abs(x[i][:3] - l['x'][:3]).max():.2e} t {np.abs(x[i][3:] - l['x'][3:]).max():.2e}")
print("calls", tot, "differing", bad)
ctx.close()
