"""How much do the fp64 atomics of the BA accumulation (Schur blocks, per-camera sums, LM scalars) move a solve from run to run?
Repeats the config-2 solve and two harder ones; reports the spread of the final cost / parameters in units of the last bit and whether the
iteration / accepted-step counts ever change.  usage: python scripts/soak_ba_repeat.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spherical_sfm_amd import ba, synth
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = ba.Context(0)
cases = [("config 2, general BA, focal fixed", dict(num_cameras=300, num_points=100000, obs_per_point=6, spherical=False, focal_fixed=True)),
         ("config 2, spherical BA, focal free", dict(num_cameras=300, num_points=100000, obs_per_point=6, spherical=True, focal_fixed=False)),
         ("60 cameras, hard start (2 deg rotation noise)", dict(num_cameras=60, num_points=6000, obs_per_point=6, spherical=False, focal_fixed=True, rot_noise_deg=2.0))]
for name, kw in cases:
    prob = synth.make_circle(**kw)
    adj = ba.BundleAdjuster(ctx, prob)
    its, succ, costs, cams = [], [], [], []
    for r in range(rep):
        adj.reset(); s = adj.run(); c, p, f = adj.download()
        its.append(s["iterations"]); succ.append(s["num_successful_steps"]); costs.append(s["final_cost"]); cams.append(np.array(c, copy=True))
    costs = np.array(costs); cams = np.array(cams)
    spread_cost = (costs.max() - costs.min()) / costs.mean()
    spread_cam = (cams.max(0) - cams.min(0)).max() / np.abs(cams).max()
    print(f"{name}: {rep} solves, iterations {sorted(set(its))}, accepted steps {sorted(set(succ))}, distinct final costs {len(set(costs.tolist()))}, "
          f"cost spread {spread_cost:.2e} (eps = 2.2e-16), camera spread {spread_cam:.2e}")
    adj.close()
ctx.close()
