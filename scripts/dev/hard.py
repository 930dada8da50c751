import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from spherical_sfm_amd import synth, ba
from oracle import oracle as O
ctx = ba.Context(0)
p = synth.make_circle(60, 300, 6, spherical=False, rot_noise_deg=8.0, point_noise=0.25)
cams, pts, f, s = ba.optimize(ctx, p, verbose=1, initial_trust_region_radius=1e12)
oc, op, of, os_ = O.ba_solve(p, verbose=1, initial_trust_region_radius=1e12)
print({k: s[k] for k in ('termination','iterations','num_successful_steps','num_unsuccessful_steps','final_cost')})
print({k: os_[k] for k in ('termination','iterations','num_successful_steps','num_unsuccessful_steps','final_cost')})
print(np.abs(cams-oc).max()/np.abs(oc).max(), (np.linalg.norm(pts-op,axis=1)/np.linalg.norm(op,axis=1)).max())
