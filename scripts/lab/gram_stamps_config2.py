import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spherical_sfm_amd import ba, synth
ctx = ba.Context(0)
prob = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
_, _, _, s = ba.optimize(ctx, prob)
print(s["iterations"])
