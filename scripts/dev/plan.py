import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spherical_sfm_amd import ba, synth
p = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
for i in range(3):
    t=time.perf_counter(); info,_,_,_ = ba.plan(p); print("plan total %.2f ms" % (1e3*(time.perf_counter()-t)), flush=True)
