import sys, time
sys.path.insert(0, '/root/repo')
from spherical_sfm_amd import ba, synth
ctx = ba.Context(0)
for K, npts in ((8, 109152), (6, 109152)):
    p = synth.make_circle(300, npts, K, spherical=False, focal_fixed=True, check_in_frame=False, xy_range=0.25)
    adj = ba.BundleAdjuster(ctx, p); adj.run(); adj.reset(); adj.set_profiling(True); s = adj.run()
    print(K, npts, {k: round(1e3 * v['total_ms'] / v['launches'], 1) for k, v in adj.kernel_times().items() if v['launches']})
    adj.close()
