"""k_schur_gram launch time for the library in SSFM_LIB_PATH: config 2 general BA, spherical BA, and the configs[4] size.  python scripts/prof_gram_ld.py [big=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from spherical_sfm_amd import ba, synth
ctx = ba.Context(0)
cases = [("general 300x100k K6", dict(num_cameras=300, num_points=100000, obs_per_point=6, spherical=False)),
         ("spherical 300x100k K6", dict(num_cameras=300, num_points=100000, obs_per_point=6, spherical=True))]
if len(sys.argv) > 1 and sys.argv[1] == "1": cases.append(("general 4000x1.5M K8", dict(num_cameras=4000, num_points=1500000, obs_per_point=8, spherical=False)))
out = []
for name, kw in cases:
    p = synth.make_circle(**kw)
    adj = ba.BundleAdjuster(ctx, p)
    adj.reset(); adj.run()
    best = None
    for _ in range(3):
        adj.set_profiling(True); adj.reset(); s = adj.run(); kt = adj.kernel_times(); adj.set_profiling(False)
        us = 1e3 * kt["k_schur_gram"]["total_ms"] / kt["k_schur_gram"]["launches"]
        best = us if best is None else min(best, us)
    out.append(f"{name}: k_schur_gram {best:.1f} us, iterations {s['iterations']}")
    adj.close()
print(os.path.basename(os.environ.get("SSFM_LIB_PATH", "libssfm_hip.so")), "|", " | ".join(out))
