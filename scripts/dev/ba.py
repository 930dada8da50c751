"""Developer probe (GPU box): config-2 BA in the four modes, timings + parity vs the oracle."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from spherical_sfm_amd import synth, ba
from oracle import oracle as O
Nc, Np = int(os.environ.get("NC", 300)), int(os.environ.get("NP", 100000))
verbose = int(os.environ.get("VERBOSE", 0)); check = int(os.environ.get("CHECK", 1)); prof = int(os.environ.get("PROF", 0))
ctx = ba.Context(0)
for sph, ff in [(True, True), (True, False), (False, True), (False, False)]:
    p = synth.make_circle(Nc, Np, 6, spherical=sph, focal_fixed=ff)
    adj = ba.BundleAdjuster(ctx, p, verbose=verbose, preconditioner=int(os.environ.get("PRECOND", 0)))
    s = adj.run()
    adj.reset(); adj.set_profiling(bool(prof)); t = time.time(); s = adj.run(); dt = time.time() - t
    cams, pts, f = adj.download()
    print("spherical=%d focal_fixed=%d" % (sph, ff), {k: (round(s[k], 4) if isinstance(s[k], float) else s[k]) for k in ('termination', 'iterations', 'num_linearizations', 'pcg_iterations_total', 'final_cost', 't_solve_s', 't_kernel_linearize_ms', 't_kernel_schur_ms', 't_kernel_pcg_ms', 't_kernel_update_ms')},
          "obs/s=%.3e" % (s['num_residual_blocks'] * s['num_linearizations'] / dt))
    if prof:
        for k, v in adj.kernel_times().items(): print("    %-18s launches %6d total %9.3f ms avg %8.2f us" % (k, v['launches'], v['total_ms'], 1e3 * v['total_ms'] / v['launches']))
    if check:
        oc, op, of, os_ = O.ba_solve(p)
        print("    oracle its %d cost %.6f t=%.2fs | rel cam %.2e pt %.2e f %.2e" % (os_['iterations'], os_['final_cost'], os_['t_total_s'],
              np.abs(cams - oc).max() / np.abs(oc).max(), (np.linalg.norm(pts - op, axis=1) / np.linalg.norm(op, axis=1)).max(), abs(f - of) / of))
    adj.close()
