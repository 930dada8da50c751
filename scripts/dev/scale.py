"""Scale probe (GPU box): BASELINE configs[4]-sized BA (4000 cams / 1.5M pts / 12M obs) on ONE GPU + oracle parity."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from spherical_sfm_amd import synth, ba
from oracle import oracle as O
Nc, Np, K = int(os.environ.get("NC", 4000)), int(os.environ.get("NP", 1500000)), int(os.environ.get("K", 8))
sph = int(os.environ.get("SPH", 0))
t = time.time(); p = synth.make_circle(Nc, Np, K, spherical=bool(sph), focal_fixed=True); print("generate %.1fs" % (time.time() - t), flush=True)
ctx = ba.Context(0)
t = time.time(); adj = ba.BundleAdjuster(ctx, p); print("flatten+upload %.2fs" % (time.time() - t), flush=True)
s = adj.run(); adj.reset(); adj.set_profiling(True)
t = time.time(); s = adj.run(); dt = time.time() - t
cams, pts, f = adj.download()
M = s['num_residual_blocks']
print({k: (round(s[k], 4) if isinstance(s[k], float) else s[k]) for k in ('termination', 'iterations', 'num_linearizations', 'pcg_iterations_total', 'final_cost', 't_solve_s', 'reduced_blocks', 'band_half_width', 'band_segments', 'band_separators', 't_kernel_linearize_ms', 't_kernel_schur_ms', 't_kernel_pcg_ms', 't_kernel_update_ms')})
print("obs/s = %.3e ; B_alg = %.1f MB/iter ; achieved %.1f GB/s" % (M * s['num_linearizations'] / dt, (72 * M + 240 * Np) / 1e6, (72 * M + 240 * Np) * s['num_linearizations'] / dt / 1e9))
for k, v in adj.kernel_times().items(): print("    %-18s launches %6d avg %9.2f us" % (k, v['launches'], 1e3 * v['total_ms'] / v['launches']))
if int(os.environ.get("CHECK", 1)):
    t = time.time(); oc, op, of, os_ = O.ba_solve(p); print("oracle %.1fs its %d" % (time.time() - t, os_['iterations']))
    print("rel cam %.2e pt %.2e ; speedup LM loop %.1fx" % (np.abs(cams - oc).max() / np.abs(oc).max(), (np.linalg.norm(pts - op, axis=1) / np.linalg.norm(op, axis=1)).max(), (os_['t_total_s'] - os_['t_flatten_s']) / dt))
