"""Ring-native reduced solve (round 5, band_ring.h) against the oracle and against the folded plan: python scripts/dev/ring.py [case ...]
cases: ring1000 sph2000 ragged14 rot2000 rot4000 c4 (configs[4] size, timing only unless CHECK=1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spherical_sfm_amd import ba, rotavg, synth
from oracle import oracle as O

def rel(a, b): return np.abs(a - b).max() / np.abs(b).max()
def prel(a, b): return (np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)).max()

def run_ba(name, p, check=True):
    ctx = ba.Context(0)
    adj = ba.BundleAdjuster(ctx, p)
    s = adj.run(); adj.reset()
    best = 1e9
    for _ in range(5): t = time.time(); s = adj.run(); best = min(best, time.time() - t); adj.reset()
    print(f"[{name}] unprofiled best of 5: {1e3 * best:.3f} ms = {1e6 * best / max(s['num_linearizations'], 1):.0f} us/iter  obs/s {s['num_residual_blocks'] * s['num_linearizations'] / best:.3e}", flush=True)
    adj.set_profiling(True)
    t = time.time(); s = adj.run(); dt = time.time() - t
    cams, pts, f = adj.download()
    print(f"[{name}] its {s['iterations']} lin {s['num_linearizations']} term {s['termination']} pcg {s['pcg_iterations_total']} b {s['band_half_width']} segs {s['band_segments']} seps {s['band_separators']} "
          f"solve {1e3 * dt:.2f} ms = {1e6 * dt / max(s['num_linearizations'], 1):.0f} us/iter  obs/s {s['num_residual_blocks'] * s['num_linearizations'] / dt:.3e}", flush=True)
    for k, v in adj.kernel_times().items():
        if v['launches']: print("    %-20s launches %6d avg %9.2f us  per-iteration %8.1f us" % (k, v['launches'], 1e3 * v['total_ms'] / v['launches'], 1e3 * v['total_ms'] / max(s['num_linearizations'], 1)))
    if check:
        oc, op, of, os_ = O.ba_solve(p)
        print(f"    oracle its {os_['iterations']}: rel cam {rel(cams, oc):.2e} pt {prel(pts, op):.2e}", flush=True)
    adj.close(); ctx.close()

def run_rot(name, n):
    from scipy.spatial.transform import Rotation
    ctx = ba.Context(0)
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8)
    cap = int(os.environ.get("ROT_IT", 8))
    R, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel, max_num_iterations=cap)
    t = time.time(); R, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel, max_num_iterations=cap); dt = time.time() - t
    O.pose_graph_test_options(cap); Ro, co, so = O.optimize_rotations(R0.copy(), i0, i1, Rrel); O.pose_graph_test_options(0)
    ang = np.linalg.norm(Rotation.from_matrix(np.einsum('nij,nkj->nik', R, Ro)).as_rotvec(), axis=1).max()
    t = time.time(); R5, c5, s5 = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel); dt5 = time.time() - t
    print(f"[{name}] {cap} its: gpu it {s['iterations']} ok {s['num_successful_steps']} cost {c!r} | oracle it {so['iterations']} ok {so['num_successful_steps']} cost {co!r} | dcost {abs(c - co) / co:.2e} angle {ang:.2e} | "
          f"{1e3 * dt:.2f} ms; default 50-iteration call {1e3 * dt5:.2f} ms ({s5['iterations']} its)", flush=True)
    ctx.close()

cases = sys.argv[1:] or ["ring1000", "sph2000", "ragged14", "rot2000"]
chk = int(os.environ.get("CHECK", 1))
for c in cases:
    if c == "ring1000": run_ba(c, synth.make_circle(1000, 40000, 6, spherical=False, focal_fixed=True, seed=3), chk)
    elif c == "ring1100f": run_ba(c, synth.make_circle(1100, 33000, 6, spherical=False, focal_fixed=False, seed=12), chk)
    elif c == "sph2000": run_ba(c, synth.make_circle(2000, 24000, 6, spherical=True, focal_fixed=True, seed=8), chk)
    elif c == "ragged14": run_ba(c, synth.make_ragged_circle(300, 600000, 3, 14), chk)
    elif c == "ragged8": run_ba(c, synth.make_ragged_circle(300, 600000, 3, 8), chk)
    elif c.startswith("bigragged"): run_ba(c, synth.make_ragged_circle(1000, 3000000, 3, int(c[9:])), chk)
    elif c.startswith("ragged"): run_ba(c, synth.make_ragged_circle(300, 600000, 3, int(c[6:])), chk)
    elif c.startswith("sragged"): run_ba(c, synth.make_ragged_circle(300, 600000, 3, int(c[7:]), spherical=True), chk)
    elif c.startswith("circle"): run_ba(c, synth.make_circle(int(c[6:]), 100 * int(c[6:]), 6, spherical=False, focal_fixed=True, seed=3), chk)
    elif c == "c4": run_ba(c, synth.make_circle(4000, 1500000, 8, spherical=False, focal_fixed=True), int(os.environ.get("CHECK", 0)))
    elif c == "config2": run_ba(c, synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True), chk)
    elif c.startswith("rot"): run_rot(c, int(c[3:]))
