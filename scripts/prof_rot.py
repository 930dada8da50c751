"""optimize_rotations at 300 / 2000 / 4000 nodes (edges i -> i+1..8) for rocprofv3 and plain timing.  usage: python scripts/prof_rot.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spherical_sfm_amd import ba, synth, rotavg
ctx = ba.Context(0)
for n in ([int(a) for a in sys.argv[1:]] or [300]):
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8)
    rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel)
    ts = []
    for k in range(5):
        t = time.perf_counter(); Rg, c, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel); ts.append(time.perf_counter() - t)
    print(f"n={n} edges={len(i0)} call {1e3 * min(ts):.3f} ms  solve {1e3 * s['t_solve_s']:.3f} ms  iterations {s['iterations']} linearizations {s['num_linearizations']} cost {c:.12e}")
ctx.close()
