"""Timing of the pose-graph solves at the config-2 size (300 cameras, edges i -> i+1..8): GPU vs oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from spherical_sfm_amd import ba, rotavg, synth
from oracle import oracle as O
n = int(os.environ.get("NC", 300))
R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(n, 8, seed=5, outlier_frac=0.02)
ctx = ba.Context(0)
for rep in range(3):
    t = time.perf_counter(); R, cost, s = rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel); dt = time.perf_counter() - t
print("gpu optimize_rotations %.2f ms  iterations %d  lin %d  band %d  t_solve %.2f ms" % (1e3 * dt, s["iterations"], s["num_linearizations"], s["band_half_width"], 1e3 * s["t_solve_s"]))
t = time.perf_counter(); Ro, co, so = O.optimize_rotations(R0, i0, i1, Rrel); dto = time.perf_counter() - t
print("oracle %.2f ms iterations %d ; rel diff %.2e ; cost %.6e vs %.6e" % (1e3 * dto, so["iterations"], np.abs(R - Ro).max(), cost, co))
for rep in range(2):
    t = time.perf_counter(); R2, f2, c2, s2 = rotavg.optimize_rotations_and_focal_length(ctx, R0, i0, i1, Rrel, 800.0, 400.0, 1600.0); dt2 = time.perf_counter() - t
print("gpu rotations+focal %.2f ms iterations %d" % (1e3 * dt2, s2["iterations"]))
