"""One pass over the non-BA kernels at the BASELINE config-2 sizes, for rocprofv3 (scripts/gpu_profile_misc.sh):
optimize_rotations + optimize_rotations_and_focal_length (300 cameras, edges i -> i+1..8), Retriangulate (300 / 100k / 600k),
the 1024-trial focal search (300 cameras, 2372 matches), pairwise RANSAC (20 000 pairs x 500, both modes)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from spherical_sfm_amd import ba, rotavg, ransac, synth
ctx = ba.Context(0)
R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(300, 8, seed=5, outlier_frac=0.02)
for _ in range(2):
    rotavg.optimize_rotations(ctx, R0, i0, i1, Rrel)
    rotavg.optimize_rotations_and_focal_length(ctx, R0, i0, i1, Rrel, 800.0, 400.0, 1600.0)
focals = np.linspace(400.0, 1600.0, 1024)
for _ in range(2):
    rotavg.focal_search(ctx, 300, i0, i1, Rrel, 800.0, focals)
prob = synth.make_circle(300, 100000, 6, rot_noise_deg=0.0, pixel_noise=0.5, seed=3)
synth.corrupt_observations(prob, 0.1, seed=5)
for _ in range(2):
    ba.retriangulate(ctx, prob)
probs = [synth.make_relative_pose_problem(500, seed=1000 + k, noise=1e-3, outlier_frac=0.3, rotation_deg=1 + (k % 60)) for k in range(500)]
U = np.concatenate([p[0] for p in probs] * 40); V = np.concatenate([p[1] for p in probs] * 40); ptr = (np.arange(20001) * 500).astype(np.int32)
for mode in (1, 0):
    for _ in range(2):
        o = ransac.estimate_flat(ctx, ptr, U, V, (2e-3) ** 2, min_num_inliers=20, mode=mode)
print("done", int(o["iterations"].sum()))
ctx.close()
