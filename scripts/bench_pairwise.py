"""BASELINE configs[3]: exhaustive pairwise spherical RANSAC of a 2000-frame circle = 1 999 000 image pairs x 500 correspondences
(problem_generator-style geometry, 30 % uniform outliers, noise 1 px / f, threshold (2 px / f)^2, f = 1000), one GPU.
The 48 GB of rays of the full pair list are not materialised: a pool of distinct pairs is generated once and submitted slab after
slab (100 000 pairs = 2.4 GB per ssfm_ransac_batch call, each call streaming its own slabs through the pinned double buffer).
Prints one JSON line: pairs/s end to end (host buffers in, results out), and the FP64 rate of the Sampson scoring.
usage: python scripts/bench_pairwise.py [total_pairs] [pairs_per_call] [mode] [indexed]
indexed = 1: the same pairs through ssfm_ransac_batch_indexed -- per-frame feature rays uploaded once (frame k = the 500 u-rays and the 500 v-rays of
pool problem k) and 8 bytes of match indices per correspondence instead of 48 bytes of rays (8 GB instead of 48 GB at the full size)."""
import json
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from spherical_sfm_amd import synth, ba, ransac  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 1999000
per_call = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
indexed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
F = 1000.0; THR = (2 / F) ** 2; NC = 500; POOL = 2000
probs = [synth.make_relative_pose_problem(NC, seed=1000 + k, noise=1 / F, outlier_frac=0.3, rotation_deg=1 + (k % 60)) for k in range(POOL)]
reps = (per_call + POOL - 1) // POOL
U = np.ascontiguousarray(np.concatenate([p[0] for p in probs] * reps)[:per_call * NC]); V = np.ascontiguousarray(np.concatenate([p[1] for p in probs] * reps)[:per_call * NC])
ptr = (np.arange(per_call + 1, dtype=np.int64) * NC).astype(np.int32)
import os
KW = {k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in os.environ.get("SSFM_PW_OPTS", "").split(",") if kv)}   # option overrides, e.g. final_least_squares=0
ctx = ba.Context(0)
if indexed:
    feat_ptr = (np.arange(POOL + 1, dtype=np.int64) * 2 * NC).astype(np.int32)
    feat_rays = np.ascontiguousarray(np.concatenate([np.concatenate([p[0], p[1]]) for p in probs]))
    fr = (np.arange(per_call) % POOL).astype(np.int32)
    m0 = np.tile(np.arange(NC, dtype=np.int32), per_call); m1 = m0 + NC
    del U, V
    def run(n):
        return ransac.estimate_indexed(ctx, feat_ptr, feat_rays, fr[:n], fr[:n], ptr[:n + 1], m0[:n * NC], m1[:n * NC], THR, min_num_inliers=20, mode=mode, **KW)
else:
    def run(n):
        return ransac.estimate_flat(ctx, ptr[:n + 1], U[:n * NC], V[:n * NC], THR, min_num_inliers=20, mode=mode, **KW)
run(2000)      # warm-up (module load, pinned buffers)
done = 0; t0 = time.perf_counter(); its = 0; acc = 0
while done < total:
    n = min(per_call, total - done)
    o = run(n)
    its += int(o["iterations"].sum()) if mode == 1 else n * 1024; acc += int((o["num_inliers"] > 20).sum()); done += n
dt = time.perf_counter() - t0
flop = its * 4 * NC * 48.0                     # models per sample x rays x flop of one Sampson score (48)
print(json.dumps({"workload": f"{total} pairs x {NC} correspondences, 30% outliers, mode {'reference trace' if mode == 1 else 'fixed budget 1024'}",
                  "pairs_per_s_end_to_end": total / dt, "seconds": dt, "accepted_pairs": acc, "mean_iterations": its / total,
                  "scoring_tflops_end_to_end": flop / dt / 1e12, "h2d_gb": total * NC * (8 if indexed else 48) / 1e9, "entry_point": "ssfm_ransac_batch_indexed" if indexed else "ssfm_ransac_batch"}))
ctx.close()
