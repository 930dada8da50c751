"""Test helper: a small estimate_pairwise-shaped input (per-frame feature rays + per-pair match lists) with ragged pairs."""
import numpy as np

from spherical_sfm_amd import synth


def indexed_problem(num_frames=6, sizes=(120, 75, 33, 2, 200, 64, 97, 150, 40)):
    """frame f holds the u-rays then the v-rays of pool problem f; pair k matches frame k % F (its u part) with itself (its v part) on a
    random subset of `sizes[k]` correspondences -> (feat_ptr, feat_rays, frame0, frame1, match_ptr, idx0, idx1)"""
    NC = 220
    probs = [synth.make_relative_pose_problem(NC, seed=300 + f, noise=1 / 600, outlier_frac=0.3, rotation_deg=8 + f) for f in range(num_frames)]
    feat_ptr = (np.arange(num_frames + 1) * 2 * NC).astype(np.int32)
    feat_rays = np.ascontiguousarray(np.concatenate([np.concatenate([q[0], q[1]]) for q in probs]))
    rng = np.random.default_rng(9)
    f0 = (np.arange(len(sizes)) % num_frames).astype(np.int32)
    ptr = np.zeros(len(sizes) + 1, np.int32); i0 = []; i1 = []
    for k, n in enumerate(sizes):
        sel = np.sort(rng.choice(NC, n, replace=False)).astype(np.int32)
        i0.append(sel); i1.append(sel + NC); ptr[k + 1] = ptr[k] + n
    return feat_ptr, feat_rays, f0, f0.copy(), ptr, np.concatenate(i0), np.concatenate(i1)
