"""Test-only generator (uses the oracle, so it lives outside the package)."""
import numpy as np

from spherical_sfm_amd import synth


def make_uncalibrated_loop(O, num_cameras=60, max_offset=3, *, focal_true=1000.0, focal_guess=1300.0, noise_deg=0.05, seed=7):
    """Pose graph as the uncalibrated pipeline sees it (examples/run_spherical_sfm_uncalib.cpp): the relative rotations were
    decomposed from essential matrices estimated in coordinates normalised by the GUESSED focal, i.e. from
    E_guess = T^-1 E_true T^-1, T = diag(f_true/f_guess, f_true/f_guess, 1).  Edges (i, i+d), d = 1..max_offset, open chain
    plus the loop closures back to camera 0..max_offset-1.  O: the oracle module (decomposition of a general E).  Returns (index0, index1, R_rel_guess (E,3,3), R_gt)."""
    Nc = int(num_cameras); rng = np.random.default_rng(seed)
    ang = 2 * np.pi * np.arange(Nc) / Nc
    r_gt = np.stack([np.zeros(Nc), ang, np.zeros(Nc)], axis=1)
    r_gt[:, 1] = np.where(r_gt[:, 1] > np.pi, r_gt[:, 1] - 2 * np.pi, r_gt[:, 1])
    R_gt = synth.so3exp(r_gt)
    i0, i1 = [], []
    for i in range(Nc):
        for d in range(1, max_offset + 1):
            j = i + d
            if j < Nc:
                i0.append(i); i1.append(j)
            elif d == 1 or j - Nc < max_offset:                     # loop closures
                i0.append(i); i1.append(j - Nc)
    i0 = np.array(i0, np.int32); i1 = np.array(i1, np.int32)
    Tinv = np.diag([focal_guess / focal_true, focal_guess / focal_true, 1.0])
    R_rel = np.zeros((len(i0), 3, 3))
    for k, (a, b) in enumerate(zip(i0, i1)):
        Rt = synth.so3exp(rng.normal(0.0, np.deg2rad(noise_deg), size=(1, 3)))[0] @ R_gt[b] @ R_gt[a].T
        Eg = Tinv @ O.make_spherical_essential_matrix(Rt, False) @ Tinv
        r, _ = O.decompose_spherical_essential_matrix(Eg, False)
        R_rel[k] = synth.so3exp(np.asarray(r)[None])[0]
    return i0, i1, R_rel, R_gt


def make_hard_bounded_case(O, seed):
    """optimize_rotations_and_focal_length started far from the solution (random or drifting initial rotations, 10 % corrupted edges,
    focal guess / bounds drawn per seed): the trust-region steps overshoot, so Ceres' projected line search contracts some of them
    (seeds 4, 5, 9, 12, 28, 32 do; oracle.pose_graph_last_line_search_contractions()).  -> R0, i0, i1, R_rel, focal_guess, lo, hi"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(12, 50)); fg = float(rng.choice([400, 700, 1300, 2500, 4000])); nd = float(rng.choice([0.05, 1.0, 3.0, 8.0]))
    i0, i1, Rrel, Rgt = make_uncalibrated_loop(O, n, 3, focal_guess=fg, noise_deg=nd, seed=seed)
    for k in rng.choice(len(i0), max(1, len(i0) // 10), replace=False):
        Rrel[k] = synth.so3exp(rng.normal(size=(1, 3)) * 0.8)[0] @ Rrel[k]
    R0 = np.tile(np.eye(3), (n, 1, 1))
    if seed % 2 == 0:
        for k in range(1, n):
            e = [q for q in range(len(i0)) if i0[q] == k - 1 and i1[q] == k][0]
            R0[k] = Rrel[e] @ R0[k - 1]
    else:
        R0 = synth.so3exp(rng.normal(size=(n, 3)) * 0.5)
    lo, hi = fg * rng.choice([0.2, 0.5, 0.9]), fg * rng.choice([1.1, 2.0, 5.0])
    return R0, i0, i1, Rrel, fg, float(lo), float(hi)
