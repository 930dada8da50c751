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
