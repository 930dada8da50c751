"""Synthetic feature-track directory in the reference's on-disk format (examples/spherical_sfm_io.cpp:10-59):
keyframes.txt, features.dat, matches.dat + an intrinsics file, for an outward-facing camera circle."""
import os
import struct

import numpy as np

from spherical_sfm_amd import synth


def write_tracks(outdir, num_cameras=40, num_points=1500, K=6, max_offset=3, focal=1000.0, cx=960.0, cy=540.0, pixel_noise=0.3, rot_noise_deg=0.05, seed=5,
                 focal_guess=None, oracle=None, raw_matches=False, wrong_match_frac=0.0):
    """focal_guess + oracle: the matches carry the rotations a pairwise estimator would find in coordinates normalised by the GUESSED
    focal (decomposition of T^-1 E_true T^-1, T = diag(f/f_guess, f/f_guess, 1)), as in the uncalibrated pipeline."""
    rng = np.random.default_rng(seed)
    prob = synth.make_circle(num_cameras, num_points, K, spherical=True, focal_fixed=True, pixel_noise=pixel_noise, rot_noise_deg=0.0, seed=seed, focal=focal)
    R_gt = synth.so3exp(prob.gt_cameras[:, 3:])
    feats = [[] for _ in range(num_cameras)]; fid = {}                      # per keyframe: list of (x, y); (camera, point) -> feature index
    for o in range(len(prob.obs_cam)):
        c, p = int(prob.obs_cam[o]), int(prob.obs_pt[o])
        fid[(c, p)] = len(feats[c]); feats[c].append((prob.obs_xy[o, 0] + cx, prob.obs_xy[o, 1] + cy))
    pts_of = [set() for _ in range(num_cameras)]
    for (c, p) in fid:
        pts_of[c].add(p)
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "keyframes.txt"), "w") as f:
        f.write("%d\n" % num_cameras)
        for i in range(num_cameras):
            f.write("%d %06d.jpg\n" % (i, i + 1))
    with open(os.path.join(outdir, "features.dat"), "wb") as f:
        for i in range(num_cameras):
            f.write(struct.pack("i", len(feats[i])))
            for (x, y) in feats[i]:
                f.write(struct.pack("2f", x, y)); f.write(b"\0" * (4 * 128))
    matches = []
    for i in range(num_cameras):
        for d in range(1, max_offset + 1):
            j = (i + d) % num_cameras
            shared = sorted(pts_of[i] & pts_of[j])
            if len(shared) < 8:
                continue
            a, b = (i, j)                                                   # index0 = i (the chain needs (k-1, k)); closures keep (Nc-1, 0)
            if raw_matches and a > b:
                a, b = b, a                                                 # a matcher stores every pair once, index0 < index1
            noise = synth.so3exp(rng.normal(0, np.deg2rad(rot_noise_deg), (1, 3)))[0]
            Rrel = noise @ R_gt[b] @ R_gt[a].T
            if focal_guess is not None:
                Tinv = np.diag([focal_guess / focal, focal_guess / focal, 1.0])
                r, _ = oracle.decompose_spherical_essential_matrix(Tinv @ oracle.make_spherical_essential_matrix(Rrel, False) @ Tinv, False)
                Rrel = synth.so3exp(np.asarray(r)[None])[0]
            pairs = dict((fid[(a, p)], fid[(b, p)]) for p in shared)
            if raw_matches:
                Rrel = np.eye(3)
                for fa in list(pairs)[::max(1, int(round(1 / wrong_match_frac)))] if wrong_match_frac > 0 else []:
                    pairs[fa] = int(rng.integers(len(feats[b])))            # wrong pairing
            matches.append((a, b, sorted(pairs.items()), Rrel))
    with open(os.path.join(outdir, "matches.dat"), "wb") as f:
        f.write(struct.pack("i", len(matches)))
        for (a, b, m, R) in matches:
            f.write(struct.pack("3i", a, b, len(m)))
            for (u, v) in m:
                f.write(struct.pack("2i", u, v))
            f.write(np.asarray(R, np.float64).T.tobytes())                  # Eigen: column-major
    with open(os.path.join(outdir, "intrinsics.txt"), "w") as f:
        f.write("%.17g %.17g %.17g\n" % (focal, cx, cy))
    return dict(R_gt=R_gt, prob=prob, num_matches=len(matches))
