"""Oracle restatement of SfM::Retriangulate (reference src/sfm.cpp:156-192 + src/triangulation_estimator.cpp) on CPU:
known-answer cases.  The reference holds no fixtures for this path (parity unpinned)."""
import dataclasses

import numpy as np

from spherical_sfm_amd import synth


def _exact(prob):
    p = dataclasses.replace(prob, cameras=prob.gt_cameras.copy())
    return p


def test_noise_free_recovers_ground_truth(oracle):
    prob = _exact(synth.make_circle(60, 300, 6, rot_noise_deg=0.0, pixel_noise=0.0, seed=1))
    prob.points = np.zeros_like(prob.points)                       # the initial value is irrelevant (src/sfm.cpp:172)
    X, nin = oracle.retriangulate(prob, 4)
    assert (nin == 6).all()
    assert (np.linalg.norm(X - prob.gt_points, axis=1) / np.linalg.norm(prob.gt_points, axis=1)).max() < 1e-7


def test_fewer_than_three_observations_become_zero(oracle):
    prob = _exact(synth.make_circle(60, 200, 6, rot_noise_deg=0.0, pixel_noise=0.0, seed=2))
    keep = ~((prob.obs_pt < 50) & (np.arange(len(prob.obs_pt)) % 6 >= 2))   # points 0..49 keep 2 observations
    prob = dataclasses.replace(prob, obs_xy=prob.obs_xy[keep], obs_cam=prob.obs_cam[keep], obs_pt=prob.obs_pt[keep])
    X, nin = oracle.retriangulate(prob, 4)
    assert not X[:50].any() and (nin[:50] == 0).all()               # src/sfm.cpp:173
    assert X[50:].any(axis=1).all() and (nin[50:] == 6).all()


def test_gross_outlier_is_rejected(oracle):
    prob = _exact(synth.make_circle(60, 400, 6, rot_noise_deg=0.0, pixel_noise=0.0, seed=3))
    bad = synth.corrupt_observations(prob, 0.25, seed=7)
    X, nin = oracle.retriangulate(prob, 4)
    mask = np.zeros(len(X), bool); mask[bad] = True
    assert (nin[mask] == 5).all() and (nin[~mask] == 6).all()
    assert (np.linalg.norm(X - prob.gt_points, axis=1) / np.linalg.norm(prob.gt_points, axis=1)).max() < 1e-6


def test_point_behind_all_cameras_or_inconsistent_becomes_zero(oracle):
    # three observations that do not agree on any point within 2 px: < 3 inliers -> zero (src/sfm.cpp:186)
    prob = _exact(synth.make_circle(30, 100, 3, rot_noise_deg=0.0, pixel_noise=0.0, seed=4))
    xy = prob.obs_xy.copy()
    sel = np.nonzero(prob.obs_pt == 7)[0]
    xy[sel[0]] += [0.0, 60.0]
    prob.obs_xy = xy
    X, nin = oracle.retriangulate(prob, 2)
    assert not X[7].any() and nin[7] < 3
    assert X[np.arange(100) != 7].any(axis=1).all()


def test_retriangulate_golden(oracle):
    """tests/golden/retriangulate.npz: the oracle reproduces its committed points and inlier counts (std::mt19937 streams included)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "retriangulate.npz"))
    Np = len(g["points"])
    p = synth.BAProblem(cameras=g["cameras"], points=np.ones((Np, 3)), focal=float(g["focal"]), obs_xy=g["obs_xy"], obs_cam=g["obs_cam"], obs_pt=g["obs_pt"],
                        rot_fixed=np.zeros(len(g["cameras"]), np.uint8), trans_fixed=np.ones(len(g["cameras"]), np.uint8), pt_fixed=np.zeros(Np, np.uint8),
                        focal_fixed=True, gt_cameras=g["cameras"], gt_points=g["points"], gt_focal=0.0)
    X, nin, it, lo, fl = oracle.retriangulate_ex(p, 2)
    # the triangulation unit of the oracle is compiled without fused multiply-adds, so its trace is a property of IEEE arithmetic, not of
    # this machine's compiler: bit-identical points, the same RansacStatistics, the same inlier sets
    assert np.array_equal(nin, g["num_inliers"]) and np.array_equal(X, g["points"])
    assert np.array_equal(it, g["iterations"]) and np.array_equal(lo, g["lo_runs"]) and np.array_equal(fl, g["inlier_flags"])
    assert (it >= 100).all() and (lo >= 1).all()                        # min_num_iterations_; LO at iteration 50 at the latest
    mask = np.zeros(Np, bool); mask[g["corrupted"]] = True
    assert (nin[mask] <= 5).all() and (nin[~mask] == 6).mean() > 0.99
