"""Synthetic "circle" problems.

The reference has no problem generator for the BA path (`examples/make_circle_views.cpp` is an
image-based view synthesiser), so the workload is defined here.  Geometry follows the
reference's own conventions:

* cameras on the unit sphere looking outward, `t = (0,0,-1)`, `R_i = so3exp((0, 2*pi*i/Nc, 0))`
  (`examples/spherical_sfm_tools.cpp:879-881`, `evaluation/problem_generator/problem_generator.cpp:27`);
* observations are principal-point-centred pixels (`examples/spherical_sfm_tools.cpp:907-908`);
* depth range as `problem_generator.cpp:45-46` (outward: 4..8).

Everything is float64 / int32 numpy; deterministic for a given seed (numpy PCG64).
"""
from dataclasses import dataclass
import numpy as np


def so3exp(r):
    """Rodrigues, reference src/so3.cpp:16-23.  r: (...,3) -> (...,3,3)."""
    r = np.asarray(r, dtype=np.float64)
    th = np.linalg.norm(r, axis=-1)[..., None, None]
    safe = np.where(th < 1e-10, 1.0, th)
    k = r[..., None, :] / safe  # (...,1,3)
    K = np.zeros(r.shape[:-1] + (3, 3))
    K[..., 0, 1] = -k[..., 0, 2]; K[..., 0, 2] = k[..., 0, 1]
    K[..., 1, 0] = k[..., 0, 2]; K[..., 1, 2] = -k[..., 0, 0]
    K[..., 2, 0] = -k[..., 0, 1]; K[..., 2, 1] = k[..., 0, 0]
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    return np.where(th < 1e-10, np.eye(3), R)


@dataclass
class BAProblem:
    cameras: np.ndarray      # (Nc,6) [t;r]  initial state
    points: np.ndarray       # (Np,3)        initial state
    focal: float             # initial focal
    obs_xy: np.ndarray       # (M,2)
    obs_cam: np.ndarray      # (M,) int32
    obs_pt: np.ndarray       # (M,) int32
    rot_fixed: np.ndarray    # (Nc,) uint8
    trans_fixed: np.ndarray  # (Nc,) uint8
    pt_fixed: np.ndarray     # (Np,) uint8
    focal_fixed: bool
    gt_cameras: np.ndarray
    gt_points: np.ndarray
    gt_focal: float

    def copy(self):
        return BAProblem(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.__dict__.items()})


def make_circle(num_cameras=60, num_points=20000, obs_per_point=6, *, spherical=True, focal_fixed=True,
                seed=1234, focal=1000.0, pixel_noise=0.5, rot_noise_deg=0.5, point_noise=0.01,
                focal_init_factor=1.1, trans_noise=0.0, check_in_frame=True, xy_range=None):
    """Outward-facing circle.  M = obs_per_point * num_points exactly.

    spherical=True  -> every translation fixed (examples/spherical_sfm_tools.cpp:883), camera 0 rotation fixed.
    spherical=False -> only camera 0 fixed (general BA, run_spherical_sfm_uncalib.cpp:197-201).
    """
    Nc, Np, K = int(num_cameras), int(num_points), int(obs_per_point)
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * np.arange(Nc) / Nc
    r_gt = np.stack([np.zeros(Nc), ang, np.zeros(Nc)], axis=1)
    # keep angle-axis in (-pi, pi] so that so3ln round trips are unambiguous
    r_gt[:, 1] = np.where(r_gt[:, 1] > np.pi, r_gt[:, 1] - 2 * np.pi, r_gt[:, 1])
    t_gt = np.tile(np.array([0.0, 0.0, -1.0]), (Nc, 1))
    R = so3exp(r_gt)                                      # (Nc,3,3)
    anchor = (np.arange(Np, dtype=np.int64) * Nc // Np).astype(np.int64)
    stride = max(1, int(round(Nc / (15.0 * (K - 1))))) if K > 1 else 1
    span_deg = (K - 1) * stride * 360.0 / Nc
    if xy_range is None:                                   # SURVEY 8d quotes 0.45 for the ~24 degree span of Nc >= 75
        xy_range = 0.45 if span_deg <= 24.5 else 0.30      # small rings (config 1: 30 degrees) need a narrower anchor window
    xy = rng.uniform(-xy_range, xy_range, size=(Np, 2))
    depth = rng.uniform(4.0, 8.0, size=Np)
    pc = np.concatenate([xy, np.ones((Np, 1))], axis=1) * depth[:, None]
    X = np.einsum('nji,nj->ni', R[anchor], pc - t_gt[anchor])   # R_a^T (p - t)
    offs = stride * (np.arange(K) - K // 2)
    cam = (anchor[:, None] + offs[None, :]) % Nc                # (Np,K)
    cam = np.sort(cam, axis=1)                                  # point-major, camera ascending (std::map order)
    Xc = np.einsum('nkij,nj->nki', R[cam], X) + t_gt[cam]
    assert (Xc[..., 2] > 0.5).all(), "generator: a point fell behind a camera"
    proj = focal * Xc[..., :2] / Xc[..., 2:3]
    if check_in_frame:
        assert (np.abs(proj[..., 0]) < 960).all() and (np.abs(proj[..., 1]) < 540).all(), "projection outside 1920x1080"
    obs = proj + rng.normal(0.0, pixel_noise, size=proj.shape)
    cams0 = np.concatenate([t_gt, r_gt], axis=1).copy()
    noise_r = rng.normal(0.0, np.deg2rad(rot_noise_deg), size=(Nc, 3)); noise_r[0] = 0
    cams0[:, 3:] += noise_r
    if not spherical and trans_noise > 0:
        nt = rng.normal(0.0, trans_noise, size=(Nc, 3)); nt[0] = 0
        cams0[:, :3] += nt
    pts0 = X * (1.0 + rng.normal(0.0, point_noise, size=(Np, 1)))
    rot_fixed = np.zeros(Nc, np.uint8); rot_fixed[0] = 1
    trans_fixed = np.ones(Nc, np.uint8) if spherical else np.zeros(Nc, np.uint8)
    trans_fixed[0] = 1
    return BAProblem(cameras=cams0, points=pts0, focal=float(focal if focal_fixed else focal * focal_init_factor),
                     obs_xy=np.ascontiguousarray(obs.reshape(-1, 2)),
                     obs_cam=np.ascontiguousarray(cam.reshape(-1).astype(np.int32)),
                     obs_pt=np.repeat(np.arange(Np, dtype=np.int32), K),
                     rot_fixed=rot_fixed, trans_fixed=trans_fixed, pt_fixed=np.zeros(Np, np.uint8),
                     focal_fixed=bool(focal_fixed), gt_cameras=np.concatenate([t_gt, r_gt], axis=1), gt_points=X,
                     gt_focal=float(focal))


def make_ragged_circle(num_cameras=300, num_observations=600000, min_len=3, max_len=14, *, spherical=False, focal_fixed=True,
                       seed=4321, focal=1000.0, pixel_noise=0.5, rot_noise_deg=0.5, point_noise=0.01, focal_init_factor=1.1, shuffle_within_frame=True):
    """Video-like tracks on the outward circle: a point is seen by `L` CONSECUTIVE cameras f0 .. f0+L-1 (mod Nc), L uniform in [min_len, max_len], f0 uniform --
    ragged track lengths, and point ids issued the way build_sfm issues them (examples/spherical_sfm_tools.cpp:862-955: a track gets its id when its first
    match is met, so ids ascend with the first frame and tracks of different length that start in the same frame are interleaved).  About
    `num_observations` observations in total (the last track is cut so that M is exact when possible)."""
    Nc = int(num_cameras)
    rng = np.random.default_rng(seed)
    mean_len = 0.5 * (min_len + max_len)
    Np = int(np.ceil(num_observations / mean_len * 1.02)) + 8
    L = rng.integers(min_len, max_len + 1, size=Np)
    keep = np.searchsorted(np.cumsum(L), num_observations, side='left') + 1
    L = L[:keep]; Np = len(L)
    over = int(L.sum() - num_observations)
    if over > 0 and L[-1] - over >= 3: L[-1] -= over
    f0 = np.sort(rng.integers(0, Nc, size=Np))                    # ids ascend with the first frame
    if shuffle_within_frame:                                      # ... and lengths are interleaved inside a frame
        L = L[np.lexsort((rng.random(Np), f0))]
    ang = 2 * np.pi * np.arange(Nc) / Nc
    r_gt = np.stack([np.zeros(Nc), ang, np.zeros(Nc)], axis=1)
    r_gt[:, 1] = np.where(r_gt[:, 1] > np.pi, r_gt[:, 1] - 2 * np.pi, r_gt[:, 1])
    t_gt = np.tile(np.array([0.0, 0.0, -1.0]), (Nc, 1))
    R = so3exp(r_gt)
    # anchor = the middle camera of the window; narrow image window so that every projection of the (at most 14-frame, 16.8 degree) span stays in the frame
    mid = (f0 + L // 2) % Nc
    xy = rng.uniform(-0.30, 0.30, size=(Np, 2))
    depth = rng.uniform(4.0, 8.0, size=Np)
    pc = np.concatenate([xy, np.ones((Np, 1))], axis=1) * depth[:, None]
    X = np.einsum('nji,nj->ni', R[mid], pc - t_gt[mid])
    pt = np.repeat(np.arange(Np, dtype=np.int64), L)
    k = np.arange(int(L.sum()), dtype=np.int64) - np.repeat(np.cumsum(L) - L, L)
    cam = (f0[pt] + k) % Nc
    # point-major, cameras ascending (std::map order)
    order = np.lexsort((cam, pt)); pt = pt[order]; cam = cam[order]
    Xc = np.einsum('nij,nj->ni', R[cam], X[pt]) + t_gt[cam]
    assert (Xc[:, 2] > 0.5).all(), "generator: a point fell behind a camera"
    proj = focal * Xc[:, :2] / Xc[:, 2:3]
    assert (np.abs(proj[:, 0]) < 960).all() and (np.abs(proj[:, 1]) < 540).all(), "projection outside 1920x1080"
    obs = proj + rng.normal(0.0, pixel_noise, size=proj.shape)
    cams0 = np.concatenate([t_gt, r_gt], axis=1).copy()
    noise_r = rng.normal(0.0, np.deg2rad(rot_noise_deg), size=(Nc, 3)); noise_r[0] = 0
    cams0[:, 3:] += noise_r
    pts0 = X * (1.0 + rng.normal(0.0, point_noise, size=(Np, 1)))
    rot_fixed = np.zeros(Nc, np.uint8); rot_fixed[0] = 1
    trans_fixed = np.ones(Nc, np.uint8) if spherical else np.zeros(Nc, np.uint8)
    trans_fixed[0] = 1
    return BAProblem(cameras=cams0, points=pts0, focal=float(focal if focal_fixed else focal * focal_init_factor),
                     obs_xy=np.ascontiguousarray(obs), obs_cam=np.ascontiguousarray(cam.astype(np.int32)), obs_pt=np.ascontiguousarray(pt.astype(np.int32)),
                     rot_fixed=rot_fixed, trans_fixed=trans_fixed, pt_fixed=np.zeros(Np, np.uint8), focal_fixed=bool(focal_fixed),
                     gt_cameras=np.concatenate([t_gt, r_gt], axis=1), gt_points=X, gt_focal=float(focal))


def make_rotation_graph(num_cameras=300, max_offset=8, *, seed=1234, noise_deg=0.2, outlier_frac=0.02):
    """Pose graph of SURVEY.md 8d: edges (i, i+d mod Nc), d=1..max_offset, R_rel = R_j R_i^T with noise,
    a fraction of gross outliers.  Returns (R_init (Nc,3,3), index0, index1, R_rel (E,3,3), R_gt)."""
    Nc = int(num_cameras)
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * np.arange(Nc) / Nc
    r_gt = np.stack([np.zeros(Nc), ang, np.zeros(Nc)], axis=1)
    r_gt[:, 1] = np.where(r_gt[:, 1] > np.pi, r_gt[:, 1] - 2 * np.pi, r_gt[:, 1])
    R_gt = so3exp(r_gt)
    i0 = np.repeat(np.arange(Nc), max_offset)
    i1 = (i0 + np.tile(np.arange(1, max_offset + 1), Nc)) % Nc
    keep = i0 != i1
    i0, i1 = i0[keep], i1[keep]
    E = len(i0)
    noise = so3exp(rng.normal(0.0, np.deg2rad(noise_deg), size=(E, 3)))
    R_rel = noise @ R_gt[i1] @ np.transpose(R_gt[i0], (0, 2, 1))
    n_out = int(round(outlier_frac * E))
    if n_out:
        which = rng.choice(E, n_out, replace=False)
        axis = rng.normal(size=(n_out, 3)); axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        R_rel[which] = so3exp(axis * rng.uniform(0.5, 2.5, size=(n_out, 1)))
    # sequential initialisation as initialize_rotations_sequential (examples/spherical_sfm_tools.cpp:794-813)
    R_init = np.tile(np.eye(3), (Nc, 1, 1))
    first = {(int(a), int(b)): k for k, (a, b) in reversed(list(enumerate(zip(i0, i1))))}
    Rc = np.eye(3)
    for idx in range(1, Nc):
        k = first.get((idx - 1, idx))
        if k is not None:
            Rc = R_rel[k] @ Rc
            R_init[idx] = Rc
    return R_init, i0.astype(np.int32), i1.astype(np.int32), R_rel, R_gt


def make_relative_pose_problem(num_corr=100, *, inward=False, rotation_deg=None, noise=0.0, outlier_frac=0.0, seed=0):
    """Synthetic spherical pair with the geometry of the reference's generator
    (evaluation/problem_generator/problem_generator.cpp:14-65): t = R e_z - e_z (negated if inward), points at
    depth U(4,8) outward / U(0.25,0.75) inward in front of camera 1, image-plane noise `noise` (= sigma_px / focal);
    a fraction of correspondences replaced by uniform outliers.  Returns u (n,3), v (n,3), R_gt, E_gt, inlier mask."""
    rng = np.random.default_rng(seed)
    while True:
        angle = rng.uniform(0, np.pi) if rotation_deg is None else np.deg2rad(rotation_deg)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        R = so3exp(ax * angle)
        t = R[:, 2] - np.array([0.0, 0.0, 1.0])
        if inward:
            t = -t
        u = np.concatenate([rng.normal(size=(num_corr, 2)), np.ones((num_corr, 1))], axis=1)
        depth = rng.uniform(0.25, 0.75, num_corr) if inward else rng.uniform(4.0, 8.0, num_corr)
        X = u * depth[:, None]
        P2 = X @ R.T + t
        if (P2[:, 2] <= 0).any():
            continue
        v = np.concatenate([P2[:, :2] / P2[:, 2:3], np.ones((num_corr, 1))], axis=1)
        break
    u[:, :2] += noise * rng.normal(size=(num_corr, 2)); v[:, :2] += noise * rng.normal(size=(num_corr, 2))
    inl = np.ones(num_corr, bool)
    n_out = int(round(outlier_frac * num_corr))
    if n_out:
        idx = rng.choice(num_corr, n_out, replace=False)
        v[idx, :2] = rng.uniform(-1.5, 1.5, size=(n_out, 2)); inl[idx] = False
    S = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return np.ascontiguousarray(u), np.ascontiguousarray(v), R, S @ R, inl


def corrupt_observations(prob, frac=0.1, seed=5, lo=30.0, hi=80.0):
    """Gross outliers for Retriangulate tests: one observation of `frac` of the points is displaced by lo..hi px.
    Returns the ids of the touched points; prob.obs_xy is replaced."""
    rng = np.random.default_rng(seed)
    xy = np.array(prob.obs_xy, np.float64, copy=True)
    pts = rng.choice(len(prob.points), int(frac * len(prob.points)), replace=False)
    order = np.argsort(prob.obs_pt, kind="stable")
    start = np.searchsorted(prob.obs_pt[order], np.arange(len(prob.points) + 1))
    for p in pts:
        i = order[rng.integers(start[p], start[p + 1])]
        ang = rng.uniform(0, 2 * np.pi); r = rng.uniform(lo, hi)
        xy[i] += r * np.array([np.cos(ang), np.sin(ang)])
    prob.obs_xy = xy
    return pts



def make_circle_pairs(num_frames=200, num_points=3000, *, focal=600.0, half_fov_tan=0.8, noise_px=0.5, outlier_frac=0.2, seed=7):
    """Exhaustive pair list of an outward-facing camera circle -- the input estimate_pairwise loops over
    (examples/spherical_sfm_tools.cpp:320-332: every (index0 < index1)).  Frame i sits on the unit sphere with
    x_cam = R_i X + t, t = (0,0,-1), R_i = so3exp((0, 2 pi i / N, 0)) (the reference's spherical camera model,
    tools.cpp:879-881); scene points on a ring at depth U(4,8).  Frames see the points inside their field of view,
    so neighbouring frames share many points and opposite frames none: most far pairs have no correspondence at all.
    Matches of a pair = the shared points (pixel noise noise_px / focal on both rays) + outlier_frac wrong associations.
    Returns pair_ptr (P+1,), U (total,3), V (total,3), pairs (P,2) frame indices, R_rel (P,3,3) ground truth R_j R_i^T."""
    rng = np.random.default_rng(seed)
    N = num_frames
    ang = 2.0 * np.pi * np.arange(N) / N
    Rs = np.stack([so3exp(np.array([0.0, a, 0.0])) for a in ang])
    t = np.array([0.0, 0.0, -1.0])
    phi = rng.uniform(0, 2 * np.pi, num_points); d = rng.uniform(4.0, 8.0, num_points); h = rng.uniform(-2.5, 2.5, num_points)
    X = np.stack([d * np.sin(phi), h, d * np.cos(phi)], axis=1)
    vis, proj = [], []
    for i in range(N):
        pc = X @ Rs[i].T + t
        ok = (pc[:, 2] > 0.5) & (np.abs(pc[:, 0]) < half_fov_tan * pc[:, 2]) & (np.abs(pc[:, 1]) < half_fov_tan * pc[:, 2])
        ids = np.nonzero(ok)[0]
        xy = pc[ids, :2] / pc[ids, 2:3] + (noise_px / focal) * rng.normal(size=(len(ids), 2))
        vis.append(ids); proj.append(xy)
    ptr = [0]; Us, Vs, pairs, Rrel = [], [], [], []
    for i in range(N):
        for j in range(i + 1, N):
            common, ia, ib = np.intersect1d(vis[i], vis[j], assume_unique=True, return_indices=True)
            n = len(common)
            u = np.concatenate([proj[i][ia], np.ones((n, 1))], axis=1); v = np.concatenate([proj[j][ib], np.ones((n, 1))], axis=1)
            n_out = int(round(outlier_frac * n))
            if n_out:
                sel = rng.choice(n, n_out, replace=False)
                v[sel, :2] = rng.uniform(-half_fov_tan, half_fov_tan, size=(n_out, 2))
            Us.append(u); Vs.append(v); ptr.append(ptr[-1] + n); pairs.append((i, j)); Rrel.append(Rs[j] @ Rs[i].T)
    return (np.asarray(ptr, np.int32), np.ascontiguousarray(np.concatenate(Us)), np.ascontiguousarray(np.concatenate(Vs)),
            np.asarray(pairs, np.int32), np.stack(Rrel))
