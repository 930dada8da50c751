"""Generates tests/golden/*.npz from the CPU oracle (oracle/), in this repository's container.

The reference holds no golden vectors for this path and cannot be built here, so these fixtures pin
the ORACLE's behaviour (regression + cross-machine determinism) and give the HIP path fixed targets; they do not
pin the reference.  Re-run:  python tests/golden/make_golden.py [--only=ransac,retriangulate,so3,ba,rotavg]
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O          # noqa: E402
from spherical_sfm_amd import synth     # noqa: E402


def so3_cases():
    rng = np.random.default_rng(11)
    angles = [0.0, 1e-12, 1e-9, 1e-4, np.pi / 4, np.pi / 2, 3 * np.pi / 4, np.pi - 1e-6, np.pi - 1e-3, 2.0]
    rs = []
    for a in angles:
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        rs.append(ax * a)
    rs = np.array(rs)
    R = np.array([O.so3exp(r) for r in rs])
    ln = np.array([O.so3ln(Ri) for Ri in R])
    Rc = np.array([O.angle_axis_to_rotation_matrix(r) for r in rs])
    aac = np.array([O.rotation_matrix_to_angle_axis(Ri) for Ri in R])
    pts = rng.normal(size=(len(rs), 3))
    rot = np.array([O.angle_axis_rotate_point(r, p) for r, p in zip(rs, pts)])
    return dict(r=rs, so3exp=R, so3ln=ln, ceres_R=Rc, ceres_aa=aac, pts=pts, rotated=rot)


def ba_case(spherical, focal_fixed):
    p = synth.make_circle(60, 240, 6, spherical=spherical, focal_fixed=focal_fixed, seed=99)
    p.cameras[0, 3:] = 0.0
    cost, res, jac, used = O.ba_evaluate(p)
    cams, pts, f, s = O.ba_solve(p)
    return dict(cameras0=p.cameras, points0=p.points, focal0=p.focal, obs_xy=p.obs_xy, obs_cam=p.obs_cam, obs_pt=p.obs_pt,
                rot_fixed=p.rot_fixed, trans_fixed=p.trans_fixed, pt_fixed=p.pt_fixed, focal_fixed=int(p.focal_fixed),
                cost0=cost, residuals0=res, jacobians0=jac, cameras=cams, points=pts, focal=f,
                iterations=s["iterations"], final_cost=s["final_cost"], initial_cost=s["initial_cost"])


def rot_case():
    R0, i0, i1, Rrel, Rgt = synth.make_rotation_graph(24, 4, seed=5, outlier_frac=0.05)
    R, cost, s = O.optimize_rotations(R0, i0, i1, Rrel)
    c0 = O.get_cost(R0, i0, i1, Rrel)
    R2, f2, cost2, s2 = O.optimize_rotations_and_focal_length(R0, i0, i1, Rrel, 800.0, 400.0, 1600.0)
    edges = []
    rng = np.random.default_rng(3)
    for kind in (0, 1, 2):
        for _ in range(4):
            r0 = rng.normal(size=3) * 0.5; r1 = rng.normal(size=3) * 0.5; Rm = O.so3exp(rng.normal(size=3) * 0.4)
            res, jac = O.rotation_edge(kind, r0, r1, 1.1, Rm, 0.7)
            edges.append(np.concatenate([[kind], r0, r1, Rm.reshape(-1), res, jac.reshape(-1)]))
    return dict(R0=R0, i0=i0, i1=i1, Rrel=Rrel, R=R, cost=cost, cost0=c0, iterations=s["iterations"], R_focal=R2, focal=f2,
                cost_focal=cost2, edges=np.array(edges))


def ransac_case():
    """3-point solvers (both variants), Sampson values, make/decompose round trip, Sampson least squares, LO-MSAC on a
    100-correspondence / 30 % outlier pair (SURVEY 8c list)."""
    rng = np.random.default_rng(17)
    u, v, R, E, inl = synth.make_relative_pose_problem(100, seed=4, noise=1 / 600, outlier_frac=0.3, rotation_deg=12)
    samples = np.array([rng.choice(100, 3, replace=False) for _ in range(12)], np.int32)
    Es_am = np.array([np.array(O.spherical_solver(u, v, s)) for s in samples])
    poly = [O.spherical_solver_poly(u, v, s) for s in samples]
    Es_poly = np.array([np.array(p[0]) for p in poly]); im_poly = np.array([p[1] for p in poly])
    samp = np.array([O.sampson(E, u[i], v[i]) for i in range(100)])
    Rs = synth.so3exp(rng.normal(size=(6, 3)) * 0.5)
    Em = np.array([O.make_spherical_essential_matrix(Ri, False) for Ri in Rs]); Emi = np.array([O.make_spherical_essential_matrix(Ri, True) for Ri in Rs])
    dec = np.array([np.concatenate(O.decompose_spherical_essential_matrix(Ei, False)) for Ei in Em])
    thr = (2 / 600) ** 2
    o = O.ransac_pair(u, v, thr, min_num_inliers=20)
    # SphericalEstimator::LeastSquares with its six free parameters [r1; t1] (src/spherical_estimator.cpp:140-144 leaves t1 free):
    # 8 fits on subsets of the true inliers from perturbed starts; x = [r1; t1] at the end, LM iterations
    good = np.nonzero(inl)[0]
    lsq_lists = np.full((8, 40), -1, np.int32); lsq_start = np.zeros((8, 3, 3)); lsq_E = np.zeros((8, 3, 3)); lsq_x = np.zeros((8, 6)); lsq_it = np.zeros(8, np.int32)
    for k in range(8):
        m = [21, 7, 3, 40][k % 4]
        lst = rng.choice(good, m, replace=False).astype(np.int32); lsq_lists[k, :m] = lst
        lsq_start[k] = O.make_spherical_essential_matrix(synth.so3exp(rng.normal(size=(1, 3)) * 0.02)[0] @ R, False)
        f = O.sampson_least_squares_ex(u, v, lst, lsq_start[k])
        lsq_E[k] = f["E"]; lsq_x[k] = f["x"]; lsq_it[k] = f["iterations"]
    return dict(lsq_lists=lsq_lists, lsq_start=lsq_start, lsq_E=lsq_E, lsq_x=lsq_x, lsq_iterations=lsq_it,
                u=u, v=v, R=R, E=E, inlier_gt=inl, samples=samples, Es_action=Es_am, Es_poly=Es_poly, poly_imag=im_poly, sampson=samp, Rs=Rs, E_outward=Em,
                E_inward=Emi, decomposed=dec, thr=thr, ransac_E=o["E"], ransac_R=o["R"], ransac_inliers=o["inliers"], ransac_num_inliers=o["num_inliers"],
                ransac_iterations=o["iterations"], ransac_score=o["score"])


def retri_case():
    p = synth.make_circle(60, 400, 6, rot_noise_deg=0.0, pixel_noise=0.4, seed=23)
    bad = synth.corrupt_observations(p, 0.1, seed=9)
    # the oracle's triangulation unit is compiled without fused multiply-adds (oracle/Makefile): the trace depends on last bits
    X, nin, it, lo, fl = O.retriangulate_ex(p, 4)
    return dict(cameras=p.cameras, focal=p.focal, obs_xy=p.obs_xy, obs_cam=p.obs_cam, obs_pt=p.obs_pt, corrupted=bad, points=X, num_inliers=nin,
                iterations=it, lo_runs=lo, inlier_flags=fl)


if __name__ == "__main__":
    import sys
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    want = lambda name: not only or name in only[0]
    if want("ransac"):
        np.savez_compressed(os.path.join(HERE, "ransac.npz"), **ransac_case())
    if want("retriangulate"):
        np.savez_compressed(os.path.join(HERE, "retriangulate.npz"), **retri_case())
    if want("so3"):
        np.savez_compressed(os.path.join(HERE, "so3.npz"), **so3_cases())
    if want("ba"):
        for sph in (True, False):
            for ff in (True, False):
                np.savez_compressed(os.path.join(HERE, f"ba_s{int(sph)}_f{int(ff)}.npz"), **ba_case(sph, ff))
    if want("rotavg"):
        np.savez_compressed(os.path.join(HERE, "rotavg.npz"), **rot_case())
