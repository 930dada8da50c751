"""Oracle chain for the drivers' stage sequence (TEST INFRASTRUCTURE like the rest of oracle/: only tests/, smoke() and bench.py's cpu_baseline leg import it).

The reference's drivers run  Optimize -> Retriangulate -> Optimize -> [unfix translations] -> Optimize -> Normalize -> Retriangulate -> Optimize -> Normalize
(examples/run_spherical_sfm.cpp:93-112, examples/run_spherical_sfm_uncalib.cpp:176-222).  `replay` feeds every stage of a demo_circle dump through the CPU
restatement FROM THE STATE THE GPU LEFT BEFORE IT (identical input bits, so Retriangulate's trace must agree exactly), `chain` runs the same sequence on the
oracle's own outputs from the start.  Normalize / Apply / Pose are restated here from src/sfm.cpp:341-381,535-571 and src/sfm_types.cpp:8-51 on top of the
oracle's so3exp / so3ln (src/so3.cpp)."""
import struct

import numpy as np

from spherical_sfm_amd import synth

OPT, RETRI, NORM, UNFIX = 0, 1, 2, 3


def read_dump(path):
    b = open(path, "rb").read()
    Nc, Np, M, ns = struct.unpack_from("4i", b, 0); off = 16
    rec = np.frombuffer(b, dtype=np.dtype([("c", "<i4"), ("p", "<i4"), ("x", "<f8"), ("y", "<f8")]), count=M, offset=off); off += M * 24
    order = np.lexsort((rec["c"], rec["p"]))                      # point-major, cameras ascending (the order SfM's maps iterate in)
    rec = rec[order]

    def state(o):
        cams = np.frombuffer(b, "<f8", Nc * 6, o).reshape(Nc, 6).copy(); o += Nc * 48
        pts = np.frombuffer(b, "<f8", Np * 3, o).reshape(Np, 3).copy(); o += Np * 24
        f = float(np.frombuffer(b, "<f8", 1, o)[0]); o += 8
        return (cams, pts, f), o
    states = []; stages = []
    s, off = state(off); states.append(s)
    for _ in range(ns):
        kind, it, ok, _pad = struct.unpack_from("4i", b, off); off += 16
        cost = struct.unpack_from("d", b, off)[0]; off += 8
        s, off = state(off); states.append(s); stages.append(dict(kind=kind, iterations=it, ok=ok, cost=cost))
    assert off == len(b)
    obs = dict(cam=rec["c"].astype(np.int32), pt=rec["p"].astype(np.int32), xy=np.ascontiguousarray(np.stack([rec["x"], rec["y"]], 1)))
    return dict(Nc=Nc, Np=Np, M=M, obs=obs, states=states, stages=stages)


def problem(d, state, general, focal_fixed):
    cams, pts, f = state; Nc, Np = d["Nc"], d["Np"]
    rf = np.zeros(Nc, np.uint8); rf[0] = 1
    tf = np.ones(Nc, np.uint8)
    if general: tf[1:] = 0                                        # run_spherical_sfm.cpp:101-104
    return synth.BAProblem(cameras=cams.copy(), points=pts.copy(), focal=f, obs_xy=d["obs"]["xy"], obs_cam=d["obs"]["cam"], obs_pt=d["obs"]["pt"], rot_fixed=rf,
                           trans_fixed=tf, pt_fixed=np.zeros(Np, np.uint8), focal_fixed=focal_fixed, gt_cameras=cams, gt_points=pts, gt_focal=0.0)


def _exp_all(O, r): return np.stack([O.so3exp(x) for x in r])
def _ln_all(O, R): return np.stack([O.so3ln(x) for x in R])


def normalize(O, cams, pts, inward=False):
    """SfM::Normalize (src/sfm.cpp:535-571) through Apply(Pose) / Apply(double) (:341-381) and Pose (src/sfm_types.cpp)."""
    cams = cams.copy(); pts = pts.copy()
    nz = (pts * pts).sum(1) != 0                                   # points at the origin are "removed" and stay there (:353,373)
    R = _exp_all(O, cams[:, 3:])                                    # GetPose: P = [so3exp(r) | t]
    centres = -np.einsum('nji,nj->ni', R, cams[:, :3])             # getCenter = R^T (-t)
    centroid = centres.sum(0) / len(cams)
    # Apply(Pose(-centroid, 0)): every camera is post-multiplied by the inverse pose [I | centroid]; r is re-derived from the product (postMultiply)
    cams[:, :3] = np.einsum('nij,j->ni', R, centroid) + cams[:, :3]
    cams[:, 3:] = _ln_all(O, R)
    pts[nz] = pts[nz] - centroid
    R = _exp_all(O, cams[:, 3:])
    centres = -np.einsum('nji,nj->ni', R, cams[:, :3])
    avg = np.linalg.norm(centres, axis=1).sum() / len(cams)
    s = 1.0 / avg
    cams[:, :3] *= s; pts[nz] *= s                                 # Apply(1 / avg_scale)
    if (inward and cams[0, 2] < 0) or (not inward and cams[0, 2] > 0):
        cams[:, :3] *= -1.0; pts[nz] *= -1.0                       # Apply(-1)
    return cams, pts


def run_stage(O, d, state, kind, general, focal_fixed):
    """One stage on the CPU restatement -> (state after, info dict)."""
    cams, pts, f = state
    if kind == OPT:
        c, p, fo, s = O.ba_solve(problem(d, state, general, focal_fixed))
        return (c, p, fo), dict(iterations=s["iterations"], cost=s["final_cost"], termination=s["termination"])
    if kind == RETRI:
        X, nin, it, lo, fl = O.retriangulate_ex(problem(d, state, general, focal_fixed), 16)
        return (cams.copy(), X, f), dict(num_inliers=nin, iterations=it, lo=lo)
    if kind == NORM:
        c, p = normalize(O, cams, pts)
        return (c, p, f), {}
    return (cams.copy(), pts.copy(), f), {}


def compare_states(a, b):
    """-> dict of max relative differences between two states (cameras, nonzero points, focal) and the symmetric difference of their zero sets"""
    (ca, pa, fa), (cb, pb, fb) = a, b
    za = ~pa.any(1); zb = ~pb.any(1)
    both = ~za & ~zb
    dp = (np.linalg.norm(pa[both] - pb[both], axis=1) / np.linalg.norm(pb[both], axis=1)) if both.any() else np.zeros(1)
    return dict(cam=float(np.abs(ca - cb).max() / max(np.abs(cb).max(), 1e-300)), pt_max=float(dp.max()), pt_q999=float(np.quantile(dp, 0.999)),
                focal=abs(fa - fb) / abs(fb), zero_diff=int((za != zb).sum()), zeros=int(zb.sum()))
