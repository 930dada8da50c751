"""Python host side of the batched spherical relative-pose RANSAC (include/ssfm.h: ssfm_ransac_batch).
Mirrors what estimate_pairwise (examples/spherical_sfm_tools.cpp:309-431) does per image pair."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import RansacOptionsC, c_double_p, c_i32_p, c_u8_p


def default_options(**kw):
    o = RansacOptionsC()
    _lib.lib().ssfm_ransac_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _unflat(a):
    return np.transpose(np.asarray(a).reshape(-1, 3, 3), (0, 2, 1)).copy()


def estimate_pairs(ctx, pairs, squared_inlier_threshold, options=None, sharded=False, **kw):
    """pairs: list of (u (n,3), v (n,3)).  -> dict(E (P,3,3), R (P,3,3), inliers [list of bool arrays], num_inliers, scores).
    sharded=True: ssfm_ransac_batch_sharded -- every rank of ctx's communicator passes the same list and gets every result."""
    ptr = np.zeros(len(pairs) + 1, np.int32)
    for i, (u, v) in enumerate(pairs):
        ptr[i + 1] = ptr[i] + len(u)
    U = np.ascontiguousarray(np.concatenate([np.asarray(u, np.float64) for u, _ in pairs]))
    V = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64) for _, v in pairs]))
    o = options or default_options(**kw)
    P = len(pairs)
    E = np.zeros(9 * P); R = np.zeros(9 * P); mask = np.zeros(int(ptr[-1]), np.uint8); nin = np.zeros(P, np.int32); sc = np.zeros(P)
    fn = _lib.lib().ssfm_ransac_batch_sharded if sharded else _lib.lib().ssfm_ransac_batch
    _lib.check(fn(ctx._p, P, ptr.ctypes.data_as(c_i32_p), U.ctypes.data_as(c_double_p), V.ctypes.data_as(c_double_p),
                                            squared_inlier_threshold, C.byref(o), E.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p),
                                            mask.ctypes.data_as(c_u8_p), nin.ctypes.data_as(c_i32_p), sc.ctypes.data_as(c_double_p)), ctx._p)
    return dict(E=_unflat(E), R=_unflat(R), inliers=[mask[ptr[i]:ptr[i + 1]].astype(bool) for i in range(P)], num_inliers=nin, scores=sc)


def solver_probe(ctx, u, v, samples, poly=False):
    """The minimal solver (action matrix, or the quartic variant with poly=True) on given 3-point samples ->
    list (per sample) of lists of E (3,3)."""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64); s = np.ascontiguousarray(samples, np.int32).reshape(-1, 3)
    S = len(s); Es = np.zeros(36 * S); cnt = np.zeros(S, np.int32)
    fn = _lib.lib().ssfm_spherical_solver_poly_probe if poly else _lib.lib().ssfm_spherical_solver_probe
    _lib.check(fn(ctx._p, len(u), u.ctypes.data_as(c_double_p), v.ctypes.data_as(c_double_p), S,
                                                      s.ctypes.data_as(c_i32_p), Es.ctypes.data_as(c_double_p), cnt.ctypes.data_as(c_i32_p)), ctx._p)
    out = []
    for i in range(S):
        M = _unflat(Es[36 * i:36 * i + 36])
        out.append([M[k] for k in range(cnt[i])])
    return out
