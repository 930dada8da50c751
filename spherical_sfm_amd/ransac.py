"""Python host side of the batched spherical relative-pose RANSAC (include/ssfm.h: ssfm_ransac_batch).
Mirrors what estimate_pairwise (examples/spherical_sfm_tools.cpp:309-431) does per image pair."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import RansacOptionsC, c_double_p, c_i32_p, c_u8_p, c_u32_p, RANSAC_FIXED_BUDGET, RANSAC_REFERENCE_TRACE  # noqa: F401


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
    """pairs: list of (u (n,3), v (n,3)).  -> dict(E (P,3,3), R (P,3,3), inliers [list of bool arrays], num_inliers, scores,
    iterations, lo_runs).  Options by keyword (ssfm_ransac_options fields; mode defaults to the reference-trace LO-MSAC).
    sharded=True: ssfm_ransac_batch_sharded -- every rank of ctx's communicator passes the same list and gets every result."""
    ptr = np.zeros(len(pairs) + 1, np.int32)
    for i, (u, v) in enumerate(pairs):
        ptr[i + 1] = ptr[i] + len(u)
    U = np.ascontiguousarray(np.concatenate([np.asarray(u, np.float64).reshape(-1, 3) for u, _ in pairs]))
    V = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64).reshape(-1, 3) for _, v in pairs]))
    out = estimate_flat(ctx, ptr, U, V, squared_inlier_threshold, options=options, sharded=sharded, **kw)
    mask = out.pop("mask")
    out["inliers"] = [mask[ptr[i]:ptr[i + 1]].astype(bool) for i in range(len(pairs))]
    return out


def estimate_flat(ctx, pair_ptr, U, V, squared_inlier_threshold, options=None, sharded=False, **kw):
    """The C call itself on CSR-style arrays: pair p owns rays [pair_ptr[p], pair_ptr[p+1]) of U, V (total, 3)."""
    ptr = np.ascontiguousarray(pair_ptr, np.int32); U = np.ascontiguousarray(U, np.float64); V = np.ascontiguousarray(V, np.float64)
    o = options or default_options(**kw)
    P = len(ptr) - 1
    E = np.zeros(9 * P); R = np.zeros(9 * P); mask = np.zeros(max(int(ptr[-1]), 1), np.uint8); nin = np.zeros(P, np.int32); sc = np.zeros(P)
    st = np.zeros(2 * P, np.uint32)
    fn = _lib.lib().ssfm_ransac_batch_sharded if sharded else _lib.lib().ssfm_ransac_batch
    _lib.check(fn(ctx._p, P, ptr.ctypes.data_as(c_i32_p), U.ctypes.data_as(c_double_p), V.ctypes.data_as(c_double_p),
                  squared_inlier_threshold, C.byref(o), E.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p),
                  mask.ctypes.data_as(c_u8_p), nin.ctypes.data_as(c_i32_p), sc.ctypes.data_as(c_double_p), st.ctypes.data_as(c_u32_p)), ctx._p)
    return dict(E=_unflat(E), R=_unflat(R), mask=mask[:int(ptr[-1])], num_inliers=nin, scores=sc, iterations=st[0::2].copy(), lo_runs=st[1::2].copy())


def estimate_indexed(ctx, feat_ptr, feat_rays, pair_frame0, pair_frame1, match_ptr, match_idx0, match_idx1, squared_inlier_threshold, options=None, sharded=False, **kw):
    """ssfm_ransac_batch_indexed: per-frame feature rays (feat_rays[feat_ptr[f] + k]) and per-pair match lists instead of materialised ray pairs.
    sharded=True: ssfm_ransac_batch_indexed_sharded (pairs round robin over the ranks of the context's communicator, every rank gets every result)."""
    fp = np.ascontiguousarray(feat_ptr, np.int32); fr = np.ascontiguousarray(feat_rays, np.float64)
    f0 = np.ascontiguousarray(pair_frame0, np.int32); f1 = np.ascontiguousarray(pair_frame1, np.int32)
    mp = np.ascontiguousarray(match_ptr, np.int32); m0 = np.ascontiguousarray(match_idx0, np.int32); m1 = np.ascontiguousarray(match_idx1, np.int32)
    o = options or default_options(**kw)
    P = len(mp) - 1
    E = np.zeros(9 * P); R = np.zeros(9 * P); mask = np.zeros(max(int(mp[-1]), 1), np.uint8); nin = np.zeros(P, np.int32); sc = np.zeros(P)
    st = np.zeros(2 * P, np.uint32)
    fn = _lib.lib().ssfm_ransac_batch_indexed_sharded if sharded else _lib.lib().ssfm_ransac_batch_indexed
    _lib.check(fn(ctx._p, len(fp) - 1, fp.ctypes.data_as(c_i32_p), fr.ctypes.data_as(c_double_p), P,
                                                   f0.ctypes.data_as(c_i32_p), f1.ctypes.data_as(c_i32_p), mp.ctypes.data_as(c_i32_p),
                                                   m0.ctypes.data_as(c_i32_p), m1.ctypes.data_as(c_i32_p), C.c_double(squared_inlier_threshold), C.byref(o),
                                                   E.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p), mask.ctypes.data_as(c_u8_p),
                                                   nin.ctypes.data_as(c_i32_p), sc.ctypes.data_as(c_double_p), st.ctypes.data_as(c_u32_p)), ctx._p)
    return dict(E=_unflat(E), R=_unflat(R), mask=mask[:int(mp[-1])], num_inliers=nin, scores=sc, iterations=st[0::2].copy(), lo_runs=st[1::2].copy())


def last_kernel_ms(ctx):
    """device time of the kernels of the context's last estimate_* call (ssfm_ransac_last_kernel_ms)"""
    ms = C.c_double(0)
    _lib.check(_lib.lib().ssfm_ransac_last_kernel_ms(ctx._p, C.byref(ms)), ctx._p)
    return ms.value


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


def _csr(lists):
    ptr = np.zeros(len(lists) + 1, np.int32)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32).reshape(-1) for l in lists]) if len(lists) and ptr[-1] else np.zeros(1, np.int32), np.int32)
    return ptr, flat


def sampson_refine_probe(ctx, u, v, lists, Es, inward=False):
    """SphericalEstimator::LeastSquares on the device: task t refines Es[t] (3,3) on the rays lists[t] of the pair (u, v)."""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    ptr, flat = _csr(lists)
    E = np.ascontiguousarray(np.transpose(np.asarray(Es, np.float64).reshape(-1, 3, 3), (0, 2, 1))).reshape(-1).copy()
    _lib.check(_lib.lib().ssfm_sampson_refine_probe(ctx._p, len(u), u.ctypes.data_as(c_double_p), v.ctypes.data_as(c_double_p), len(lists),
                                                    ptr.ctypes.data_as(c_i32_p), flat.ctypes.data_as(c_i32_p), int(inward), E.ctypes.data_as(c_double_p)), ctx._p)
    return _unflat(E)


def sampson_refine_probe_ex(ctx, u, v, lists, Es, inward=False, wave=False):
    """The same fit with its trace.  -> (E (T,3,3), x (T,6) = [r1; t1], iterations (T,), status (T,), initial cost (T,), final cost (T,)).
    wave=True runs the one-wave form the batched LO-MSAC kernel uses instead of the workgroup-cooperative one."""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    ptr, flat = _csr(lists)
    E = np.ascontiguousarray(np.transpose(np.asarray(Es, np.float64).reshape(-1, 3, 3), (0, 2, 1))).reshape(-1).copy()
    tr = np.zeros(10 * len(lists))
    _lib.check(_lib.lib().ssfm_sampson_refine_probe_ex(ctx._p, len(u), u.ctypes.data_as(c_double_p), v.ctypes.data_as(c_double_p), len(lists),
                                                       ptr.ctypes.data_as(c_i32_p), flat.ctypes.data_as(c_i32_p), int(inward), int(wave),
                                                       E.ctypes.data_as(c_double_p), tr.ctypes.data_as(c_double_p)), ctx._p)
    tr = tr.reshape(-1, 10)
    return _unflat(E), tr[:, :6].copy(), tr[:, 6].astype(int), tr[:, 7].astype(int), tr[:, 8].copy(), tr[:, 9].copy()


def decompose_probe(ctx, Es, inward=False):
    """decompose_spherical_essential_matrix + so3exp on the device -> (r (T,3), R (T,3,3))"""
    E = np.ascontiguousarray(np.transpose(np.asarray(Es, np.float64).reshape(-1, 3, 3), (0, 2, 1))).reshape(-1).copy()
    T = len(E) // 9; r = np.zeros(3 * T); R = np.zeros(9 * T)
    _lib.check(_lib.lib().ssfm_decompose_probe(ctx._p, T, E.ctypes.data_as(c_double_p), int(inward), r.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p)), ctx._p)
    return r.reshape(T, 3), _unflat(R)


def nonminimal_probe(ctx, u, v, samples):
    """SphericalEstimator::NonMinimalSolver on samples of 3..9 rays -> (ok (T,), E (T,3,3))"""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    ptr, flat = _csr(samples); T = len(samples)
    E = np.zeros(9 * T); ok = np.zeros(T, np.int32)
    _lib.check(_lib.lib().ssfm_nonminimal_probe(ctx._p, len(u), u.ctypes.data_as(c_double_p), v.ctypes.data_as(c_double_p), T, ptr.ctypes.data_as(c_i32_p),
                                                flat.ctypes.data_as(c_i32_p), E.ctypes.data_as(c_double_p), ok.ctypes.data_as(c_i32_p)), ctx._p)
    return ok, _unflat(E)


def so3_probe(ctx, what, x):
    """what: 'exp' | 'ln' | 'aa2R' | 'R2aa' -- the device's so3exp / so3ln / AngleAxisToRotationMatrix / RotationMatrixToAngleAxis."""
    code = {"exp": 0, "ln": 1, "aa2R": 2, "R2aa": 3}[what]
    if code in (0, 2):
        a = np.ascontiguousarray(np.asarray(x, np.float64).reshape(-1, 3)); n = len(a); out = np.zeros(9 * n)
        _lib.check(_lib.lib().ssfm_so3_probe(ctx._p, code, n, a.ctypes.data_as(c_double_p), out.ctypes.data_as(c_double_p)), ctx._p)
        return _unflat(out)
    a = np.ascontiguousarray(np.transpose(np.asarray(x, np.float64).reshape(-1, 3, 3), (0, 2, 1))).reshape(-1).copy(); n = len(a) // 9; out = np.zeros(3 * n)
    _lib.check(_lib.lib().ssfm_so3_probe(ctx._p, code, n, a.ctypes.data_as(c_double_p), out.ctypes.data_as(c_double_p)), ctx._p)
    return out.reshape(n, 3)


def mt19937_probe(ctx, seed, lo, hi, nraw=0):
    """(raw words, uniform_int_distribution<int>(lo[i], hi[i]) draws) of std::mt19937(seed) as the device restates them"""
    lo = np.ascontiguousarray(lo, np.int32); hi = np.ascontiguousarray(hi, np.int32)
    out = np.zeros(max(len(lo), 1), np.int32); raw = np.zeros(max(nraw, 1), np.uint32)
    _lib.check(_lib.lib().ssfm_mt19937_probe(ctx._p, seed, len(lo), lo.ctypes.data_as(c_i32_p), hi.ctypes.data_as(c_i32_p), out.ctypes.data_as(c_i32_p),
                                             nraw, raw.ctypes.data_as(c_u32_p)), ctx._p)
    return raw[:nraw], out[:len(lo)]
