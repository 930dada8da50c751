"""Python host side of ssfm_build_tracks: the track assignment of build_sfm (examples/spherical_sfm_tools.cpp:862-950)."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import c_double_p, c_i32_p, c_u8_p, c_i64_p


def build_tracks(features, image_matches, centerx=0.0, centery=0.0, merge=True):
    """features: list (per keyframe) of (n_k,2) pixel arrays.  image_matches: list of (index0, index1, [(f0,f1),...]).
    -> dict(tracks [list of int arrays per keyframe], num_points, alive (bool), obs_cam, obs_pt, obs_xy)."""
    K = len(features)
    fptr = np.zeros(K + 1, np.int32)
    for k, f in enumerate(features):
        fptr[k + 1] = fptr[k] + len(f)
    fxy = np.ascontiguousarray(np.concatenate([np.asarray(f, np.float64).reshape(-1, 2) for f in features])) if fptr[-1] else np.zeros((0, 2))
    S = len(image_matches)
    i0 = np.array([m[0] for m in image_matches], np.int32); i1 = np.array([m[1] for m in image_matches], np.int32)
    mptr = np.zeros(S + 1, np.int32); f0, f1 = [], []
    for s, m in enumerate(image_matches):
        pairs = sorted(dict(m[2]).items())            # std::map<size_t,size_t>: unique first index, ascending
        mptr[s + 1] = mptr[s] + len(pairs); f0 += [a for a, _ in pairs]; f1 += [b for _, b in pairs]
    f0 = np.array(f0, np.int32); f1 = np.array(f1, np.int32)
    npairs = max(1, len(f0))
    tracks = np.zeros(int(fptr[-1]), np.int32); npts = C.c_int32(0); alive = np.zeros(npairs, np.uint8); nobs = C.c_int64(0)
    oc = np.zeros(2 * npairs, np.int32); op = np.zeros(2 * npairs, np.int32); oxy = np.zeros((2 * npairs, 2))
    p = lambda a, t: a.ctypes.data_as(t)
    rc = _lib.lib().ssfm_build_tracks(K, p(fptr, c_i32_p), p(fxy, c_double_p), S, p(i0, c_i32_p), p(i1, c_i32_p), p(mptr, c_i32_p), p(f0, c_i32_p),
                                      p(f1, c_i32_p), centerx, centery, 1 if merge else 0, p(tracks, c_i32_p), C.byref(npts), p(alive, c_u8_p),
                                      C.byref(nobs), p(oc, c_i32_p), p(op, c_i32_p), p(oxy, c_double_p))
    if rc != 0:
        raise _lib.SsfmError(f"ssfm_build_tracks failed ({rc})")
    n = nobs.value
    return dict(tracks=[tracks[fptr[k]:fptr[k + 1]].copy() for k in range(K)], num_points=npts.value, alive=alive[:npts.value].astype(bool),
                obs_cam=oc[:n].copy(), obs_pt=op[:n].copy(), obs_xy=oxy[:n].copy())
