"""Python host side of the SO(3) pose-graph path (include/ssfm.h: ssfm_rotavg_*, ssfm_posegraph_focal_solve).

Function names and argument meaning follow the reference: `optimize_rotations` (src/rotation_averaging.cpp:44),
`get_cost`, `optimize_rotations_and_focal_length` (src/uncalibrated_pose_graph.cpp:116,147).  Rotations are (n,3,3)
arrays indexed R[i,j]; the column-major flattening the C ABI wants is done here."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import BAOptionsC, BASummaryC, c_double_p, c_i32_p


def default_options(**kw):
    o = BAOptionsC()
    _lib.lib().ssfm_rotavg_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _cm(Rs):
    return np.ascontiguousarray(np.transpose(np.asarray(Rs, np.float64), (0, 2, 1))).reshape(-1).copy()


def _edges(i0, i1, Rrel):
    return np.ascontiguousarray(i0, np.int32), np.ascontiguousarray(i1, np.int32), _cm(Rrel)


def optimize_rotations(ctx, rotations, index0, index1, rel_rotations, options=None, **kw):
    """-> (rotations (n,3,3), final_cost, summary)"""
    buf = _cm(rotations); i0, i1, rel = _edges(index0, index1, rel_rotations)
    o = options or default_options(**kw); s = BASummaryC()
    _lib.check(_lib.lib().ssfm_rotavg_solve(ctx._p, len(rotations), buf.ctypes.data_as(c_double_p), len(i0), i0.ctypes.data_as(c_i32_p),
                                            i1.ctypes.data_as(c_i32_p), rel.ctypes.data_as(c_double_p), C.byref(o), C.byref(s)), ctx._p)
    return np.transpose(buf.reshape(-1, 3, 3), (0, 2, 1)).copy(), s.final_cost, s.as_dict()


def get_cost(ctx, rotations, index0, index1, rel_rotations):
    buf = _cm(rotations); i0, i1, rel = _edges(index0, index1, rel_rotations)
    c = C.c_double(0)
    _lib.check(_lib.lib().ssfm_rotavg_cost(ctx._p, len(rotations), buf.ctypes.data_as(c_double_p), len(i0), i0.ctypes.data_as(c_i32_p),
                                           i1.ctypes.data_as(c_i32_p), rel.ctypes.data_as(c_double_p), C.byref(c)), ctx._p)
    return c.value


def optimize_rotations_and_focal_length(ctx, rotations, index0, index1, rel_rotations, focal_length, min_focal, max_focal, options=None, **kw):
    """-> (rotations, focal_length, final_cost, summary)"""
    buf = _cm(rotations); i0, i1, rel = _edges(index0, index1, rel_rotations)
    o = options or default_options(**kw); s = BASummaryC(); f = C.c_double(focal_length)
    _lib.check(_lib.lib().ssfm_posegraph_focal_solve(ctx._p, len(rotations), buf.ctypes.data_as(c_double_p), len(i0), i0.ctypes.data_as(c_i32_p),
                                                     i1.ctypes.data_as(c_i32_p), rel.ctypes.data_as(c_double_p), C.byref(f), min_focal, max_focal,
                                                     C.byref(o), C.byref(s)), ctx._p)
    return np.transpose(buf.reshape(-1, 3, 3), (0, 2, 1)).copy(), f.value, s.final_cost, s.as_dict()


def focal_search(ctx, num_cameras, index0, index1, rel_rotations, focal_guess, focals, inward=False, return_matches=False):
    """find_best_focal_length_random (examples/spherical_sfm_tools.cpp:1418-1496) for caller-supplied trial focals ->
    (costs (T,), best_trial, rotations at the best focal (n,3,3)); with return_matches=True also the matches' rotations
    re-derived at the best focal (E,3,3)."""
    i0, i1, rel = _edges(index0, index1, rel_rotations)
    fv = np.ascontiguousarray(focals, np.float64); T = len(fv)
    costs = np.zeros(T); best = C.c_int32(0); rot = np.zeros(9 * num_cameras); relb = np.zeros(9 * len(i0))
    _lib.check(_lib.lib().ssfm_focal_search(ctx._p, num_cameras, len(i0), i0.ctypes.data_as(c_i32_p), i1.ctypes.data_as(c_i32_p),
                                            rel.ctypes.data_as(c_double_p), int(bool(inward)), float(focal_guess), T, fv.ctypes.data_as(c_double_p),
                                            costs.ctypes.data_as(c_double_p), C.byref(best), rot.ctypes.data_as(c_double_p),
                                            relb.ctypes.data_as(c_double_p)), ctx._p)
    out = (costs, best.value, np.transpose(rot.reshape(-1, 3, 3), (0, 2, 1)).copy())
    return out + (np.transpose(relb.reshape(-1, 3, 3), (0, 2, 1)).copy(),) if return_matches else out
