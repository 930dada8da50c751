"""ORACLE (test infrastructure only) -- ctypes loader for oracle/libssfm_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY PARTLY PINNED: see oracle/ssfm_oracle.h (what the reference itself can compute in this image pins the RANSAC / solver part; Ceres / Eigen paths are restated).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_i32_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)


class BAProblemC(C.Structure):
    _fields_ = [("num_cameras", C.c_int32), ("num_points", C.c_int32), ("num_observations", C.c_int64),
                ("cameras", c_double_p), ("points", c_double_p), ("focal", c_double_p),
                ("obs_xy", c_double_p), ("obs_cam", c_i32_p), ("obs_pt", c_i32_p),
                ("rot_fixed", c_u8_p), ("trans_fixed", c_u8_p), ("pt_fixed", c_u8_p), ("focal_fixed", C.c_int32)]


class LMOptionsC(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("min_relative_decrease", C.c_double), ("loss_type", C.c_int32), ("loss_scale", C.c_double),
                ("jacobi_scaling", C.c_int32), ("num_threads", C.c_int32), ("verbose", C.c_int32)]


class SummaryC(C.Structure):
    _fields_ = [("termination", C.c_int32), ("iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("num_linear_solves", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("num_residual_blocks", C.c_int64),
                ("num_points_used", C.c_int32), ("threads_used", C.c_int32),
                ("t_total_s", C.c_double), ("t_flatten_s", C.c_double), ("t_linearize_s", C.c_double),
                ("t_schur_s", C.c_double), ("t_cholesky_s", C.c_double), ("t_cost_s", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    so = os.path.join(_HERE, "libssfm_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "libssfm_oracle.so"] + (["-B"] if force else []))
    return so


class _Missing:
    """stands for a symbol the loaded library does not export: takes the signature assignments, refuses the call"""
    def __init__(self, name):
        object.__setattr__(self, "_name", name)

    def __setattr__(self, k, v):
        pass

    def __call__(self, *a):
        raise AttributeError("oracle library has no symbol " + self._name)


class _Lenient:
    """CDLL view for oracle/_ref/libssfm_ref.so, which holds the RANSAC / triangulation part only"""
    def __init__(self, L):
        object.__setattr__(self, "_L", L)

    def __getattr__(self, n):
        try:
            return getattr(self._L, n)
        except AttributeError:
            return _Missing(n)


def reference_lib_path():
    """oracle/_ref/libssfm_ref.so (`make -C oracle ref`, build container only): the estimators of this oracle driven by the REFERENCE'S OWN
    include/RansacLib template (lomsac_reference.hpp) + its SolveQuartic (ref_quartic_wrap.cpp); None when it has not been built"""
    so = os.path.join(_HERE, "_ref", "libssfm_ref.so")
    return so if os.path.exists(so) else None


_REF_LIB = None


class reference_ransaclib:
    """with oracle.reference_ransaclib(): ...   every RANSAC / Retriangulate wrapper of this module runs on oracle/_ref/libssfm_ref.so"""
    def __enter__(self):
        global _LIB, _REF_LIB
        if _REF_LIB is None:
            so = reference_lib_path()
            if so is None:
                raise FileNotFoundError("oracle/_ref/libssfm_ref.so: run `make -C oracle ref` where /root/reference exists")
            _REF_LIB = _Lenient(C.CDLL(so))
            _declare(_REF_LIB)
            _REF_LIB.oracle_lomsac_is_reference.restype = C.c_int32
            assert _REF_LIB.oracle_lomsac_is_reference() == 1
        lib()
        self._saved = _LIB
        _LIB = _REF_LIB
        return _REF_LIB

    def __exit__(self, *exc):
        global _LIB
        _LIB = self._saved
        return False


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libssfm_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        _declare(L)
        _LIB = L
    return _LIB


def _declare(L):
    if True:
        L.oracle_ba_default_options.argtypes = [C.POINTER(LMOptionsC)]
        L.oracle_ba_solve.argtypes = [C.POINTER(BAProblemC), C.POINTER(LMOptionsC), C.POINTER(SummaryC)]
        L.oracle_ba_solve.restype = C.c_int
        L.oracle_ba_evaluate.argtypes = [C.POINTER(BAProblemC), C.POINTER(LMOptionsC), C.c_int32, c_double_p,
                                         c_double_p, c_double_p, c_u8_p]
        L.oracle_ba_evaluate.restype = C.c_int
        L.oracle_ba_reduced_system.argtypes = [C.POINTER(BAProblemC), C.POINTER(LMOptionsC), C.c_double, c_double_p, c_double_p]
        L.oracle_ba_reduced_system.restype = C.c_int
        for name in ("oracle_so3exp", "oracle_so3ln", "oracle_angle_axis_to_rotation_matrix",
                     "oracle_rotation_matrix_to_angle_axis"):
            getattr(L, name).argtypes = [c_double_p, c_double_p]
            getattr(L, name).restype = None
        L.oracle_angle_axis_rotate_point.argtypes = [c_double_p, c_double_p, c_double_p]
        L.oracle_optimize_rotations.argtypes = [C.c_int32, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_double_p, C.POINTER(SummaryC)]
        L.oracle_optimize_rotations.restype = C.c_double
        L.oracle_get_cost.argtypes = [C.c_int32, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_double_p]
        L.oracle_get_cost.restype = C.c_double
        L.oracle_optimize_rotations_and_focal_length.argtypes = [C.c_int32, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_double_p,
                                                                 c_double_p, C.c_double, C.c_double, C.POINTER(SummaryC)]
        L.oracle_optimize_rotations_and_focal_length.restype = C.c_double
        L.oracle_rotation_edge.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_double, c_double_p, C.c_double, c_double_p, c_double_p]
        L.oracle_rotation_edge.restype = None
        L.oracle_sampson.argtypes = [c_double_p, c_double_p, c_double_p]; L.oracle_sampson.restype = C.c_double
        L.oracle_spherical_solver.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_double_p]; L.oracle_spherical_solver.restype = C.c_int
        L.oracle_spherical_solver_poly.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_double_p, c_double_p]; L.oracle_spherical_solver_poly.restype = C.c_int
        L.oracle_make_spherical_essential_matrix.argtypes = [c_double_p, C.c_int32, c_double_p]; L.oracle_make_spherical_essential_matrix.restype = None
        L.oracle_decompose_spherical_essential_matrix.argtypes = [c_double_p, C.c_int32, c_double_p, c_double_p]
        L.oracle_decompose_spherical_essential_matrix.restype = None
        L.oracle_sampson_least_squares.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, C.c_int32, c_double_p]
        L.oracle_sampson_least_squares.restype = None
        L.oracle_ransac_pair.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_int32, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                                         c_double_p, c_double_p, c_u8_p, C.POINTER(C.c_uint32), c_double_p]
        L.oracle_ransac_pair.restype = C.c_int
        L.oracle_lomsac_pair.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_int32, C.c_int32, C.c_double, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32,
                                         C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, c_double_p, c_double_p,
                                         c_u8_p, C.POINTER(C.c_uint32), c_double_p]
        L.oracle_lomsac_pair.restype = C.c_int
        L.oracle_nonminimal_solver.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_double_p]; L.oracle_nonminimal_solver.restype = C.c_int
        L.oracle_mt19937_draws.argtypes = [C.c_uint32, C.c_int32, c_i32_p, c_i32_p, c_i32_p, C.c_int32, C.POINTER(C.c_uint32)]; L.oracle_mt19937_draws.restype = None
        L.oracle_retriangulate.argtypes = [C.POINTER(BAProblemC), C.c_int32, c_i32_p]; L.oracle_retriangulate.restype = C.c_int
        L.oracle_solver_from_basis.argtypes = [c_double_p, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p]; L.oracle_solver_from_basis.restype = C.c_int
        L.oracle_solve_quartic.argtypes = [C.c_double] * 5 + [c_double_p]; L.oracle_solve_quartic.restype = None


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_i32_p)


def _up(a):
    return a.ctypes.data_as(c_u8_p)


def default_options(**kw):
    o = LMOptionsC()
    lib().oracle_ba_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class _Held:
    """Keeps the numpy buffers of a BAProblemC alive."""
    def __init__(self, prob):
        self.cams = np.ascontiguousarray(prob.cameras, np.float64).copy()
        self.pts = np.ascontiguousarray(prob.points, np.float64).copy()
        self.focal = np.array([prob.focal], np.float64)
        self.xy = np.ascontiguousarray(prob.obs_xy, np.float64)
        self.oc = np.ascontiguousarray(prob.obs_cam, np.int32)
        self.op = np.ascontiguousarray(prob.obs_pt, np.int32)
        self.rf = np.ascontiguousarray(prob.rot_fixed, np.uint8)
        self.tf = np.ascontiguousarray(prob.trans_fixed, np.uint8)
        self.pf = np.ascontiguousarray(prob.pt_fixed, np.uint8)
        self.c = BAProblemC(len(self.cams), len(self.pts), len(self.oc), _dp(self.cams), _dp(self.pts), _dp(self.focal),
                            _dp(self.xy), _ip(self.oc), _ip(self.op), _up(self.rf), _up(self.tf), _up(self.pf),
                            1 if prob.focal_fixed else 0)


def ba_solve(prob, options=None, **kw):
    """prob: spherical_sfm_amd.synth.BAProblem-like.  Returns (cameras, points, focal, summary dict)."""
    h = _Held(prob)
    o = options or default_options(**kw)
    s = SummaryC()
    rc = lib().oracle_ba_solve(C.byref(h.c), C.byref(o), C.byref(s))
    assert rc == 0
    return h.cams, h.pts, float(h.focal[0]), s.as_dict()


def ba_evaluate(prob, raw=False, options=None):
    """Returns cost, residuals (M,2), jacobians (M,2,10), used mask (M,)."""
    h = _Held(prob)
    o = options or default_options()
    M = len(h.oc)
    cost = C.c_double(0)
    res = np.zeros((M, 2)); jac = np.zeros((M, 2, 10)); used = np.zeros(M, np.uint8)
    lib().oracle_ba_evaluate(C.byref(h.c), C.byref(o), 1 if raw else 0, C.byref(cost), _dp(res), _dp(jac), _up(used))
    return cost.value, res, jac, used


def ba_reduced_system(prob, mu=1.0, options=None):
    """Dense S ((6Nc+1)^2) and rhs of the unscaled Schur complement at the problem's state (tests only)."""
    h = _Held(prob)
    o = options or default_options()
    n = 6 * len(h.cams) + 1
    S = np.zeros((n, n)); rhs = np.zeros(n)
    lib().oracle_ba_reduced_system(C.byref(h.c), C.byref(o), mu, _dp(S), _dp(rhs))
    return S, rhs


def _vec3(fn, a, n_out):
    a = np.ascontiguousarray(a, np.float64); out = np.zeros(n_out)
    fn(_dp(a), _dp(out)); return out


def so3exp(r):
    """-> (3,3) row/col indexed R[i,j] (library is column-major)."""
    return _vec3(lib().oracle_so3exp, r, 9).reshape(3, 3).T.copy()


def so3ln(R):
    return _vec3(lib().oracle_so3ln, np.asarray(R, np.float64).T.copy(), 3)


def angle_axis_to_rotation_matrix(r):
    return _vec3(lib().oracle_angle_axis_to_rotation_matrix, r, 9).reshape(3, 3).T.copy()


def rotation_matrix_to_angle_axis(R):
    return _vec3(lib().oracle_rotation_matrix_to_angle_axis, np.asarray(R, np.float64).T.copy(), 3)


def angle_axis_rotate_point(r, p):
    r = np.ascontiguousarray(r, np.float64); p = np.ascontiguousarray(p, np.float64); out = np.zeros(3)
    lib().oracle_angle_axis_rotate_point(_dp(r), _dp(p), _dp(out)); return out


def _colmajor(Rs):
    """(n,3,3) row-indexed -> flat column-major (n*9,)"""
    return np.ascontiguousarray(np.transpose(np.asarray(Rs, np.float64), (0, 2, 1))).reshape(-1).copy()


def optimize_rotations(R, i0, i1, Rrel):
    buf = _colmajor(R); rel = _colmajor(Rrel); s = SummaryC()
    i0 = np.ascontiguousarray(i0, np.int32); i1 = np.ascontiguousarray(i1, np.int32)
    cost = lib().oracle_optimize_rotations(len(R), _dp(buf), len(i0), _ip(i0), _ip(i1), _dp(rel), C.byref(s))
    return np.transpose(buf.reshape(-1, 3, 3), (0, 2, 1)).copy(), cost, s.as_dict()


def get_cost(R, i0, i1, Rrel):
    buf = _colmajor(R); rel = _colmajor(Rrel)
    i0 = np.ascontiguousarray(i0, np.int32); i1 = np.ascontiguousarray(i1, np.int32)
    return lib().oracle_get_cost(len(R), _dp(buf), len(i0), _ip(i0), _ip(i1), _dp(rel))


def optimize_rotations_and_focal_length(R, i0, i1, Rrel, focal, min_focal, max_focal):
    buf = _colmajor(R); rel = _colmajor(Rrel); s = SummaryC(); f = C.c_double(focal)
    i0 = np.ascontiguousarray(i0, np.int32); i1 = np.ascontiguousarray(i1, np.int32)
    cost = lib().oracle_optimize_rotations_and_focal_length(len(R), _dp(buf), len(i0), _ip(i0), _ip(i1), _dp(rel),
                                                            C.byref(f), min_focal, max_focal, C.byref(s))
    return np.transpose(buf.reshape(-1, 3, 3), (0, 2, 1)).copy(), f.value, cost, s.as_dict()


def pose_graph_test_options(max_iterations=0, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8):
    """Tolerances of the following pose-graph solves (0 iterations = back to the reference's Ceres defaults)."""
    lib().oracle_pose_graph_test_options.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_double]
    lib().oracle_pose_graph_test_options(max_iterations, function_tolerance, gradient_tolerance, parameter_tolerance)


def pose_graph_last_line_search_contractions():
    lib().oracle_pose_graph_last_line_search_contractions.restype = C.c_int32
    return lib().oracle_pose_graph_last_line_search_contractions()


def rotation_edge(kind, r0, r1, f, Rmeas, scale):
    r0 = np.ascontiguousarray(r0, np.float64); r1 = np.ascontiguousarray(r1, np.float64)
    Rm = np.ascontiguousarray(np.asarray(Rmeas, np.float64).T).reshape(-1).copy()
    res = np.zeros(3); jac = np.zeros(21)
    lib().oracle_rotation_edge(kind, _dp(r0), _dp(r1), f, _dp(Rm), scale, _dp(res), _dp(jac))
    return res, jac.reshape(3, 7)


# ---------------------------------------------------------------- spherical relative pose
def _m(M):
    """(3,3) indexed [i,j] -> column-major flat"""
    return np.ascontiguousarray(np.asarray(M, np.float64).T).reshape(-1).copy()


def _um(flat):
    return np.asarray(flat, np.float64).reshape(3, 3).T.copy()


def sampson(E, u, v):
    Ec = _m(E); u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    return lib().oracle_sampson(_dp(Ec), _dp(u), _dp(v))


def spherical_solver(u, v, sample):
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64); s = np.ascontiguousarray(sample, np.int32)
    out = np.zeros(36)
    k = lib().oracle_spherical_solver(len(u), _dp(u), _dp(v), len(s), _ip(s), _dp(out))
    return [_um(out[9 * i:9 * i + 9]) for i in range(k)]


def spherical_solver_poly(u, v, sample):
    """spherical_solver_polynomial -> (list of E (3,3), imaginary parts of the quartic roots)"""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64); s = np.ascontiguousarray(sample, np.int32)
    out = np.zeros(36); im = np.zeros(4)
    k = lib().oracle_spherical_solver_poly(len(u), _dp(u), _dp(v), len(s), _ip(s), _dp(out), _dp(im))
    return [_um(out[9 * i:9 * i + 9]) for i in range(k)], im[:k]


def make_spherical_essential_matrix(R, inward=False):
    out = np.zeros(9); Rc = _m(R)
    lib().oracle_make_spherical_essential_matrix(_dp(Rc), int(inward), _dp(out)); return _um(out)


def decompose_spherical_essential_matrix(E, inward=False):
    r = np.zeros(3); t = np.zeros(3); Ec = _m(E)
    lib().oracle_decompose_spherical_essential_matrix(_dp(Ec), int(inward), _dp(r), _dp(t)); return r, t


def sampson_least_squares(u, v, sample, E, inward=False):
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64); s = np.ascontiguousarray(sample, np.int32)
    Ec = _m(E)
    lib().oracle_sampson_least_squares(len(u), _dp(u), _dp(v), len(s), _ip(s), int(inward), _dp(Ec)); return _um(Ec)


def sampson_least_squares_ex(u, v, sample, E, inward=False, r_only=False):
    """-> dict(E, x = [r1; t1], iterations, termination, successful, unsuccessful, initial_cost, final_cost); r_only = the 3-parameter fit
    that rounds 1-2 ran by mistake (t1 pinned), kept for the negative test"""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64); s = np.ascontiguousarray(sample, np.int32)
    Ec = _m(E); x = np.zeros(6); st = (C.c_int32 * 4)(); cs = np.zeros(2)
    f = lib().oracle_sampson_least_squares_ex
    f.argtypes = [C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, C.c_int32, C.c_int32, c_double_p, c_double_p, C.POINTER(C.c_int32), c_double_p]
    f.restype = None
    f(len(u), _dp(u), _dp(v), len(s), _ip(s), int(inward), int(r_only), _dp(Ec), _dp(x), st, _dp(cs))
    return dict(E=_um(Ec), x=x, iterations=st[0], termination=st[1], successful=st[2], unsuccessful=st[3], initial_cost=cs[0], final_cost=cs[1])


def ransac_pair(u, v, sq_thresh, inward=False, min_iterations=100, max_iterations=10000, seed=0, min_num_inliers=0):
    """estimate_pairwise's per-pair work.  -> dict(E, R, inliers mask, num_inliers, iterations, score)"""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    E = np.zeros(9); R = np.zeros(9); mask = np.zeros(len(u), np.uint8); it = C.c_uint32(0); sc = C.c_double(0)
    n = lib().oracle_ransac_pair(len(u), _dp(u), _dp(v), int(inward), sq_thresh, min_iterations, max_iterations, seed, min_num_inliers,
                                 _dp(E), _dp(R), _up(mask), C.byref(it), C.byref(sc))
    return dict(E=_um(E), R=_um(R), inliers=mask.astype(bool), num_inliers=n, iterations=it.value, score=sc.value)


def lomsac_pair(u, v, sq_thresh, inward=False, use_poly=False, min_iterations=100, max_iterations=10000, success_probability=0.9999, seed=0,
                num_lo_steps=0, num_lsq_iterations=0, threshold_multiplier=2.0 ** 0.5, min_sample_multiplicator=7, non_min_sample_multiplier=3,
                lo_starting_iterations=50, final_least_squares=True, min_num_inliers=0):
    """LocallyOptimizedMSAC + estimate_pairwise's tail with every option exposed (defaults = estimate_pairwise's, spherical_sfm_tools.cpp:314-318)."""
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    E = np.zeros(9); R = np.zeros(9); mask = np.zeros(max(len(u), 1), np.uint8); st = (C.c_uint32 * 2)(); sc = C.c_double(0)
    n = lib().oracle_lomsac_pair(len(u), _dp(u), _dp(v), int(inward), int(use_poly), sq_thresh, min_iterations, max_iterations, success_probability, seed,
                                 num_lo_steps, num_lsq_iterations, threshold_multiplier, min_sample_multiplicator, non_min_sample_multiplier,
                                 lo_starting_iterations, int(final_least_squares), min_num_inliers, _dp(E), _dp(R), _up(mask), st, C.byref(sc))
    return dict(E=_um(E), R=_um(R), inliers=mask[:len(u)].astype(bool), num_inliers=n, iterations=st[0], lo_runs=st[1], score=sc.value)


def solver_from_basis(B, variant):
    """both minimal solvers behind their nullspace basis B (6x3): -> dict(C (6,10) as the reference lays it out, Es (4,3,3), imag (4,), abcde (5,))
    variant 0: spherical_solver_action_matrix (src/spherical_solvers.cpp:127-308), 1: spherical_solver_polynomial (:339-654)"""
    B = np.ascontiguousarray(B, np.float64).reshape(6, 3)
    Cm = np.zeros(60); Es = np.zeros(36); im = np.zeros(4); ab = np.zeros(5)
    k = lib().oracle_solver_from_basis(_dp(B), int(variant), _dp(Cm), _dp(Es), _dp(im), _dp(ab))
    return dict(n=k, C=Cm.reshape(6, 10), Es=np.stack([_um(Es[9 * i:9 * i + 9]) for i in range(4)]), imag=im, abcde=ab)


def solve_quartic(a, b, c, d, e):
    """SolveQuartic (src/spherical_solvers.cpp:15-69) as restated -> 4 complex roots in the reference's order"""
    out = np.zeros(8)
    lib().oracle_solve_quartic(a, b, c, d, e, _dp(out))
    return out[0::2] + 1j * out[1::2]


def nonminimal_solver(u, v, sample):
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64); s = np.ascontiguousarray(sample, np.int32)
    out = np.zeros(9)
    ok = lib().oracle_nonminimal_solver(len(u), _dp(u), _dp(v), len(s), _ip(s), _dp(out))
    return ok, _um(out)


def mt19937_draws(seed, lo, hi, nraw=0):
    """(raw words (nraw,), uniform_int_distribution<int>(lo[i], hi[i]) draws) of one std::mt19937(seed)"""
    lo = np.ascontiguousarray(lo, np.int32); hi = np.ascontiguousarray(hi, np.int32)
    out = np.zeros(max(len(lo), 1), np.int32); raw = np.zeros(max(nraw, 1), np.uint32)
    lib().oracle_mt19937_draws(seed, len(lo), _ip(lo), _ip(hi), _ip(out), nraw, raw.ctypes.data_as(C.POINTER(C.c_uint32)))
    return raw[:nraw], out[:len(lo)]


def reference_style_flatten(prob):
    """(residual blocks, seconds) of the reference's own O(Np x Nc) build loop on std::map storage (src/sfm.cpp:240-263)."""
    h = _Held(prob); sec = C.c_double(0)
    lib().oracle_reference_style_flatten.restype = C.c_int64
    n = lib().oracle_reference_style_flatten(C.byref(h.c), C.byref(sec))
    return int(n), sec.value


def retriangulate(prob, num_threads=16):
    """SfM::Retriangulate on a BAProblem-like object -> (points (Np,3), num_inliers (Np,))"""
    h = _Held(prob)
    nin = np.zeros(len(h.pts), np.int32)
    lib().oracle_retriangulate(C.byref(h.c), num_threads, _ip(nin))
    return h.pts, nin


def retriangulate_ex(prob, num_threads=16):
    """-> (points, num_inliers, iterations (Np,), local-optimisation runs (Np,), inlier flags (M,) per observation)"""
    h = _Held(prob)
    nin = np.zeros(len(h.pts), np.int32); st = np.zeros(2 * max(len(h.pts), 1), np.uint32); fl = np.zeros(max(len(h.oc), 1), np.uint8)
    f = lib().oracle_retriangulate_ex
    f.argtypes = [C.c_void_p, C.c_int32, c_i32_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]; f.restype = C.c_int
    f(C.byref(h.c), num_threads, _ip(nin), st.ctypes.data_as(C.POINTER(C.c_uint32)), _up(fl))
    st = st[:2 * len(h.pts)].reshape(-1, 2)
    return h.pts, nin, st[:, 0].copy(), st[:, 1].copy(), fl[:len(h.oc)].astype(bool)


def tri_probe(prob, what, task_pt, lists, X_in=None):
    """TriangulationEstimator's pieces (triangulation_oracle.cpp: oracle_tri_probe) -> (tasks, 4)"""
    h = _Held(prob)
    task_pt = np.ascontiguousarray(task_pt, np.int32); T = len(task_pt)
    ptr = np.zeros(T + 1, np.int32)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32).reshape(-1) for l in lists]) if ptr[-1] else np.zeros(1, np.int32), np.int32)
    X = np.ascontiguousarray(X_in if X_in is not None else np.zeros((T, 3)), np.float64).reshape(-1)
    out = np.zeros(4 * T)
    f = lib().oracle_tri_probe
    f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, c_i32_p, c_i32_p, c_i32_p, c_double_p, c_double_p]; f.restype = C.c_int
    rc = f(C.byref(h.c), what, T, _ip(task_pt), _ip(ptr), _ip(flat), _dp(X), _dp(out))
    assert rc == 0, rc
    return out.reshape(T, 4)


def lsq_log(fn):
    """runs fn() with the call log on -> (fn's result, [dict(sample, E_in, E_out, x, iterations)] LeastSquares calls in order); the
    NonMinimalSolver calls of the same run are left in lsq_log.nonminimal = [dict(sample, E_out, after_lsq_calls)]"""
    L = lib()
    L.oracle_lsq_log_get.argtypes = [C.c_int32, C.c_int32, c_i32_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int32)]; L.oracle_lsq_log_get.restype = C.c_int32
    L.oracle_lsq_log_enable(1)
    try:
        res = fn()
    finally:
        n = L.oracle_lsq_log_count(); log = []
        for i in range(n):
            smp = np.zeros(4096, np.int32); Ei = np.zeros(9); Eo = np.zeros(9); x = np.zeros(6); it = C.c_int32(0)
            ns = L.oracle_lsq_log_get(i, len(smp), _ip(smp), _dp(Ei), _dp(Eo), _dp(x), C.byref(it))
            log.append(dict(sample=smp[:ns].copy(), E_in=_um(Ei), E_out=_um(Eo), x=x, iterations=it.value))
        L.oracle_nonmin_log_get.argtypes = [C.c_int32, c_i32_p, c_double_p, C.POINTER(C.c_int32)]; L.oracle_nonmin_log_get.restype = C.c_int32
        nm = []
        for i in range(L.oracle_nonmin_log_count()):
            smp = np.zeros(9, np.int32); Eo = np.zeros(9); after = C.c_int32(0)
            ns = L.oracle_nonmin_log_get(i, _ip(smp), _dp(Eo), C.byref(after))
            nm.append(dict(sample=smp[:ns].copy(), E_out=_um(Eo), after_lsq_calls=after.value))
        lsq_log.nonminimal = nm
        L.oracle_lsq_log_enable(0)
    return res, log
