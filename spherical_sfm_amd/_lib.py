"""ctypes binding of libssfm_hip.so (the C ABI declared in include/ssfm.h).

There is no CPU fallback: if the HIP library is missing, or no GPU is visible when a context is created,
this raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSFM_LIB_PATH") or os.path.join(_HERE, "libssfm_hip.so")      # SSFM_LIB_PATH: a variant build of the same library (kernel experiments, scripts/gpu_gram_ld.sh)
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_u8_p = C.POINTER(C.c_uint8)
c_u32_p = C.POINTER(C.c_uint32)


class BAProblemC(C.Structure):
    _fields_ = [("num_cameras", C.c_int32), ("num_points", C.c_int32), ("num_observations", C.c_int64),
                ("cameras", c_double_p), ("points", c_double_p), ("focal", c_double_p),
                ("obs_xy", c_double_p), ("obs_cam", c_i32_p), ("obs_pt", c_i32_p),
                ("rot_fixed", c_u8_p), ("trans_fixed", c_u8_p), ("pt_fixed", c_u8_p), ("focal_fixed", C.c_int32)]


class BAOptionsC(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("min_relative_decrease", C.c_double), ("loss_type", C.c_int32), ("loss_scale", C.c_double),
                ("jacobi_scaling", C.c_int32), ("pcg_max_iterations", C.c_int32), ("pcg_tolerance", C.c_double),
                ("preconditioner", C.c_int32), ("verbose", C.c_int32)]


class BASummaryC(C.Structure):
    _fields_ = [("termination", C.c_int32), ("iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("num_linearizations", C.c_int32), ("pcg_iterations_total", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("num_residual_blocks", C.c_int64),
                ("num_residual_blocks_global", C.c_int64), ("num_points_used", C.c_int32), ("camera_dof", C.c_int32),
                ("t_flatten_s", C.c_double), ("t_upload_s", C.c_double), ("t_solve_s", C.c_double), ("t_download_s", C.c_double),
                ("t_kernel_linearize_ms", C.c_double), ("t_kernel_schur_ms", C.c_double), ("t_kernel_pcg_ms", C.c_double),
                ("t_kernel_update_ms", C.c_double), ("reduced_blocks", C.c_int32), ("band_half_width", C.c_int32),
                ("band_segments", C.c_int32), ("band_separators", C.c_int32),
                ("num_line_search_evaluations", C.c_int32), ("num_line_search_contractions", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BAPlanInfoC(C.Structure):
    _fields_ = [("camera_dof", C.c_int32), ("num_points_used", C.c_int32), ("num_points_used_global", C.c_int32),
                ("reduced_blocks", C.c_int32), ("band_half_width", C.c_int32), ("max_row_blocks", C.c_int32),
                ("num_observations_used", C.c_int64), ("num_observations_used_global", C.c_int64),
                ("band_segments", C.c_int32), ("band_separators", C.c_int32),
                ("num_points_grouped", C.c_int64), ("num_observations_grouped", C.c_int64), ("group_tasks", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RansacOptionsC(C.Structure):
    _fields_ = [("num_hypotheses", C.c_int32), ("seed", C.c_uint32), ("min_num_inliers", C.c_int32), ("final_least_squares", C.c_int32),
                ("inward", C.c_int32), ("use_poly_solver", C.c_int32), ("mode", C.c_int32), ("min_num_iterations", C.c_uint32),
                ("max_num_iterations", C.c_uint32), ("success_probability", C.c_double), ("num_lo_steps", C.c_int32),
                ("num_lsq_iterations", C.c_int32), ("threshold_multiplier", C.c_double), ("min_sample_multiplicator", C.c_int32),
                ("non_min_sample_multiplier", C.c_int32), ("lo_starting_iterations", C.c_uint32), ("fast_shuffle", C.c_int32)]


RANSAC_FIXED_BUDGET, RANSAC_REFERENCE_TRACE = 0, 1


# every symbol include/ssfm.h declares (tests check that the library exports all of them)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_uint64, C.c_int32)

DECLARED_SYMBOLS = [
    "ssfm_ctx_create", "ssfm_ctx_destroy", "ssfm_last_error", "ssfm_version", "ssfm_comm_unique_id", "ssfm_comm_init", "ssfm_comm_init_host",
    "ssfm_ba_default_options", "ssfm_ba_plan", "ssfm_ba_solve", "ssfm_ba_create", "ssfm_ba_reset", "ssfm_ba_run", "ssfm_ba_download",
    "ssfm_ba_destroy", "ssfm_ba_evaluate", "ssfm_ba_set_profiling", "ssfm_ba_kernel_times",
    "ssfm_rotavg_default_options", "ssfm_rotavg_solve", "ssfm_rotavg_cost", "ssfm_posegraph_focal_solve",
    "ssfm_ransac_default_options", "ssfm_ransac_batch", "ssfm_ransac_batch_sharded", "ssfm_ransac_batch_indexed", "ssfm_ransac_batch_indexed_sharded", "ssfm_ransac_last_kernel_ms", "ssfm_band_solve_probe", "ssfm_snode_solve_probe", "ssfm_snode_plan_probe", "ssfm_spherical_solver_probe", "ssfm_spherical_solver_poly_probe", "ssfm_build_tracks", "ssfm_debug_timing_skip_collectives", "ssfm_debug_copy_bandwidth", "ssfm_retriangulate", "ssfm_retriangulate_mode", "ssfm_retriangulate_ex", "ssfm_tri_probe", "ssfm_focal_search",
    "ssfm_sampson_refine_probe", "ssfm_sampson_refine_probe_ex", "ssfm_decompose_probe", "ssfm_nonminimal_probe", "ssfm_so3_probe", "ssfm_mt19937_probe",
    "ssfm_minimal_solver_probe", "ssfm_sampson_probe",
    "ssfm_estimator_create", "ssfm_estimator_destroy", "ssfm_estimator_minimal_solver", "ssfm_estimator_non_minimal_solver",
    "ssfm_estimator_evaluate_model", "ssfm_estimator_least_squares", "ssfm_estimator_decompose",
]


class SsfmError(RuntimeError):
    pass


def lib():
    """Load libssfm_hip.so.  Importing torch first makes the process share torch's HIP runtime
    (same SONAME libamdhip64.so.7), which is what bench.py relies on for streams / torch.distributed."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SsfmError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(make -C spherical_sfm_amd/csrc).  There is no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever is
    # loaded first wins, and loading the second copy later corrupts the heap at exit -- so make sure torch's is first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    L.ssfm_ctx_create.argtypes = [C.c_int32, vp, C.POINTER(vp)]; L.ssfm_ctx_create.restype = C.c_int
    L.ssfm_ctx_destroy.argtypes = [vp]; L.ssfm_ctx_destroy.restype = None
    L.ssfm_last_error.argtypes = [vp]; L.ssfm_last_error.restype = C.c_char_p
    L.ssfm_version.restype = C.c_int
    L.ssfm_comm_unique_id.argtypes = [c_u8_p]; L.ssfm_comm_unique_id.restype = C.c_int
    L.ssfm_comm_init.argtypes = [vp, c_u8_p, C.c_int32, C.c_int32]; L.ssfm_comm_init.restype = C.c_int
    L.ssfm_comm_init_host.argtypes = [vp, C.c_int32, C.c_int32, HOST_ALLREDUCE_FN, vp]; L.ssfm_comm_init_host.restype = C.c_int
    L.ssfm_ba_default_options.argtypes = [C.POINTER(BAOptionsC)]; L.ssfm_ba_default_options.restype = None
    L.ssfm_ba_plan.argtypes = [C.POINTER(BAProblemC), C.c_int32, C.c_int32, C.POINTER(BAPlanInfoC), c_i32_p, c_u8_p, c_i32_p]; L.ssfm_ba_plan.restype = C.c_int
    L.ssfm_ba_solve.argtypes = [vp, C.POINTER(BAProblemC), C.POINTER(BAOptionsC), C.POINTER(BASummaryC)]; L.ssfm_ba_solve.restype = C.c_int
    L.ssfm_ba_create.argtypes = [vp, C.POINTER(BAProblemC), C.POINTER(BAOptionsC), C.POINTER(vp)]; L.ssfm_ba_create.restype = C.c_int
    L.ssfm_ba_reset.argtypes = [vp]; L.ssfm_ba_reset.restype = C.c_int
    L.ssfm_ba_run.argtypes = [vp, C.POINTER(BASummaryC)]; L.ssfm_ba_run.restype = C.c_int
    L.ssfm_ba_download.argtypes = [vp, C.POINTER(BAProblemC)]; L.ssfm_ba_download.restype = C.c_int
    L.ssfm_ba_destroy.argtypes = [vp]; L.ssfm_ba_destroy.restype = None
    L.ssfm_ba_evaluate.argtypes = [vp, c_double_p, c_double_p, c_double_p]; L.ssfm_ba_evaluate.restype = C.c_int
    L.ssfm_ba_set_profiling.argtypes = [vp, C.c_int32]; L.ssfm_ba_set_profiling.restype = C.c_int
    L.ssfm_ba_kernel_times.argtypes = [vp, C.c_int32, C.c_void_p, c_i64_p, c_double_p]; L.ssfm_ba_kernel_times.restype = C.c_int
    L.ssfm_rotavg_default_options.argtypes = [C.POINTER(BAOptionsC)]; L.ssfm_rotavg_default_options.restype = None
    L.ssfm_rotavg_solve.argtypes = [vp, C.c_int32, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_double_p, C.POINTER(BAOptionsC), C.POINTER(BASummaryC)]
    L.ssfm_rotavg_solve.restype = C.c_int
    L.ssfm_rotavg_cost.argtypes = [vp, C.c_int32, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_double_p, c_double_p]; L.ssfm_rotavg_cost.restype = C.c_int
    L.ssfm_posegraph_focal_solve.argtypes = [vp, C.c_int32, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_double_p, c_double_p, C.c_double, C.c_double,
                                             C.POINTER(BAOptionsC), C.POINTER(BASummaryC)]
    L.ssfm_posegraph_focal_solve.restype = C.c_int
    L.ssfm_band_solve_probe.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_i32_p, c_double_p, c_double_p, c_i32_p, c_double_p, c_double_p, c_double_p]
    L.ssfm_band_solve_probe.restype = C.c_int
    L.ssfm_snode_solve_probe.argtypes = [vp, C.c_int32, C.c_int32, c_i32_p, c_i32_p, c_double_p, c_double_p, C.c_int32, c_double_p, c_i32_p]
    L.ssfm_snode_solve_probe.restype = C.c_int
    L.ssfm_snode_plan_probe.argtypes = [C.c_int32, C.c_int32, c_i32_p, c_i32_p, C.c_int32, c_i32_p, c_i32_p, c_i32_p, c_i32_p, c_i32_p, c_i32_p]
    L.ssfm_snode_plan_probe.restype = C.c_int
    L.ssfm_ransac_default_options.argtypes = [C.POINTER(RansacOptionsC)]; L.ssfm_ransac_default_options.restype = None
    L.ssfm_ransac_batch.argtypes = [vp, C.c_int32, c_i32_p, c_double_p, c_double_p, C.c_double, C.POINTER(RansacOptionsC), c_double_p, c_double_p,
                                    c_u8_p, c_i32_p, c_double_p, c_u32_p]
    L.ssfm_ransac_batch.restype = C.c_int
    L.ssfm_sampson_refine_probe_ex.argtypes = [vp, C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_i32_p, C.c_int32, C.c_int32, c_double_p, c_double_p]; L.ssfm_sampson_refine_probe_ex.restype = C.c_int
    L.ssfm_sampson_refine_probe.argtypes = [vp, C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_i32_p, C.c_int32, c_double_p]; L.ssfm_sampson_refine_probe.restype = C.c_int
    L.ssfm_decompose_probe.argtypes = [vp, C.c_int32, c_double_p, C.c_int32, c_double_p, c_double_p]; L.ssfm_decompose_probe.restype = C.c_int
    L.ssfm_nonminimal_probe.argtypes = [vp, C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_double_p, c_i32_p]; L.ssfm_nonminimal_probe.restype = C.c_int
    L.ssfm_so3_probe.argtypes = [vp, C.c_int32, C.c_int32, c_double_p, c_double_p]; L.ssfm_so3_probe.restype = C.c_int
    L.ssfm_mt19937_probe.argtypes = [vp, C.c_uint32, C.c_int32, c_i32_p, c_i32_p, c_i32_p, C.c_int32, c_u32_p]; L.ssfm_mt19937_probe.restype = C.c_int
    L.ssfm_ransac_batch_sharded.argtypes = L.ssfm_ransac_batch.argtypes; L.ssfm_ransac_batch_sharded.restype = C.c_int
    L.ssfm_ransac_batch_indexed.argtypes = [vp, C.c_int32, c_i32_p, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_i32_p, c_i32_p, c_i32_p, C.c_double, C.POINTER(RansacOptionsC),
                                            c_double_p, c_double_p, c_u8_p, c_i32_p, c_double_p, c_u32_p]
    L.ssfm_ransac_batch_indexed.restype = C.c_int
    L.ssfm_ransac_batch_indexed_sharded.argtypes = L.ssfm_ransac_batch_indexed.argtypes; L.ssfm_ransac_batch_indexed_sharded.restype = C.c_int
    L.ssfm_ransac_last_kernel_ms.argtypes = [vp, c_double_p]; L.ssfm_ransac_last_kernel_ms.restype = C.c_int
    L.ssfm_spherical_solver_probe.argtypes = [vp, C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_double_p, c_i32_p]
    L.ssfm_spherical_solver_probe.restype = C.c_int
    L.ssfm_spherical_solver_poly_probe.argtypes = [vp, C.c_int32, c_double_p, c_double_p, C.c_int32, c_i32_p, c_double_p, c_i32_p]
    L.ssfm_spherical_solver_poly_probe.restype = C.c_int
    L.ssfm_build_tracks.argtypes = [C.c_int32, c_i32_p, c_double_p, C.c_int32, c_i32_p, c_i32_p, c_i32_p, c_i32_p, c_i32_p, C.c_double, C.c_double,
                                    C.c_int32, c_i32_p, c_i32_p, c_u8_p, c_i64_p, c_i32_p, c_i32_p, c_double_p]
    L.ssfm_build_tracks.restype = C.c_int
    L.ssfm_retriangulate.argtypes = [vp, C.POINTER(BAProblemC), c_i32_p]
    L.ssfm_retriangulate.restype = C.c_int
    L.ssfm_debug_copy_bandwidth.argtypes = [vp, C.c_uint64, C.c_int32, C.POINTER(C.c_double)]
    L.ssfm_debug_copy_bandwidth.restype = C.c_int
    L.ssfm_debug_timing_skip_collectives.argtypes = [vp, C.c_int32]
    L.ssfm_debug_timing_skip_collectives.restype = C.c_int
    L.ssfm_retriangulate_mode.argtypes = [vp, C.POINTER(BAProblemC), C.c_int32, c_i32_p, C.POINTER(C.c_uint32), c_u8_p]
    L.ssfm_retriangulate_mode.restype = C.c_int
    L.ssfm_retriangulate_ex.argtypes = [vp, C.POINTER(BAProblemC), c_i32_p, C.POINTER(C.c_uint32), c_u8_p]
    L.ssfm_retriangulate_ex.restype = C.c_int
    L.ssfm_tri_probe.argtypes = [vp, C.POINTER(BAProblemC), C.c_int32, C.c_int32, c_i32_p, c_i32_p, c_i32_p, c_double_p, c_double_p]
    L.ssfm_tri_probe.restype = C.c_int
    L.ssfm_focal_search.argtypes = [vp, C.c_int32, C.c_int32, c_i32_p, c_i32_p, c_double_p, C.c_int32, C.c_double, C.c_int32, c_double_p, c_double_p,
                                    c_i32_p, c_double_p, c_double_p]
    L.ssfm_focal_search.restype = C.c_int
    _LIB = L
    return L


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().ssfm_last_error(ctx)
        raise SsfmError(f"ssfm error {rc}: {msg.decode() if msg else '?'}")
