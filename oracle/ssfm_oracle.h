/* ORACLE (test infrastructure only) -- C API of the CPU restatement.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (spherical_sfm_amd/) never links, imports or calls it.
 *
 * PARITY PARTLY PINNED (round 5).  The reference holds no golden vectors / known-answer tests for any function on this path
 * (finding F7) and its un-vendored dependencies (Ceres 2.2.0, Eigen 3.4) are absent here, so the reference as a whole cannot be run.
 * What CAN run here does: include/RansacLib/{ransac,sampling,utils}.h are compiled as they stand and drive this oracle's estimators (oracle/_ref, lomsac_reference.hpp),
 * src/spherical_solvers.cpp:14-98 (SolveQuartic*) is compiled as it stands, and the generated coefficient code of both minimal solvers is evaluated from the
 * reference's own lines -- tests/golden/ref_*.npz, tests/test_reference_pins_cpu.py pin lomsac.hpp, the quartic, the constraint matrices and both solver back ends to them.
 * Everything that runs through Ceres or Eigen in the reference (the trust-region loops, losses, Jets, QR / LU / eigen / SVD decompositions) remains a restatement anchored
 * on the reference's call sites (cited per function) and on oracle-free known-answer tests (tests/test_oracle_*.py, tests/test_scipy_anchor_cpu.py).
 */
#ifndef SSFM_ORACLE_H
#define SSFM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t num_cameras;
    int32_t num_points;
    int64_t num_observations;
    double* cameras;           /* [num_cameras*6] = [t;r] per camera (sfm.cpp:104-105), in/out */
    double* points;            /* [num_points*3], in/out */
    double* focal;             /* shared focal (sfm.cpp:220), in/out */
    const double* obs_xy;      /* [M*2] principal-point-centred pixels */
    const int32_t* obs_cam;    /* [M] */
    const int32_t* obs_pt;     /* [M] */
    const uint8_t* rot_fixed;  /* [num_cameras] or NULL */
    const uint8_t* trans_fixed;/* [num_cameras] or NULL */
    const uint8_t* pt_fixed;   /* [num_points] or NULL */
    int32_t focal_fixed;
} oracle_ba_problem;

typedef struct {
    int32_t max_num_iterations;                 /* sfm.cpp:205 -> 2000 */
    int32_t max_num_consecutive_invalid_steps;  /* sfm.cpp:206 -> 100 */
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
    double min_lm_diagonal, max_lm_diagonal, min_relative_decrease;
    int32_t loss_type;        /* 0 trivial, 1 Cauchy, 2 SoftLOne */
    double loss_scale;        /* 'a' of the loss */
    int32_t jacobi_scaling;
    int32_t num_threads;      /* sfm.cpp:209 -> 16; clamped to hardware */
    int32_t verbose;
} oracle_lm_options;

typedef struct {
    int32_t termination;      /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE, 3 nothing to do */
    int32_t iterations;
    int32_t num_successful_steps, num_unsuccessful_steps, num_linear_solves;
    double initial_cost, final_cost;
    int64_t num_residual_blocks;   /* observations that entered the problem */
    int32_t num_points_used;
    int32_t threads_used;
    double t_total_s, t_flatten_s, t_linearize_s, t_schur_s, t_cholesky_s, t_cost_s;
} oracle_summary;

void oracle_ba_default_options(oracle_lm_options* o);   /* values of sfm.cpp:194-212 + Ceres 2.2.0 defaults */
int oracle_ba_solve(oracle_ba_problem* p, const oracle_lm_options* o, oracle_summary* s);

/* one evaluation at the given state: cost, per-observation residuals [M*2] (robustified),
 * and Jacobian blocks per observation [M*2*10] (columns: focal, t(3), r(3), X(3); unrobustified
 * raw autodiff Jacobian when raw!=0).  Observations of unused points get zeros.  Outputs may be NULL. */
/* the reference's O(Np x Nc) map-probing build loop (src/sfm.cpp:240-263), timed; returns the residual blocks it would add */
int64_t oracle_reference_style_flatten(const oracle_ba_problem* p, double* seconds);
int oracle_ba_evaluate(const oracle_ba_problem* p, const oracle_lm_options* o, int32_t raw,
                       double* cost, double* residuals, double* jacobians, uint8_t* obs_used);

/* dense reduced camera(+focal) system at the given state, fixed layout [focal | 6 per camera], size (6Nc+1)^2;
 * unscaled robustified Jacobian, point blocks damped by mu.  Small problems only (tests). */
int oracle_ba_reduced_system(const oracle_ba_problem* p, const oracle_lm_options* o, double mu, double* S_out, double* rhs_out);

/* SO(3) helpers (reference src/so3.cpp); matrices column-major */
void oracle_so3exp(const double r[3], double R[9]);
void oracle_so3ln(const double R[9], double r[3]);
/* Ceres rotation.h restatements */
void oracle_angle_axis_rotate_point(const double aa[3], const double pt[3], double out[3]);
void oracle_angle_axis_to_rotation_matrix(const double aa[3], double R[9]);
void oracle_rotation_matrix_to_angle_axis(const double R[9], double aa[3]);

/* rotation averaging / pose graph (reference src/rotation_averaging.cpp:44-91,
 * src/uncalibrated_pose_graph.cpp:116-203).  rotations: [n*9] column-major in/out. */
double oracle_optimize_rotations(int32_t n, double* rotations, int32_t num_edges, const int32_t* index0,
                                 const int32_t* index1, const double* rel_rotations, oracle_summary* s);
double oracle_get_cost(int32_t n, const double* rotations, int32_t num_edges, const int32_t* index0,
                       const int32_t* index1, const double* rel_rotations);
double oracle_optimize_rotations_and_focal_length(int32_t n, double* rotations, int32_t num_edges,
                                                  const int32_t* index0, const int32_t* index1,
                                                  const double* rel_rotations, double* focal_length,
                                                  double min_focal, double max_focal, oracle_summary* s);
/* test hooks: tolerances of the next pose-graph solves (max_iterations = 0 restores the reference's Ceres defaults); how many iterations
 * of the last solve had their step shortened by the projected line search (bounded problems only) */
void oracle_pose_graph_test_options(int32_t max_iterations, double function_tolerance, double gradient_tolerance, double parameter_tolerance);
int32_t oracle_pose_graph_last_line_search_contractions(void);
/* residual + 3x6 (or 3x7 with focal) Jacobian of one edge, for kernel parity tests.
 * kind: 0 RotationError (meas = R), 1 PoseGraphError, 2 UncalibratedPoseGraphError */
void oracle_rotation_edge(int32_t kind, const double r0[3], const double r1[3], double f,
                          const double Rmeas[9], double scale, double res[3], double jac[21]);

/* spherical relative pose (reference src/spherical_estimator.cpp, src/spherical_solvers.cpp, src/spherical_utils.cpp,
 * include/RansacLib); rays u,v: [n*3]; all 3x3 matrices column-major */
double oracle_sampson(const double E[9], const double u[3], const double v[3]);
int oracle_spherical_solver(int32_t n, const double* u, const double* v, int32_t num_sample, const int32_t* sample, double Es[36]);
/* spherical_solver_polynomial (src/spherical_solvers.cpp:313-660); imag: imaginary parts of the quartic roots whose real parts were used */
int oracle_spherical_solver_poly(int32_t n, const double* u, const double* v, int32_t num_sample, const int32_t* sample, double Es[36], double imag[4]);
void oracle_make_spherical_essential_matrix(const double R[9], int32_t inward, double E[9]);
void oracle_decompose_spherical_essential_matrix(const double E[9], int32_t inward, double r[3], double t[3]);
void oracle_sampson_least_squares(int32_t n, const double* u, const double* v, int32_t num_sample, const int32_t* sample, int32_t inward, double E[9]);
/* same, trace exposed: x_out = [r1; t1] (the reference leaves t1 FREE, src/spherical_estimator.cpp:140-144), stats = {iterations, termination,
 * successful, unsuccessful steps}, costs = {initial, final}; r_only != 0: the 3-parameter fit of rounds 1-2 (negative test only) */
void oracle_sampson_least_squares_ex(int32_t n, const double* u, const double* v, int32_t num_sample, const int32_t* sample, int32_t inward, int32_t r_only,
                                     double E[9], double x_out[6], int32_t stats[4], double costs[2]);
int oracle_ransac_pair(int32_t n, const double* u, const double* v, int32_t inward, double squared_inlier_threshold, uint32_t min_iterations,
                       uint32_t max_iterations, uint32_t seed, int32_t min_num_inliers, double E[9], double R[9], uint8_t* inlier_mask,
                       uint32_t* iterations, double* best_score);

/* LocallyOptimizedMSAC (include/RansacLib/ransac.h:128-420) with every LORansacOptions field exposed; stats[2] = num_iterations, number_lo_iterations */
int oracle_lomsac_pair(int32_t n, const double* u, const double* v, int32_t inward, int32_t use_poly, double sq_thresh, uint32_t min_it, uint32_t max_it,
                       double success_prob, uint32_t seed, int32_t num_lo_steps, int32_t num_lsq_it, double thresh_mult, int32_t min_sample_mult,
                       int32_t non_min_mult, uint32_t lo_start, int32_t final_lsq, int32_t min_num_inliers, double E[9], double R[9],
                       uint8_t* inlier_mask, uint32_t* stats, double* best_score);
int oracle_nonminimal_solver(int32_t n, const double* u, const double* v, int32_t num_sample, const int32_t* sample, double E[9]);
/* std::mt19937(seed): nraw raw words, then uniform_int_distribution<int>(lo[i], hi[i]) draws (libstdc++, as a build of the reference draws them) */
void oracle_mt19937_draws(uint32_t seed, int32_t n, const int32_t* lo, const int32_t* hi, int32_t* out, int32_t nraw, uint32_t* raw);

/* SfM::Retriangulate (src/sfm.cpp:156-192): every point is re-estimated from its observations with the per-point LO-MSAC
 * of TriangulationEstimator; points with < 3 observations or < 3 inliers become (0,0,0).  num_inliers_out: [num_points] or NULL */
int oracle_retriangulate(oracle_ba_problem* p, int32_t num_threads, int32_t* num_inliers_out);
/* same with its trace: stats_out [2*num_points] = RansacStatistics::num_iterations, number_lo_iterations; inlier_flags_out [num_observations]
 * = membership in the final stats.inlier_indices (either may be NULL) */
int oracle_retriangulate_ex(oracle_ba_problem* p, int32_t num_threads, int32_t* num_inliers_out, uint32_t* stats_out, uint8_t* inlier_flags_out);
/* TriangulationEstimator's pieces on chosen observation subsets (src/triangulation_estimator.cpp:46-127); see triangulation_oracle.cpp */
int oracle_tri_probe(oracle_ba_problem* p, int32_t what, int32_t tasks, const int32_t* task_pt, const int32_t* task_ptr, const int32_t* lists,
                     const double* X_in, double* out);

#ifdef __cplusplus
}
#endif
#endif
