/* ssfm.h -- C ABI of the MI355X-native spherical-sfm optimisation core (libssfm_hip.so).
 *
 * Plain pointers and sizes only; no C++/torch types.  Every entry point names the reference
 * interface (file:line under jonathanventura/spherical-sfm) that it replaces.  Matrices crossing this
 * boundary are COLUMN-MAJOR 3x3 (Eigen's default; the reference hands `Matrix3d::data()` to Ceres at
 * src/rotation_averaging.cpp:28-29), cameras are 6 doubles [t;r] (include/sphericalsfm/sfm_types.h:9,
 * src/sfm.cpp:104-105), points 3 doubles, observations principal-point-centred pixels
 * (examples/spherical_sfm_tools.cpp:907-908).
 *
 * Return value: 0 = ok, <0 = error (ssfm_last_error gives the text).  No exceptions cross the ABI.
 * A context is bound to one GPU and one HIP stream and is not thread-safe (one per host thread / rank).
 */
#ifndef SSFM_H
#define SSFM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SSFM_OK 0
#define SSFM_ERR_INVALID -1
#define SSFM_ERR_HIP -2
#define SSFM_ERR_NO_DEVICE -3
#define SSFM_ERR_COMM -4

/* termination_type of ceres::Solver::Summary as the reference reads it (src/sfm.cpp:278-289) */
#define SSFM_CONVERGENCE 0
#define SSFM_NO_CONVERGENCE 1
#define SSFM_FAILURE 2
#define SSFM_NOTHING_TO_DO 3 /* src/sfm.cpp:230,265-268: Optimize() returns false before solving */

typedef struct ssfm_ctx ssfm_ctx;

/* ---- context ------------------------------------------------------------------------------ */
/* device < 0: current device.  stream: a hipStream_t passed as void*, NULL = context-owned stream. */
int ssfm_ctx_create(int32_t device, void* stream, ssfm_ctx** out);
void ssfm_ctx_destroy(ssfm_ctx* ctx);
const char* ssfm_last_error(const ssfm_ctx* ctx); /* ctx may be NULL: last creation error */
int ssfm_version(void);

/* Multi-GPU (one process per GPU; RCCL over xGMI).  Rank 0 obtains an id, the host side broadcasts the
 * 128 bytes with whatever it has (torch.distributed in bench.py), every rank calls ssfm_comm_init. */
int ssfm_comm_unique_id(uint8_t id[128]);
int ssfm_comm_init(ssfm_ctx* ctx, const uint8_t id[128], int32_t nranks, int32_t rank);
/* Alternative to RCCL: the caller supplies the collective (MPI, gloo, a test harness).  The library stages each reduction
 * through pinned host memory and calls fn(user, buf, n, op) with op = SSFM_REDUCE_SUM / SSFM_REDUCE_MAX; fn must leave the
 * element-wise reduction over all ranks in buf and return 0.  Same sharding and the same reductions as the RCCL path. */
enum { SSFM_REDUCE_SUM = 0, SSFM_REDUCE_MAX = 1 };
typedef int (*ssfm_host_allreduce_fn)(void* user, double* buf, uint64_t n, int32_t op);
int ssfm_comm_init_host(ssfm_ctx* ctx, int32_t nranks, int32_t rank, ssfm_host_allreduce_fn fn, void* user);

/* Timing probe of bench.py (`timing_without_collective`), not a product setting: while on, the bundle-adjustment reductions of this context return at once, so
 * every rank iterates on its own shard (kernels and sizes of the sharded solve, results meaningless).  Warns on stderr when switched on. */
int ssfm_debug_timing_skip_collectives(ssfm_ctx* ctx, int32_t on);

/* Measured device copy bandwidth ("denominator = measured device copy bandwidth on the box, nominal also quoted"): `reps` launches of a float4
 * grid-stride copy of `bytes` bytes (>= 256 MB: beyond the Infinity Cache); GBs_out = bytes read + bytes written per second, in GB/s.  bench.py reports it. */
int ssfm_debug_copy_bandwidth(ssfm_ctx* ctx, uint64_t bytes, int32_t reps, double* GBs_out);

/* ---- bundle adjustment: replaces the body of sphericalsfm::SfM::Optimize (src/sfm.cpp:228-290) --- */
typedef struct {
    int32_t num_cameras;
    int32_t num_points;
    int64_t num_observations;
    double* cameras;            /* [num_cameras*6] [t;r], in/out  (GetCameraPtr, src/sfm.cpp:89-92) */
    double* points;             /* [num_points*3], in/out         (GetPointPtr,  src/sfm.cpp:94-97) */
    double* focal;              /* &intrinsics.focal, in/out      (src/sfm.cpp:220) */
    const double* obs_xy;       /* [M*2] */
    const int32_t* obs_cam;     /* [M] */
    const int32_t* obs_pt;      /* [M] */
    const uint8_t* rot_fixed;   /* [num_cameras] rotationFixed, NULL = all free     (src/sfm.cpp:224) */
    const uint8_t* trans_fixed; /* [num_cameras] translationFixed, NULL = all free  (src/sfm.cpp:223) */
    const uint8_t* pt_fixed;    /* [num_points] pointFixed, NULL = all free         (src/sfm.cpp:225) */
    int32_t focal_fixed;        /* focalFixed                                        (src/sfm.cpp:222) */
} ssfm_ba_problem;

typedef struct {
    /* ConfigureSolverOptions / PreOptimize values (src/sfm.cpp:194-212), rest Ceres 2.2.0 defaults */
    int32_t max_num_iterations;                /* 2000 */
    int32_t max_num_consecutive_invalid_steps; /* 100 */
    double function_tolerance;                 /* 1e-6 */
    double gradient_tolerance;                 /* 1e-10 */
    double parameter_tolerance;                /* 1e-8 */
    double initial_trust_region_radius;        /* 1e4 */
    double max_trust_region_radius;            /* 1e16 */
    double min_trust_region_radius;            /* 1e-32 */
    double min_lm_diagonal, max_lm_diagonal;   /* 1e-6, 1e32 */
    double min_relative_decrease;              /* 1e-3 */
    int32_t loss_type;                         /* 0 trivial, 1 Cauchy, 2 SoftLOne; default 1 */
    double loss_scale;                         /* 1.0 */
    int32_t jacobi_scaling;                    /* 1 */
    /* reduced-camera-system solver.  The reference solves it directly (SPARSE_SCHUR: sparse Cholesky, src/sfm.cpp:205) and so does this build by
     * default: an exact block-banded Cholesky in Cuthill-McKee order (twisted / substructured for long components).  The PCG of
     * north_star survives as a REFINEMENT that runs only when |rhs - S y| > pcg_tolerance |rhs| after the direct solve -- never observed, 0 sweeps in
     * every measured solve -- and as the block-Jacobi comparison solver (preconditioner = 1).  The "pcg" in field names below is historical: they time
     * and count the reduced solve, whichever solver ran. */
    int32_t pcg_max_iterations;                /* 1000 */
    double pcg_tolerance;                      /* |r| <= tol |b|, default 1e-10 (tracks a direct solve) */
    int32_t preconditioner;                    /* 0: block-banded Cholesky (exact on the Cuthill-McKee band) + PCG
                                                  refinement; 1: block-Jacobi PCG */
    int32_t verbose;                           /* 1: one line per LM iteration (minimizer_progress_to_stdout) */
} ssfm_ba_options;

typedef struct {
    int32_t termination;          /* SSFM_CONVERGENCE ... */
    int32_t iterations;
    int32_t num_successful_steps, num_unsuccessful_steps;
    int32_t num_linearizations;   /* n_LM of SURVEY 8d: every assemble+solve, accepted or rejected */
    int32_t pcg_iterations_total;
    double initial_cost, final_cost;
    int64_t num_residual_blocks;  /* observations that entered the problem (this rank) */
    int64_t num_residual_blocks_global;
    int32_t num_points_used;
    int32_t camera_dof;           /* 3 = spherical (all translations fixed), 6 = general */
    double t_flatten_s, t_upload_s, t_solve_s, t_download_s;   /* host wall-clock */
    double t_kernel_linearize_ms, t_kernel_schur_ms, t_kernel_pcg_ms, t_kernel_update_ms; /* hipEvent sums; "pcg" = the reduced solve (banded Cholesky + back substitution) */
    int32_t reduced_blocks;       /* non-zero DCxDC blocks of the reduced camera system */
    int32_t band_half_width;      /* block half-bandwidth of S in the Cuthill-McKee order */
    int32_t band_segments;        /* workgroups of the reduced-system factorisation: connected components, long ones cut */
    int32_t band_separators;      /* into segments by this many separators of band_half_width block rows (0 = none cut) */
    /* bounded problems (ssfm_posegraph_focal_solve): Ceres' projected line search inside the trust-region loop */
    int32_t num_line_search_evaluations;   /* function evaluations beyond the candidate's */
    int32_t num_line_search_contractions;  /* iterations whose step the search shortened */
} ssfm_ba_summary;

void ssfm_ba_default_options(ssfm_ba_options* o);

/* Test probe for the reduced-system solver (the stand-in for Ceres' SPARSE_SCHUR Cholesky, src/sfm.cpp:276-279): factor + solve
 * a caller-supplied symmetric positive definite block band.  band: [N][b+1][dc*dc], block d of row i = (i, i-d), rows in band
 * order; comp_ptr: [ncomp+1] contiguous row ranges of independent components; Y: [2][N*dc] right-hand sides in, solutions out.
 * segs_seps_fail: [3] = factorisation workgroups, separators (0 = no component was cut), non-positive-pivot flag.
 * Zdump [b*dc][N*dc], Ddump [seps][b*dc][b*dc], Tdump [seps][2][b*dc]: intermediates of the substructured path, or NULL. */
int ssfm_band_solve_probe(ssfm_ctx* ctx, int32_t dc, int32_t N, int32_t b, int32_t ncomp, const int32_t* comp_ptr, const double* band,
                          double* Y, int32_t* segs_seps_fail, double* Zdump, double* Ddump, double* Tdump);

/* Test probe (round 6): the supernodal ring / chain solver of the reduced camera system alone (csrc/snode.h; it takes the place of the sparse Cholesky
 * inside Ceres' SPARSE_SCHUR, reference src/sfm.cpp:205,273, whenever every component of the camera graph is a chain or a closed loop whose tracks span
 * <= 30 / dc cameras).  S: block-CSR (row_ptr [Nc + 1], col_idx), every coupled pair of cameras stored once in either orientation plus the diagonal blocks,
 * row-major dc x dc blocks; rhs2: [2][Nc * dc]; nr = 1 | 2 right-hand sides; Y: [2][Nc * dc] out (camera order).
 * info: [4] = {1 if the plan applies (else nothing ran), workgroups, rows of the closing separator, non-positive-pivot flag}. */
int ssfm_snode_solve_probe(ssfm_ctx* ctx, int32_t dc, int32_t Nc, const int32_t* row_ptr, const int32_t* col_idx, const double* S_val, const double* rhs2,
                           int32_t nr, double* Y, int32_t* info);
/* Host-only (no GPU): the plan of that solver for a block structure.  sizes: [8] = {applies, workgroups, rows reserved for the closing separator, cameras per supernode,
 * capacity of a node's camera list, ints of half_rec, of step_rec, of node_cam}; the tables (layouts: csrc/snode.h) are copied out when the pointers are non-null;
 * tab_len: capacity of tab in, its length out. */
int ssfm_snode_plan_probe(int32_t dc, int32_t Nc, const int32_t* row_ptr, const int32_t* col_idx, int32_t num_cus, int32_t* sizes, int32_t* half_rec, int32_t* step_rec,
                          int32_t* node_cam, int32_t* tab, int32_t* tab_len);

/* Host-only planning (no GPU needed): what the flatten rules of src/sfm.cpp:240-263 keep, how the used points
 * are sharded over ranks (contiguous ranges balanced by observations), and the camera elimination order. */
typedef struct {
    int32_t camera_dof, num_points_used, num_points_used_global, reduced_blocks, band_half_width, max_row_blocks;
    int64_t num_observations_used, num_observations_used_global;
    int32_t band_segments, band_separators;   /* as in ssfm_ba_summary (SSFM_BAND_SEGMENTS=1 disables cutting, =P forces P per component) */
    /* round 3: points of this rank whose Schur blocks are assembled as Gram products on the matrix cores (runs of >= 32 consecutive points with the same 3..8
     * cameras; everything else goes through the sorted pair lists), their observations, and the wave tasks they are cut into */
    int64_t num_points_grouped, num_observations_grouped;
    int32_t group_tasks, reserved;
} ssfm_ba_plan_info;
/* point_ids: [num_points] capacity or NULL (receives the original ids of this rank's used points, in order);
 * obs_used: [num_observations] or NULL (1 where the observation enters this rank's problem);
 * cam_pos: [num_cameras] or NULL (camera -> position in the elimination order). */
int ssfm_ba_plan(const ssfm_ba_problem* p, int32_t nranks, int32_t rank, ssfm_ba_plan_info* info, int32_t* point_ids,
                 uint8_t* obs_used, int32_t* cam_pos);

/* One call = flatten (src/sfm.cpp:240-263 rules) + upload + device LM loop + scatter back.
 * The context keeps the resident plan of the last problem STRUCTURE it solved (observation ids in order, fixed masks, which
 * points are zero, sizes): a call with the same structure only uploads parameters and pixels -- the drivers' Optimize ->
 * Retriangulate -> Optimize pattern.  summary.t_flatten_s is 0 on such a call.  SSFM_NO_PLAN_CACHE=1 in the environment
 * disables the cache; ssfm_ctx_destroy releases it.
 * Reproducibility: by default the assembly of the reduced camera system adds with fp64 atomics, so repeated solves agree to ~1e-11, not bit for bit (Ceres' own
 * multi-threaded evaluation behind src/sfm.cpp:276 is in the same position).  SSFM_DETERMINISTIC=1 in the environment, read when a handle is created, switches a
 * single-GPU handle to order-independent accumulation (fixed-point limbs + integer atomics, csrc/det_acc.h): bit-identical repeats at ~1.2x the iteration time;
 * a context with a communicator refuses it (SSFM_ERR_INVALID). */
int ssfm_ba_solve(ssfm_ctx* ctx, ssfm_ba_problem* p, const ssfm_ba_options* o, ssfm_ba_summary* s);

/* Staged form: problem stays resident in HBM between runs (bench.py, multi-GPU sharding). */
typedef struct ssfm_ba_handle ssfm_ba_handle;
int ssfm_ba_create(ssfm_ctx* ctx, const ssfm_ba_problem* p, const ssfm_ba_options* o, ssfm_ba_handle** out);
int ssfm_ba_reset(ssfm_ba_handle* h);                          /* restore the uploaded initial parameters */
int ssfm_ba_run(ssfm_ba_handle* h, ssfm_ba_summary* s);        /* device LM loop only */
int ssfm_ba_download(ssfm_ba_handle* h, ssfm_ba_problem* p);   /* scatter parameters back */
void ssfm_ba_destroy(ssfm_ba_handle* h);
/* one evaluation at the uploaded state, for kernel parity tests: cost, residuals [M*2] (robustified),
 * jacobians [M*2*10] (focal | t | r | X, robustified, unscaled); entries of unused observations are 0. */
int ssfm_ba_evaluate(ssfm_ba_handle* h, double* cost, double* residuals, double* jacobians);
/* on != 0: bracket every kernel launch of ssfm_ba_run with hipEvents (slower; for bench.py's profile leg) */
int ssfm_ba_set_profiling(ssfm_ba_handle* h, int32_t on);
/* per-kernel timing of the last run: name -> (launches, total ms); returns number of entries written */
int ssfm_ba_kernel_times(ssfm_ba_handle* h, int32_t max_entries, char names[][32], int64_t* launches, double* total_ms);

/* ---- SO(3) pose graphs ------------------------------------------------------------------------------
 * rotations: [n*9] column-major 3x3 (std::vector<Eigen::Matrix3d>), in/out; index0/index1/rel_rotations: the fields of
 * sphericalsfm::RelativeRotation (include/sphericalsfm/rotation_averaging.h:9-14), rel = R1 * R0^T, column-major.
 * Options: ssfm_ba_options with the pose-graph defaults (Ceres defaults: 50 iterations, 5 invalid steps;
 * SoftLOneLoss(0.03)).  summary->final_cost is the value the reference functions return. */
void ssfm_rotavg_default_options(ssfm_ba_options* o);
/* optimize_rotations (src/rotation_averaging.cpp:44-91).
 * LARGE GRAPHS AND THE 50-ITERATION CAP (src/rotation_averaging.cpp:75-80 solves with Ceres' default max_num_iterations = 50): from about 2000 nodes on the
 * capped run has NOT converged, and where it stops depends on the order in which the implementation sums -- ANY implementation, Ceres with another thread count
 * included.  Measured on the 2000 / 4000-node rings of tests/test_rotavg_gpu.py: this library and the CPU oracle, two summation orders of the same algorithm, end
 * up to 0.3 rad apart in single rotations with final costs within 5 %.  What IS reproducible is the converged minimum: with max_num_iterations raised until the
 * tolerances fire, this library's answer is a fixed point of the oracle to < 1e-5 rad (the oracle restarted from it moves less than that).  A caller who needs
 * north_star's 1e-5 on such graphs must raise options->max_num_iterations (a few hundred suffice); with the reference's cap the contract is "same cost band". */
int ssfm_rotavg_solve(ssfm_ctx* ctx, int32_t n, double* rotations, int32_t num_edges, const int32_t* index0, const int32_t* index1,
                      const double* rel_rotations, const ssfm_ba_options* o, ssfm_ba_summary* s);
/* get_cost (src/uncalibrated_pose_graph.cpp:116-145) */
int ssfm_rotavg_cost(ssfm_ctx* ctx, int32_t n, const double* rotations, int32_t num_edges, const int32_t* index0, const int32_t* index1,
                     const double* rel_rotations, double* cost);
/* optimize_rotations_and_focal_length (src/uncalibrated_pose_graph.cpp:147-203); *focal_length is multiplied by the
 * optimised multiplier, which is kept inside [min_focal, max_focal] / focal_length */
int ssfm_posegraph_focal_solve(ssfm_ctx* ctx, int32_t n, double* rotations, int32_t num_edges, const int32_t* index0,
                               const int32_t* index1, const double* rel_rotations, double* focal_length, double min_focal,
                               double max_focal, const ssfm_ba_options* o, ssfm_ba_summary* s);

/* ---- batched spherical relative-pose RANSAC ----------------------------------------------------------
 * Replaces the OpenMP loop body of estimate_pairwise (examples/spherical_sfm_tools.cpp:332-420): per image pair
 * LocallyOptimizedMSAC<..., SphericalEstimator>::EstimateModel (include/RansacLib/ransac.h:128) with the action-matrix
 * minimal solver, final least squares, inlier mask and Decompose.  Pair p owns rays [pair_ptr[p], pair_ptr[p+1]) of
 * u / v ([total*3]; u = first view, v = second view, as RayPair, include/sphericalsfm/ray.h:8-10).
 * squared_inlier_threshold = (inlier_threshold_px * Kinv(0,0))^2 (spherical_sfm_tools.cpp:315).
 * Outputs (any may be NULL): E, R [num_pairs*9] column-major; inlier_mask [total]; num_inliers, scores [num_pairs].
 * R is the identity where num_inliers <= min_num_inliers (the reference skips such pairs, :410).
 *
 * mode: how the hypotheses of a pair are generated.
 *   SSFM_RANSAC_REFERENCE_TRACE (default): RansacLib's LocallyOptimizedMSAC control flow, draw for draw -- both std::mt19937 streams
 *     (seeded with `seed`, include/RansacLib/sampling.h:52, ransac.h:145-146) and libstdc++'s uniform_int_distribution are restated
 *     on the device, so a pair evaluates the reference's minimal samples, runs LocalOptimization (ransac.h:341-407: the initial
 *     least-squares fit on <= min_sample_multiplicator * 3 shuffled inliers, then num_lo_steps x [NonMinimalSolver + iterated fits])
 *     at the reference's iterations, adapts max_num_iterations from the inlier ratio (utils.h:110-140) and stops where the reference
 *     stops.  Differences to a CPU build are floating-point rounding.  A chunk of iterations is evaluated in parallel, one lane each.
 *     Rounding caveats of that sweep: the Sampson quotient of the hypothesis phase uses the hardware reciprocal + two Newton steps (~1 ulp)
 *     where the CPU divides, so a strict "<" between two scores that agree to the last bits may go the other way (never observed to change a
 *     result: 256 of 256 test pairs and 200 sampled pairs of the full BASELINE configs[3] run replay the oracle's trace); and with LO steps
 *     on, NonMinimalSolver on an ill-conditioned sample can differ from a CPU eigen-solver by far more than rounding (tolerances).
 *   SSFM_RANSAC_FIXED_BUDGET: num_hypotheses counter-based samples per pair scored in parallel, best one refined on its inliers
 *     (no LO, no adaptive stopping; statistically equivalent result, more arithmetic). */
#define SSFM_RANSAC_FIXED_BUDGET 0
#define SSFM_RANSAC_REFERENCE_TRACE 1
typedef struct {
    int32_t num_hypotheses;      /* FIXED_BUDGET: minimal samples per pair (default 1024) */
    uint32_t seed;               /* RansacOptions::random_seed_ (default 0) */
    int32_t min_num_inliers;     /* acceptance threshold of estimate_pairwise */
    int32_t final_least_squares; /* LORansacOptions::final_least_squares_ (default 1, spherical_sfm_tools.cpp:318) */
    int32_t inward;              /* SphericalEstimator(..., inward) */
    int32_t use_poly_solver;     /* SphericalEstimator(..., use_poly_solver, ...): 0 = action matrix (estimate_pairwise's choice), 1 = quartic */
    int32_t mode;                /* SSFM_RANSAC_REFERENCE_TRACE */
    uint32_t min_num_iterations, max_num_iterations;   /* RansacOptions: 100, 10000 (ransac.h:50-51) */
    double success_probability;  /* 0.9999 (ransac.h:52) */
    int32_t num_lo_steps;        /* estimate_pairwise sets 0 (spherical_sfm_tools.cpp:316); LORansacOptions default 10 */
    int32_t num_lsq_iterations;  /* estimate_pairwise sets 0 (:317); default 4 */
    double threshold_multiplier; /* sqrt(2) (ransac.h:67) */
    int32_t min_sample_multiplicator;   /* 7: least-squares fits use <= 7 * 3 inliers (ransac.h:69,412); 1..21 */
    int32_t non_min_sample_multiplier;  /* 3: non-minimal samples hold max(4, min(3 * 3, inliers / 2)) rays (ransac.h:70,369-373); 1..3 */
    uint32_t lo_starting_iterations;    /* 50 (ransac.h:71) */
    int32_t fast_shuffle;        /* 1: the draws of a Fisher-Yates tail that is thrown away are only checked for rejections, in parallel (same stream) */
} ssfm_ransac_options;
void ssfm_ransac_default_options(ssfm_ransac_options* o);
/* stats: [num_pairs*2] = RansacStatistics::num_iterations, number_lo_iterations per pair (zeros in FIXED_BUDGET mode), or NULL.
 * Pairs are streamed through the GPU in slabs (two pinned staging buffers, the upload of slab k+1 under the kernels of slab k), so one
 * call may hold any number of pairs (BASELINE configs[3]: 2000 frames = 1 999 000 pairs); pairs with more than ~2700 correspondences
 * keep their rays in HBM/L2 instead of LDS. */
int ssfm_ransac_batch(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v,
                      double squared_inlier_threshold, const ssfm_ransac_options* o, double* E, double* R, uint8_t* inlier_mask,
                      int32_t* num_inliers, double* scores, uint32_t* stats);
/* Multi-GPU form of the same call (BASELINE configs[3]: exhaustive pairwise RANSAC over several GPUs; the reference's
 * counterpart is the `#pragma omp parallel for` over matches, spherical_sfm_tools.cpp:332).  Every rank of the context's
 * communicator (ssfm_comm_init / ssfm_comm_init_host) passes the SAME full arguments; rank r estimates pairs r, r + nranks, ...
 * with the random streams of their global indices, and one sum all-reduce returns every pair's result on every rank --
 * bit-identical to ssfm_ransac_batch on one GPU.  Without a communicator it is ssfm_ransac_batch. */
int ssfm_ransac_batch_sharded(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v,
                              double squared_inlier_threshold, const ssfm_ransac_options* o, double* E, double* R,
                              uint8_t* inlier_mask, int32_t* num_inliers, double* scores, uint32_t* stats);
/* The same estimation from what estimate_pairwise really holds (examples/spherical_sfm_tools.cpp:340-375): per-frame feature lists and per-pair
 * match lists.  feat_rays[feat_ptr[f] + k] = Kinv (x, y, 1) of feature k of frame f (:362-370, computed once per feature instead of once per
 * match); pair p matches feature match_idx0[i] of frame pair_frame0[p] with feature match_idx1[i] of frame pair_frame1[p] for i in
 * [match_ptr[p], match_ptr[p+1]).  The feature rays are uploaded once, the match lists stream through the pinned double buffer (8 bytes per
 * correspondence instead of the 48 of ssfm_ransac_batch: BASELINE configs[3] moves 8 GB instead of 48 GB) and a gather kernel lays the ray
 * pairs out on the device.  Outputs and random streams exactly as ssfm_ransac_batch on the gathered rays (inlier_mask indexed like the matches). */
int ssfm_ransac_batch_indexed(ssfm_ctx* ctx, int32_t num_frames, const int32_t* feat_ptr, const double* feat_rays,
                              int32_t num_pairs, const int32_t* pair_frame0, const int32_t* pair_frame1, const int32_t* match_ptr,
                              const int32_t* match_idx0, const int32_t* match_idx1,
                              double squared_inlier_threshold, const ssfm_ransac_options* o, double* E, double* R, uint8_t* inlier_mask,
                              int32_t* num_inliers, double* scores, uint32_t* stats);
/* The indexed form over the ranks of the context's communicator (multi-GPU estimate_pairwise, BASELINE configs[3]): every rank passes the SAME full
 * arguments, uploads the feature rays of all frames and the match lists of ITS pairs (r, r + nranks, ...), and the result table of
 * ssfm_ransac_batch_sharded is all-reduced -- bit-identical to ssfm_ransac_batch_indexed on one GPU.  Without a communicator it is that call. */
int ssfm_ransac_batch_indexed_sharded(ssfm_ctx* ctx, int32_t num_frames, const int32_t* feat_ptr, const double* feat_rays,
                                      int32_t num_pairs, const int32_t* pair_frame0, const int32_t* pair_frame1, const int32_t* match_ptr,
                                      const int32_t* match_idx0, const int32_t* match_idx1,
                                      double squared_inlier_threshold, const ssfm_ransac_options* o, double* E, double* R, uint8_t* inlier_mask,
                                      int32_t* num_inliers, double* scores, uint32_t* stats);
/* Measurement aid (the RANSAC leg is priced in FP64 flop/s of Sampson scoring, not GB/s): device time of the kernels of this context's last
 * ssfm_ransac_batch* call, summed over its slabs (hipEvent brackets on the solver stream; uploads and read-backs are outside).  No reference counterpart. */
int ssfm_ransac_last_kernel_ms(ssfm_ctx* ctx, double* ms);
/* ---- the reference's estimator interface for ONE pair (rays resident on the device) ------------------------------------------------
 * One entry point per virtual of sphericalsfm::Estimator<Eigen::Matrix3d> / EssentialEstimator (include/sphericalsfm/estimator.h:7-29) as
 * SphericalEstimator implements them (include/sphericalsfm/spherical_estimator.h:8-35, src/spherical_estimator.cpp:67-164): what a host-side
 * RANSAC driver such as ransac_lib::LocallyOptimizedMSAC (include/RansacLib/ransac.h:128) calls.  The C++ mirror of the class is
 * spherical_sfm_amd/csrc/shim/spherical_estimator.h.  Thousands of pairs go through ssfm_ransac_batch instead.  Matrices column-major.
 *   create(u, v [n*3], use_poly_solver, inward)      SphericalEstimator(correspondences, use_poly_solver, inward)   spherical_estimator.h:15
 *   minimal_solver(sample [3..9]) -> Es [36], 0 | 4    MinimalSolver: always four candidates, real parts of complex ones       :80-84
 *   non_minimal_solver(sample [3..9]) -> E, ok         NonMinimalSolver                                                          :86-108
 *   evaluate_model(E) -> errors [n]                    EvaluateModelOnPoint(E, i) for every i                                     :67-78
 *   least_squares(sample [0..n], E in/out)             LeastSquares                                                              :110-157
 *   decompose(E) -> R [9], t [3]                       Decompose                                                                  :159-164 */
typedef struct ssfm_estimator ssfm_estimator;
int ssfm_estimator_create(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t use_poly_solver, int32_t inward, ssfm_estimator** out);
void ssfm_estimator_destroy(ssfm_estimator* e);
int ssfm_estimator_minimal_solver(ssfm_estimator* e, const int32_t* sample, int32_t sample_size, double* Es, int32_t* num_models);
int ssfm_estimator_non_minimal_solver(ssfm_estimator* e, const int32_t* sample, int32_t sample_size, double* E, int32_t* ok);
int ssfm_estimator_evaluate_model(ssfm_estimator* e, const double* E, double* errors);
int ssfm_estimator_least_squares(ssfm_estimator* e, const int32_t* sample, int32_t sample_size, double* E);
int ssfm_estimator_decompose(ssfm_estimator* e, const double* E, double* R, double* t);

/* ---- deterministic probes of the estimator's pieces (parity tests; one workgroup per task) --------------------------------------
 * ssfm_sampson_refine_probe: SphericalEstimator::LeastSquares (src/spherical_estimator.cpp:110-157) -- task t refines E_inout[t] (column-
 *   major, in/out) on the rays lists[task_ptr[t] .. task_ptr[t+1]) of the ONE pair (u, v).  The fit has SIX free parameters [r1; t1]: the
 *   reference sets r0, t0, u, v constant (:140-144) and leaves t1 free; t1 is dropped when E is rebuilt from so3exp(r1) (:156).
 * ssfm_sampson_refine_probe_ex: the same with its trace -- variant 0 = the workgroup-cooperative fit, 1 = the one-wave fit of the batched
 *   LO-MSAC kernel; trace [tasks*10] = [r1; t1], Levenberg-Marquardt iterations, status (0 converged / 1 iteration limit / 2 invalid
 *   steps / 3 evaluation failure), initial cost, final cost (may be NULL).
 * ssfm_decompose_probe: decompose_spherical_essential_matrix (src/spherical_utils.cpp:16-66) + so3exp = SphericalEstimator::Decompose
 *   (src/spherical_estimator.cpp:159-164): r_out [tasks*3] angle-axis, R_out [tasks*9] column-major (either may be NULL).
 * ssfm_nonminimal_probe: SphericalEstimator::NonMinimalSolver (src/spherical_estimator.cpp:86-108) on samples of 3..9 rays.
 * ssfm_so3_probe (row a9): what = 0 so3exp (src/so3.cpp:16-23), 1 so3ln (:25-69), 2 ceres::AngleAxisToRotationMatrix,
 *   3 ceres::RotationMatrixToAngleAxis, as the device code evaluates them; matrices column-major.
 * ssfm_mt19937_probe: nraw raw words of std::mt19937(seed), then uniform_int_distribution<int>(lo[i], hi[i]) draws from the same
 *   engine, through the device generator of the reference-trace mode.
 * ssfm_minimal_solver_probe: SphericalEstimator::MinimalSolver as the reference returns it (src/spherical_estimator.cpp:80-84): always four
 *   candidates per 3-point sample, the real parts of complex solutions included; Es [S*36] column-major, counts [S] (0 or 4).
 * ssfm_sampson_probe: EvaluateModelOnPoint (src/spherical_estimator.cpp:67-78) of T models on every ray: errors [T*n]. */
int ssfm_minimal_solver_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples, int32_t use_poly_solver,
                              double* Es, int32_t* counts);
int ssfm_sampson_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t T, const double* Es, double* errors);
int ssfm_sampson_refine_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t tasks, const int32_t* task_ptr,
                              const int32_t* lists, int32_t inward, double* E_inout);
int ssfm_sampson_refine_probe_ex(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t tasks, const int32_t* task_ptr,
                                 const int32_t* lists, int32_t inward, int32_t variant, double* E_inout, double* trace);
int ssfm_decompose_probe(ssfm_ctx* ctx, int32_t tasks, const double* E, int32_t inward, double* r_out, double* R_out);
int ssfm_nonminimal_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t tasks, const int32_t* task_ptr,
                          const int32_t* lists, double* E_out, int32_t* ok_out);
int ssfm_so3_probe(ssfm_ctx* ctx, int32_t what, int32_t n, const double* in, double* out);
int ssfm_mt19937_probe(ssfm_ctx* ctx, uint32_t seed, int32_t n, const int32_t* lo, const int32_t* hi, int32_t* draws, int32_t nraw, uint32_t* raw);
/* parity probe: spherical_solver_action_matrix (src/spherical_solvers.cpp:102-311) on S given 3-point samples
 * (samples: [S*3] indices into the n rays).  Es: [S*36] = up to 4 column-major 3x3 per sample (real solutions only),
 * counts: [S]. */
int ssfm_spherical_solver_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples,
                                double* Es, int32_t* counts);
/* same for spherical_solver_polynomial (src/spherical_solvers.cpp:313-660, SolveQuartic :15-69) */
int ssfm_spherical_solver_poly_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples,
                                     double* Es, int32_t* counts);

/* ---- focal-length search around the pose graph (examples/spherical_sfm_tools.cpp:1418-1496, find_best_focal_length_random)
 * costs[t] = loop_constraint_cost_fn(focals[t]) (:1138-1157): every match's essential matrix (rebuilt from its rotation,
 * :1429-1433) is rescaled by T = diag(f/f0, f/f0, 1) on both sides and decomposed again (transform_image_matches, :1118-1132),
 * the rotations are chained over the matches (k-1, k) (initialize_rotations_sequential, :794-813; the -sequential mode, the
 * GraphOptim initialisation is out of scope) and get_cost (src/uncalibrated_pose_graph.cpp:116-145) is evaluated -- one
 * workgroup per trial.  The reference draws the trial focals from std::random_device; here the caller supplies them.
 * best_trial: first minimum (strict <, :1467-1474); rotations_best [n*9]: the chained rotations at that focal, column-major,
 * rel_rotations_best [num_edges*9]: the matches' rotations re-derived at that focal (what run_optimization, :1160-1188, feeds to
 * optimize_rotations_and_focal_length = ssfm_posegraph_focal_solve).  Any output may be NULL. */
int ssfm_focal_search(ssfm_ctx* ctx, int32_t n, int32_t num_edges, const int32_t* index0, const int32_t* index1,
                      const double* rel_rotations, int32_t inward, double focal_guess, int32_t num_trials, const double* focals,
                      double* costs, int32_t* best_trial, double* rotations_best, double* rel_rotations_best);

/* ---- SfM::Retriangulate (src/sfm.cpp:156-192) ------------------------------------------------------------------
 * Re-estimates EVERY point of the problem from its observations and the current cameras/focal with the per-point
 * ransac_lib::LocallyOptimizedMSAC<Point, ..., TriangulationEstimator> of the reference (src/triangulation_estimator.cpp:46-127,
 * include/RansacLib/ransac.h:128-428; squared inlier threshold 4 px^2, final least squares on, every other option a LORansacOptions
 * default), one GPU lane per point.  The default mode REPLAYS THE REFERENCE'S TRACE: both std::mt19937 streams run from seed 0 for
 * every point, so the sampler's pair sequence (a function of the track length only) and the raw words of the local optimisation's
 * stream are drawn once on the host with libstdc++'s generators, and the device walks RansacLib's control flow draw for draw -- same
 * iteration counts, same local-optimisation runs, same inlier sets as a CPU build (tests/test_retriangulate_gpu.py compares with the
 * oracle bit for bit).  p->points is overwritten; points with < 3 observations or < 3 inliers become (0,0,0) exactly as in the
 * reference (which removes them from later Optimize calls).  num_inliers_out: [num_points] or NULL.  The *_fixed masks are ignored,
 * like the reference does.  SSFM_RETRI_ENUMERATE=1 selects the enumerating kernel of rounds 1-2 (every observation pair once, no random
 * stream: statistical agreement only, ~2.5x faster) as the default of the two forms without a mode argument; ssfm_retriangulate_mode
 * takes the mode explicitly (SSFM_RETRI_MODE_TRACE / SSFM_RETRI_MODE_ENUMERATE).
 * WHAT "THE REFERENCE'S TRACE" MEANS: bit-for-bit agreement is with this repository's CPU restatement (oracle/triangulation_oracle.cpp: one-sided
 * Jacobi SVD for the DLT, a hand-written Levenberg-Marquardt with Ceres' rules), compiled without fused multiply-adds.  A real build of the reference
 * computes the DLT with Eigen::JacobiSVD and the point refinement with Ceres; last bits decide every `score < best` branch and therefore every later
 * draw, so against such a build the replay is the same ALGORITHM and random streams, statistically equivalent results (same inlier sets for all but
 * marginal observations), not the same trace.  Parity of this row is "unpinned" like the rest of the oracle.
 * ssfm_retriangulate_ex: the same with its trace -- stats_out [2*num_points] = RansacStatistics::num_iterations, number_lo_iterations
 *   of every point's run; inlier_flags_out [num_observations] = 1 where the observation is in the final stats.inlier_indices of its
 *   point (either may be NULL; trace mode only).
 * ssfm_tri_probe: TriangulationEstimator's pieces, one lane per task, on point task_pt[t] with the observation subset
 *   lists[task_ptr[t] .. task_ptr[t+1]) (positions in the point's observation list, cameras ascending); out [tasks*4]:
 *   what 0 NonMinimalSolver (1..6 observations) -> X, 0;  what 1 LeastSquares from X_in[t] -> X, Levenberg-Marquardt iterations;
 *   what 2 ScoreModel / GetInliers of X_in[t] -> MSAC score at 4, inliers at 4, inliers at 4 sqrt 2, error of observation 0. */
#define SSFM_RETRI_MODE_TRACE 0
#define SSFM_RETRI_MODE_ENUMERATE 1
int ssfm_retriangulate(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t* num_inliers_out);
int ssfm_retriangulate_mode(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t mode, int32_t* num_inliers_out, uint32_t* stats_out, uint8_t* inlier_flags_out);
int ssfm_retriangulate_ex(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t* num_inliers_out, uint32_t* stats_out, uint8_t* inlier_flags_out);
int ssfm_tri_probe(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t what, int32_t tasks, const int32_t* task_pt, const int32_t* task_ptr, const int32_t* lists,
                   const double* X_in, double* out);

/* ---- feature tracks: the integer part of build_sfm (examples/spherical_sfm_tools.cpp:862-950), host only ------
 * Keyframe k owns features [feat_ptr[k], feat_ptr[k+1]) of feat_xy ([total*2] pixels).  Match set s links keyframes
 * (ms_index0[s], ms_index1[s]) with pairs (m_f0[m], m_f1[m]), m in [ms_ptr[s], ms_ptr[s+1]), feature indices local to their
 * keyframe, first index ascending (std::map order).  Outputs: tracks [total] (-1 = unmatched; ids bit-exact with the
 * reference's AddPoint sequence), num_points (ids issued, merged ones included), point_alive [>= num match pairs] (0 =
 * removed by MergePoint), observations camera-major / point-ascending (the iteration order of SfM's maps), centred by
 * (centerx, centery); obs_* need capacity 2 * (number of match pairs) and may be NULL to only count. */
int ssfm_build_tracks(int32_t num_keyframes, const int32_t* feat_ptr, const double* feat_xy, int32_t num_match_sets,
                      const int32_t* ms_index0, const int32_t* ms_index1, const int32_t* ms_ptr, const int32_t* m_f0, const int32_t* m_f1,
                      double centerx, double centery, int32_t merge, int32_t* tracks, int32_t* num_points, uint8_t* point_alive,
                      int64_t* num_observations, int32_t* obs_cam, int32_t* obs_pt, double* obs_xy);

#ifdef __cplusplus
}
#endif
#endif
