// spherical_sfm_amd -- batched spherical relative-pose RANSAC.
//
// Replaces the per-pair body of estimate_pairwise (reference examples/spherical_sfm_tools.cpp:332-420), i.e.
//   LocallyOptimizedMSAC<Matrix3d, ..., SphericalEstimator>::EstimateModel   include/RansacLib/ransac.h:128-275
//   SphericalEstimator::MinimalSolver -> spherical_solver_action_matrix        src/spherical_solvers.cpp:102-311
//   SphericalEstimator::EvaluateModelOnPoint (Sampson)                         src/spherical_estimator.cpp:67-78
//   SphericalEstimator::LeastSquares (final_least_squares_ = true)             src/spherical_estimator.cpp:110-157
//   SphericalEstimator::Decompose / decompose_spherical_essential_matrix       src/spherical_utils.cpp:16-66
// with ONE launch for thousands of pairs: a workgroup per image pair keeps the pair's rays in LDS; every lane draws
// 3-point samples (counter-based RNG), solves the minimal problem (QR nullspace -> 6x10 cubic constraints -> 4x4 action
// matrix -> eigen-solutions) and MSAC-scores its up-to-4 models against all rays (LDS broadcast reads); a block arg-min
// picks the pair's best model.  A second kernel refines it on its inliers (3-dof LM on the Sampson residuals, Ceres
// rules) and decomposes E into R.  The reference's sequential, adaptively terminated sampling (std::mt19937) is replaced
// by a fixed hypothesis budget evaluated in parallel, so parity is on the final R / inlier set, not on the sample trace
// (SURVEY 7 "RANSAC determinism").  Complex eigen-pairs of the action matrix are skipped: the reference scores the real
// part of such eigenvectors, which is never a valid model.
#include <algorithm>
#include <cstdio>
#include <atomic>
#include <thread>
#include "ransac_device.h"
#include "sampson_lsq.h"

namespace ssfm {

// probe for parity tests: one lane per given sample
template <bool POLY>
__global__ void k_solver_probe(int S, const int* __restrict__ sample, const double* __restrict__ u, const double* __restrict__ v,
                               double* __restrict__ Es, int* __restrict__ counts) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double u3[9], v3[9];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { u3[3 * i + k] = u[3 * sample[3 * s + i] + k]; v3[3 * i + k] = v[3 * sample[3 * s + i] + k]; }
    double E[36];
    const int c = spherical_minimal_solver<POLY>(u3, v3, E);
    counts[s] = c;
    for (int k = 0; k < 36; k++) Es[36 * (size_t)s + k] = (k < 9 * c) ? E[k] : 0.0;
}

// ---- kernel 1: hypotheses + MSAC scores + per-pair arg-min ----------------------------------------------
template <bool POLY>
__global__ void __launch_bounds__(256)
k_ransac_hypotheses(const int* __restrict__ pair_ptr, const double* __restrict__ u, const double* __restrict__ v, double sq_thresh,
                    int num_hyp, unsigned long long seed, const int* __restrict__ pair_id, double* __restrict__ bestE,
                    double* __restrict__ bestScore) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sScore[256]; __shared__ int sIdx[256];
    const int pair = blockIdx.x;
    const int stream_id = pair_id ? pair_id[pair] : pair;      // a sharded batch keeps the random stream of the pair's global index
    const int r0 = pair_ptr[pair], n = pair_ptr[pair + 1] - r0;
    double* su = lds; double* sv = lds + (size_t)3 * n;
    for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) { su[i] = u[(size_t)3 * r0 + i]; sv[i] = v[(size_t)3 * r0 + i]; }
    __syncthreads();
    double myBest = 1.79e308; double myE[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n >= 3) {
        for (int h = threadIdx.x; h < num_hyp; h += blockDim.x) {
            // three distinct indices from a counter-based generator (sampling without replacement, sampling.h:77-97)
            int idx[3]; unsigned long long ctr = splitmix64(seed ^ ((unsigned long long)stream_id << 32) ^ (unsigned long long)h);
            for (int i = 0; i < 3; i++) {
                bool dup = true;
                while (dup) { ctr = splitmix64(ctr); idx[i] = (int)((ctr >> 11) % (unsigned long long)n); dup = false; for (int j = 0; j < i; j++) if (idx[j] == idx[i]) dup = true; }
            }
            double u3[9], v3[9];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) { u3[3 * i + k] = su[3 * idx[i] + k]; v3[3 * i + k] = sv[3 * idx[i] + k]; }
            double Es[36];
            const int cnt = spherical_minimal_solver<POLY>(u3, v3, Es);
            for (int m = 0; m < cnt; m++) {
                const double* E = Es + 9 * m;
                double sc = 0.0;
                for (int i = 0; i < n; i++) sc += fmin(sampson_err(E, su + 3 * i, sv + 3 * i), sq_thresh);      // MSAC, ransac.h:295-310
                if (sc < myBest) { myBest = sc; for (int k = 0; k < 9; k++) myE[k] = E[k]; }
            }
        }
    }
    sScore[threadIdx.x] = myBest; sIdx[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { if (sScore[threadIdx.x + s] < sScore[threadIdx.x]) { sScore[threadIdx.x] = sScore[threadIdx.x + s]; sIdx[threadIdx.x] = sIdx[threadIdx.x + s]; } }
        __syncthreads();
    }
    if (threadIdx.x == sIdx[0]) { for (int k = 0; k < 9; k++) bestE[9 * (size_t)pair + k] = myE[k]; bestScore[pair] = myBest; }
}

__global__ void __launch_bounds__(256)
k_ransac_refine(const int* __restrict__ pair_ptr, const double* __restrict__ u, const double* __restrict__ v, double sq_thresh, int inward,
                int min_num_inliers, int do_lsq, int* __restrict__ glists /* [total] scratch: the inlier list of every pair */,
                double* __restrict__ bestE, double* __restrict__ bestScore, double* __restrict__ outR,
                unsigned char* __restrict__ inlier_mask, int* __restrict__ num_inliers) {
    __shared__ double red[28 * 4];
    __shared__ double sh[64];
    __shared__ double bc;
    __shared__ int s_cnt[4];
    const int pair = blockIdx.x;
    const int r0 = pair_ptr[pair], n = pair_ptr[pair + 1] - r0;
    const double* pu = u + (size_t)3 * r0; const double* pv = v + (size_t)3 * r0;
    double E[9]; for (int k = 0; k < 9; k++) E[k] = bestE[9 * (size_t)pair + k];
    const double prev = bestScore[pair];                     // read once, before anything of this pair is written
    const bool have = prev < 1e308 && n >= 3;
    if (have && do_lsq) {
        // final least squares on the inliers of the best model (ransac.h:253-270); kept only if it scores better
        int* list = glists + r0;
        const int ni = block_inlier_list(E, pu, pv, n, sq_thresh, list, s_cnt);
        double E2[9]; for (int k = 0; k < 9; k++) E2[k] = E[k];
        block_sampson_lsq(list, ni, pu, pv, inward != 0, E2, red, sh);
        const double sc = block_msac_score(E2, pu, pv, n, sq_thresh, red, &bc);       // the same value in every thread
        if (sc < prev) {
            for (int k = 0; k < 9; k++) E[k] = E2[k];
            if (threadIdx.x == 0) { bestScore[pair] = sc; for (int k = 0; k < 9; k++) bestE[9 * (size_t)pair + k] = E2[k]; }
        }
    }
    // inlier mask (examples/spherical_sfm_tools.cpp:388-392) and rotation (:410-419)
    double cnt[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const bool in = have && sampson_err(E, pu + 3 * i, pv + 3 * i) < sq_thresh;
        inlier_mask[r0 + i] = in ? 1 : 0; cnt[0] += in ? 1.0 : 0.0;
    }
    block_sum<1>(cnt, red);
    if (threadIdx.x == 0) {
        const int nin = (int)cnt[0]; num_inliers[pair] = nin;
        double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (have && nin > min_num_inliers) { double r[3]; decompose_E_dev(E, inward != 0, r); so3exp(r, Rm); }
        for (int k = 0; k < 9; k++) outR[9 * (size_t)pair + k] = Rm[k];
    }
}

// ---- focal-length search around the pose graph ------------------------------------------------
// One workgroup per trial focal: the loop_constraint_cost_fn of examples/spherical_sfm_tools.cpp:1138-1157 --
//   transform_image_matches (:1118-1132): E_new = T E T, T = diag(f/f0, f/f0, 1); decompose -> r_new -> R = so3exp(r_new)
//   initialize_rotations_sequential (:794-813): chain over the matches (k-1, k); a camera without such a match keeps the identity
//   get_cost (src/uncalibrated_pose_graph.cpp:116-145): 1/2 sum SoftLOne_0.03(|s log(R1 R0^T R^T)|^2), s = 1 / max |log R_rel|
// Lanes share the edges (decomposition, residuals), lane 0 walks the rotation chain.  Per-trial scratch in global memory.
__global__ void __launch_bounds__(256)
k_focal_trials(int n, int E, const int* __restrict__ e0, const int* __restrict__ e1, const int* __restrict__ chain_edge,
               const double* __restrict__ Es /*[E*9] row-major*/, int inward, double focal_guess, const double* __restrict__ focals,
               double* __restrict__ rnew_all /*[trials*E*3]*/, double* __restrict__ x_all /*[trials*n*3]*/,
               double* __restrict__ rot_all /*[trials*n*9] row-major*/, double* __restrict__ costs) {
    __shared__ double red[4]; __shared__ double s_scale;
    const int trial = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double f = focals[trial], tf = f / focal_guess;
    double* rnew = rnew_all + (size_t)trial * E * 3; double* x = x_all + (size_t)trial * n * 3; double* rot = rot_all + (size_t)trial * n * 9;
    double mx = 0.0;
    for (int e = tid; e < E; e += nt) {
        double En[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) En[3 * i + j] = Es[9 * (size_t)e + 3 * i + j] * ((i < 2) ? tf : 1.0) * ((j < 2) ? tf : 1.0);
        double r[3]; decompose_E_dev(En, inward != 0, r);
        // the reference stores so3exp(r_new) and get_cost takes so3ln of it again
        double Rm[9], rr[3]; so3exp(r, Rm); so3ln(Rm, rr);
        rnew[3 * e] = rr[0]; rnew[3 * e + 1] = rr[1]; rnew[3 * e + 2] = rr[2];
        mx = fmax(mx, norm3(rr));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) { double m = 0; for (int w = 0; w < (nt >> 6); w++) m = fmax(m, red[w]); s_scale = 1.0 / m; }
    // chain (lane 0; global writes of this block are visible to it after the barrier above)
    if (tid == 0) {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; k++) rot[k] = R[k];
        for (int idx = 1; idx < n; idx++) {
            const int e = chain_edge[idx];
            double* dst = rot + 9 * (size_t)idx;
            if (e >= 0) {
                double Rm[9], Rn[9]; so3exp(rnew + 3 * e, Rm); mat3_mul(Rm, R, Rn);
                for (int k = 0; k < 9; k++) { R[k] = Rn[k]; dst[k] = Rn[k]; }
            } else { for (int k = 0; k < 9; k++) dst[k] = (k % 4 == 0) ? 1.0 : 0.0; }
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) so3ln(rot + 9 * (size_t)i, x + 3 * i);
    __syncthreads();
    const double scale = s_scale;
    double c = 0.0;
    for (int e = tid; e < E; e += nt) {
        double Rm[9], R0[9], R1[9], A[9], C[9], res[3];
        angle_axis_to_matrix(rnew + 3 * e, Rm); angle_axis_to_matrix(x + 3 * e0[e], R0); angle_axis_to_matrix(x + 3 * e1[e], R1);
        mat3_mul_bt(R1, R0, A); mat3_mul_bt(A, Rm, C);
        matrix_to_angle_axis(C, res);
        const double s2 = scale * scale * (res[0] * res[0] + res[1] * res[1] + res[2] * res[2]);
        double rho0, rho1; robust_loss(2, 0.03, s2, rho0, rho1);
        c += 0.5 * rho0;
    }
    c = wave_sum(c);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int w = 0; w < (nt >> 6); w++) t += red[w]; costs[trial] = t; }
}

}  // namespace ssfm
using namespace ssfm;

static void rm_to_cm(const double* rm, double* cm) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cm[i + 3 * j] = rm[3 * i + j]; }

extern "C" void ssfm_ransac_default_options(ssfm_ransac_options* o) {
    o->num_hypotheses = 1024;          // mode 0 only: fixed budget per pair
    o->seed = 0;                       // RansacOptions::random_seed_
    o->min_num_inliers = 0;            // estimate_pairwise's acceptance test (spherical_sfm_tools.cpp:410)
    o->final_least_squares = 1;        // spherical_sfm_tools.cpp:318
    o->inward = 0;
    o->use_poly_solver = 0;            // estimate_pairwise passes use_poly_solver = false (spherical_sfm_tools.cpp:378)
    o->mode = SSFM_RANSAC_REFERENCE_TRACE;
    o->min_num_iterations = 100; o->max_num_iterations = 10000; o->success_probability = 0.9999;        // RansacOptions, ransac.h:47-60
    o->num_lo_steps = 0; o->num_lsq_iterations = 0;                                                      // spherical_sfm_tools.cpp:316-317
    o->threshold_multiplier = 1.41421356237309504880; o->min_sample_multiplicator = 7; o->non_min_sample_multiplier = 3;   // LORansacOptions, ransac.h:66-73
    o->lo_starting_iterations = 50;
    o->fast_shuffle = 1;
}

namespace ssfm {
int lomsac_launch(ssfm_ctx* ctx, hipStream_t st, int num_pairs, int max_n, const int* d_pair_ptr, const double* d_u, const double* d_v, int total,
                  const ssfm_ransac_options& O, double sq_thresh, const unsigned* d_mt_seeded, int* d_lists, double* d_E, double* d_score, double* d_R,
                  unsigned char* d_mask, int* d_nin, unsigned* d_stats);
bool lomsac_needs_global_lists(int max_n);
}

// Pairs are streamed through the GPU in slabs (BASELINE configs[3]: 2000 frames = 2.0 M pairs x 500 correspondences = 48 GB of rays, more
// than one allocation should hold and far more than one copy should block on): slab k+1 is packed into the second pinned buffer and copied
// on the context's copy stream while slab k computes; results come back per slab.  One slab = at most SLAB_RAYS rays / SLAB_PAIRS pairs.
// indexed input (ssfm_ransac_batch_indexed): rays come from per-frame feature tables through per-pair match lists
struct RansacIndexed { int32_t num_frames; const int32_t* feat_ptr; const double* feat_rays; const int32_t* frame0; const int32_t* frame1; const int32_t* idx0; const int32_t* idx1; };
// one workgroup per pair: u[i] = rays[off0 + idx0[i]], v[i] = rays[off1 + idx1[i]]
static __global__ void __launch_bounds__(256)
k_gather_rays(const int* __restrict__ ptr, const int* __restrict__ off01, const int* __restrict__ idx0, const int* __restrict__ idx1, const double* __restrict__ rays,
              double* __restrict__ u, double* __restrict__ v) {
    const int p = blockIdx.x, a = ptr[p], b = ptr[p + 1];
    const size_t o0 = (size_t)off01[2 * p], o1 = (size_t)off01[2 * p + 1];
    for (int i = a + threadIdx.x; i < b; i += blockDim.x) {
        const double* r0 = rays + 3 * (o0 + (size_t)idx0[i]); const double* r1 = rays + 3 * (o1 + (size_t)idx1[i]);
        u[3 * (size_t)i] = r0[0]; u[3 * (size_t)i + 1] = r0[1]; u[3 * (size_t)i + 2] = r0[2];
        v[3 * (size_t)i] = r1[0]; v[3 * (size_t)i + 1] = r1[1]; v[3 * (size_t)i + 2] = r1[2];
    }
}

static int ransac_batch_impl(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v, double sq_thresh,
                             const ssfm_ransac_options& O, const int32_t* pair_id, double* E_out, double* R_out, uint8_t* inlier_mask,
                             int32_t* num_inliers, double* scores, uint32_t* stats_out, const RansacIndexed* X = nullptr) {
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (O.mode != SSFM_RANSAC_FIXED_BUDGET && O.mode != SSFM_RANSAC_REFERENCE_TRACE) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch: unknown mode");
    if (O.mode == SSFM_RANSAC_REFERENCE_TRACE && (O.min_sample_multiplicator < 1 || O.min_sample_multiplicator > 21 || O.non_min_sample_multiplier < 1 || O.non_min_sample_multiplier > 3 ||
                                                  O.num_lo_steps < 0 || O.num_lsq_iterations < 0 || !(O.success_probability > 0.0 && O.success_probability < 1.0)))
        return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch: LO-RANSAC options out of range (min_sample_multiplicator 1..21, non_min_sample_multiplier 1..3)");
    for (int p = 0; p < num_pairs; p++) if (pair_ptr[p + 1] < pair_ptr[p]) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch: pair_ptr must ascend");
    const bool trace = O.mode == SSFM_RANSAC_REFERENCE_TRACE;
    int max_n = 0; for (int p = 0; p < num_pairs; p++) max_n = std::max(max_n, pair_ptr[p + 1] - pair_ptr[p]);
    const size_t lds_fixed = (size_t)6 * max_n * sizeof(double);
    if (!trace && lds_fixed > 150 * 1024) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch: more than 3200 correspondences in one pair (fixed-budget mode keeps the rays in LDS; use the reference-trace mode)");
    const size_t SLAB_RAYS = getenv("SSFM_RANSAC_SLAB_RAYS") ? (size_t)std::max(atoll(getenv("SSFM_RANSAC_SLAB_RAYS")), 1ll) : ((size_t)4 << 20);   // 4 M rays = 200 MB of u, v per slab
    const size_t stage_threads = getenv("SSFM_RANSAC_STAGE_THREADS") ? (size_t)std::max(1, atoi(getenv("SSFM_RANSAC_STAGE_THREADS"))) : std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency()));
    const int SLAB_PAIRS = getenv("SSFM_RANSAC_SLAB_PAIRS") ? std::max(atoi(getenv("SSFM_RANSAC_SLAB_PAIRS")), 1) : 65536;   // a value <= 0 would never advance the slab loop
    // slab boundaries
    std::vector<int> slab(1, 0);
    for (int p = 0; p < num_pairs;) {
        int q = p; size_t rays = 0;
        while (q < num_pairs && q - p < SLAB_PAIRS && (q == p || rays + (size_t)(pair_ptr[q + 1] - pair_ptr[q]) <= SLAB_RAYS)) { rays += (size_t)(pair_ptr[q + 1] - pair_ptr[q]); q++; }
        slab.push_back(q); p = q;
    }
    const int ns = (int)slab.size() - 1;
    size_t cap_rays = 1; int cap_pairs = 1;
    for (int k = 0; k < ns; k++) { cap_rays = std::max(cap_rays, (size_t)(pair_ptr[slab[k + 1]] - pair_ptr[slab[k]])); cap_pairs = std::max(cap_pairs, slab[k + 1] - slab[k]); }
    hipStream_t st = ctx->stream;
    hipStream_t cs = nullptr; hipEvent_t up_done[2] = {nullptr, nullptr}, compute_done[2] = {nullptr, nullptr};
    hipEvent_t kt0[2] = {nullptr, nullptr}, kt1[2] = {nullptr, nullptr};      // around the kernels of a slab: ssfm_ransac_last_kernel_ms (bench.py: FP64 rate of the scoring)
    ctx->ransac_kernel_ms = 0.0;
    // per-slab device buffers x 2 (upload of the next slab overlaps the kernels of this one); pinned staging x 2
    struct Slot { DevBuf<int> ptr, pid, nin, lists, idx, off; DevBuf<double> u, v, E, S, R; DevBuf<unsigned char> mask; DevBuf<unsigned> stats;
                  double* h_uv = nullptr; int* h_ptr = nullptr; int* h_idx = nullptr; double* h_res = nullptr; unsigned char* h_mask = nullptr; int* h_nin = nullptr; unsigned* h_stats = nullptr; } slot[2];
    DevBuf<unsigned> dmt; DevBuf<double> frays;
    const int nslot = ns > 1 ? 2 : 1;
    const bool glists = trace ? lomsac_needs_global_lists(max_n) : true;
    int rc = SSFM_OK;
    auto body = [&]() -> int {
        if (trace) { std::vector<unsigned> seeded(624); mt_seed_host(O.seed, seeded.data()); SSFM_HIP_CHECK(ctx, upload(dmt, seeded, st)); }
        if (nslot > 1) { SSFM_HIP_CHECK(ctx, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking)); }
        if (X) {                                              // the feature rays of every frame, once
            const size_t nf = (size_t)X->feat_ptr[X->num_frames];
            SSFM_HIP_CHECK(ctx, frays.alloc(std::max<size_t>(1, 3 * nf)));
            if (nf) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(frays.p, X->feat_rays, 3 * nf * sizeof(double), hipMemcpyHostToDevice, st));
            SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));   // (the upload stream of the slabs is another one)
        }
        for (int b = 0; b < nslot; b++) {
            Slot& s = slot[b];
            SSFM_HIP_CHECK(ctx, hipEventCreateWithFlags(&up_done[b], hipEventDisableTiming)); SSFM_HIP_CHECK(ctx, hipEventCreateWithFlags(&compute_done[b], hipEventDisableTiming));
            SSFM_HIP_CHECK(ctx, hipEventCreate(&kt0[b])); SSFM_HIP_CHECK(ctx, hipEventCreate(&kt1[b]));
            SSFM_HIP_CHECK(ctx, s.ptr.alloc(cap_pairs + 1)); SSFM_HIP_CHECK(ctx, s.pid.alloc(cap_pairs)); SSFM_HIP_CHECK(ctx, s.nin.alloc(cap_pairs));
            SSFM_HIP_CHECK(ctx, s.u.alloc(3 * cap_rays)); SSFM_HIP_CHECK(ctx, s.v.alloc(3 * cap_rays)); SSFM_HIP_CHECK(ctx, s.E.alloc((size_t)9 * cap_pairs));
            SSFM_HIP_CHECK(ctx, s.S.alloc(cap_pairs)); SSFM_HIP_CHECK(ctx, s.R.alloc((size_t)9 * cap_pairs)); SSFM_HIP_CHECK(ctx, s.mask.alloc(cap_rays));
            SSFM_HIP_CHECK(ctx, s.stats.alloc((size_t)2 * cap_pairs));
            if (glists) SSFM_HIP_CHECK(ctx, s.lists.alloc((trace ? 2 : 1) * cap_rays));
            if (X) { SSFM_HIP_CHECK(ctx, s.idx.alloc(2 * cap_rays)); SSFM_HIP_CHECK(ctx, s.off.alloc((size_t)2 * cap_pairs));
                     SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&s.h_idx, (2 * cap_rays + (size_t)2 * cap_pairs) * sizeof(int), hipHostMallocDefault)); }
            else SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&s.h_uv, 6 * cap_rays * sizeof(double), hipHostMallocDefault));
            SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&s.h_ptr, (size_t)(2 * cap_pairs + 1) * sizeof(int), hipHostMallocDefault));
            SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&s.h_res, (size_t)19 * cap_pairs * sizeof(double), hipHostMallocDefault));
            SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&s.h_mask, cap_rays, hipHostMallocDefault));
            SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&s.h_nin, (size_t)cap_pairs * sizeof(int), hipHostMallocDefault));
            SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&s.h_stats, (size_t)2 * cap_pairs * sizeof(unsigned), hipHostMallocDefault));
        }
        auto stage = [&](int k) -> int {                      // pack slab k into its slot's pinned buffers and start the copies
            Slot& s = slot[k % nslot]; hipStream_t up = (nslot > 1) ? cs : st;
            const int p0 = slab[k], np = slab[k + 1] - p0, r0 = pair_ptr[p0]; const size_t nr = (size_t)(pair_ptr[slab[k + 1]] - r0);
            if (k >= nslot) SSFM_HIP_CHECK(ctx, hipEventSynchronize(compute_done[k % nslot]));      // the slot's previous slab has been read back
            if (X) {                                          // match lists + the two feature offsets of every pair; the rays are gathered on the device
                // copied AND range-checked here, on the staging threads (a check of the whole list up front would be a second pass over 8 GB at configs[3])
                int* ho = s.h_idx + 2 * cap_rays;
                std::atomic<int> bad(0);
                const int nt = (int)std::max<size_t>(1, std::min<size_t>(stage_threads, nr / (1u << 18)));
                auto part = [&](int t) {
                    const int a = (int)((int64_t)np * t / nt), b = (int)((int64_t)np * (t + 1) / nt);
                    for (int i = a; i < b; i++) {
                        const int f0 = X->frame0[p0 + i], f1 = X->frame1[p0 + i];
                        const int n0 = X->feat_ptr[f0 + 1] - X->feat_ptr[f0], n1 = X->feat_ptr[f1 + 1] - X->feat_ptr[f1];
                        ho[2 * i] = X->feat_ptr[f0]; ho[2 * i + 1] = X->feat_ptr[f1];
                        unsigned worst0 = 0, worst1 = 0;
                        for (int q = pair_ptr[p0 + i]; q < pair_ptr[p0 + i + 1]; q++) {
                            const int i0 = X->idx0[q], i1 = X->idx1[q];
                            s.h_idx[q - r0] = i0; s.h_idx[cap_rays + (q - r0)] = i1;
                            worst0 = std::max(worst0, (unsigned)i0); worst1 = std::max(worst1, (unsigned)i1);      // negative indices wrap to large values
                        }
                        if ((pair_ptr[p0 + i + 1] > pair_ptr[p0 + i]) && (worst0 >= (unsigned)n0 || worst1 >= (unsigned)n1)) bad.store(1);
                    }
                };
                { std::vector<std::thread> th; for (int t = 1; t < nt; t++) th.emplace_back(part, t); part(0); for (auto& x : th) x.join(); }
                if (bad.load()) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed: feature index out of range");
                SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.idx.p, s.h_idx, nr * sizeof(int), hipMemcpyHostToDevice, up));
                SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.idx.p + cap_rays, s.h_idx + cap_rays, nr * sizeof(int), hipMemcpyHostToDevice, up));
                SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.off.p, ho, (size_t)2 * np * sizeof(int), hipMemcpyHostToDevice, up));
            } else
            // pageable -> pinned on several threads (one thread moves ~10 GB/s; configs[3] stages 48 GB)
            {
                const size_t bytes = 3 * nr * sizeof(double);
                const int nt = (int)std::max<size_t>(1, std::min<size_t>(stage_threads, bytes / (4u << 20)));
                auto part = [&](int t) {
                    const size_t a = bytes * t / nt, b = bytes * (t + 1) / nt;
                    std::memcpy((char*)s.h_uv + a, (const char*)(u + (size_t)3 * r0) + a, b - a);
                    std::memcpy((char*)(s.h_uv + 3 * cap_rays) + a, (const char*)(v + (size_t)3 * r0) + a, b - a);
                };
                std::vector<std::thread> th;
                for (int t = 1; t < nt; t++) th.emplace_back(part, t);
                part(0);
                for (auto& x : th) x.join();
            }
            for (int i = 0; i <= np; i++) s.h_ptr[i] = pair_ptr[p0 + i] - r0;
            for (int i = 0; i < np; i++) s.h_ptr[cap_pairs + 1 + i] = pair_id ? pair_id[p0 + i] : p0 + i;      // the random stream of a pair is that of its global index
            if (!X) {
                SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.u.p, s.h_uv, 3 * nr * sizeof(double), hipMemcpyHostToDevice, up));
                SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.v.p, s.h_uv + 3 * cap_rays, 3 * nr * sizeof(double), hipMemcpyHostToDevice, up));
            }
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.ptr.p, s.h_ptr, (size_t)(np + 1) * sizeof(int), hipMemcpyHostToDevice, up));
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.pid.p, s.h_ptr + cap_pairs + 1, (size_t)np * sizeof(int), hipMemcpyHostToDevice, up));
            SSFM_HIP_CHECK(ctx, hipEventRecord(up_done[k % nslot], up));
            return SSFM_OK;
        };
        auto collect = [&](int k) -> int {                    // wait for slab k's results and scatter them to the caller's arrays
            Slot& s = slot[k % nslot];
            const int p0 = slab[k], np = slab[k + 1] - p0, r0 = pair_ptr[p0]; const size_t nr = (size_t)(pair_ptr[slab[k + 1]] - r0);
            SSFM_HIP_CHECK(ctx, hipEventSynchronize(compute_done[k % nslot]));
            { float ms = 0.0f; if (hipEventElapsedTime(&ms, kt0[k % nslot], kt1[k % nslot]) == hipSuccess) ctx->ransac_kernel_ms += ms; else (void)hipGetLastError(); }
            for (int i = 0; i < np; i++) {
                if (E_out) rm_to_cm(s.h_res + 9 * (size_t)i, E_out + 9 * (size_t)(p0 + i));
                if (R_out) rm_to_cm(s.h_res + 9 * (size_t)cap_pairs + 9 * (size_t)i, R_out + 9 * (size_t)(p0 + i));
                if (scores) scores[p0 + i] = s.h_res[18 * (size_t)cap_pairs + i];
                if (num_inliers) num_inliers[p0 + i] = s.h_nin[i];
                if (stats_out) { stats_out[2 * (size_t)(p0 + i)] = s.h_stats[2 * i]; stats_out[2 * (size_t)(p0 + i) + 1] = s.h_stats[2 * i + 1]; }
            }
            if (inlier_mask && nr) std::memcpy(inlier_mask + r0, s.h_mask, nr);
            return SSFM_OK;
        };
        { const int r = stage(0); if (r) return r; }
        for (int k = 0; k < ns; k++) {
            Slot& s = slot[k % nslot];
            const int np = slab[k + 1] - slab[k]; const size_t nr = (size_t)(pair_ptr[slab[k + 1]] - pair_ptr[slab[k]]);
            int slab_max_n = 0; for (int p = slab[k]; p < slab[k + 1]; p++) slab_max_n = std::max(slab_max_n, pair_ptr[p + 1] - pair_ptr[p]);
            SSFM_HIP_CHECK(ctx, hipStreamWaitEvent(st, up_done[k % nslot], 0));
            SSFM_HIP_CHECK(ctx, hipEventRecord(kt0[k % nslot], st));
            if (X && np > 0) hipLaunchKernelGGL(k_gather_rays, dim3(np), dim3(256), 0, st, s.ptr.p, s.off.p, s.idx.p, s.idx.p + cap_rays, frays.p, s.u.p, s.v.p);
            if (trace) {
                const int r = lomsac_launch(ctx, st, np, slab_max_n, s.ptr.p, s.u.p, s.v.p, (int)nr, O, sq_thresh, dmt.p, glists ? s.lists.p : nullptr, s.E.p, s.S.p, s.R.p, s.mask.p, s.nin.p, s.stats.p);
                if (r) return r;
            } else {
                const size_t lds = (size_t)6 * slab_max_n * sizeof(double);
                if (O.use_poly_solver) {
                    if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ransac_hypotheses<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_ransac_hypotheses<true>, dim3(np), dim3(256), lds, st, s.ptr.p, s.u.p, s.v.p, sq_thresh, O.num_hypotheses, (unsigned long long)O.seed, s.pid.p, s.E.p, s.S.p);
                } else {
                    if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ransac_hypotheses<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_ransac_hypotheses<false>, dim3(np), dim3(256), lds, st, s.ptr.p, s.u.p, s.v.p, sq_thresh, O.num_hypotheses, (unsigned long long)O.seed, s.pid.p, s.E.p, s.S.p);
                }
                hipLaunchKernelGGL(k_ransac_refine, dim3(np), dim3(256), 0, st, s.ptr.p, s.u.p, s.v.p, sq_thresh, O.inward, O.min_num_inliers, O.final_least_squares,
                                   s.lists.p, s.E.p, s.S.p, s.R.p, s.mask.p, s.nin.p);
                SSFM_HIP_CHECK(ctx, hipMemsetAsync(s.stats.p, 0, (size_t)2 * np * sizeof(unsigned), st));
            }
            SSFM_HIP_CHECK(ctx, hipEventRecord(kt1[k % nslot], st));
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.h_res, s.E.p, (size_t)9 * np * sizeof(double), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.h_res + 9 * (size_t)cap_pairs, s.R.p, (size_t)9 * np * sizeof(double), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.h_res + 18 * (size_t)cap_pairs, s.S.p, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, st));
            if (nr) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.h_mask, s.mask.p, nr, hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.h_nin, s.nin.p, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(s.h_stats, s.stats.p, (size_t)2 * np * sizeof(unsigned), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipEventRecord(compute_done[k % nslot], st));
            if (k >= 1) { const int r = collect(k - 1); if (r) return r; }        // slab k-1 is scattered while slab k computes ...
            if (k + 1 < ns) { const int r = stage(k + 1); if (r) return r; }      // ... and slab k+1 is packed and copied into the slot k-1 just left
        }
        return collect(ns - 1);
    };
    rc = body();
    if (rc != SSFM_OK) { (void)hipStreamSynchronize(st); if (cs) (void)hipStreamSynchronize(cs); }
    for (int b = 0; b < 2; b++) {
        Slot& s = slot[b];
        s.ptr.free(); s.pid.free(); s.nin.free(); s.lists.free(); s.u.free(); s.v.free(); s.E.free(); s.S.free(); s.R.free(); s.mask.free(); s.stats.free();
        s.idx.free(); s.off.free(); if (s.h_idx) (void)hipHostFree(s.h_idx);
        if (s.h_uv) (void)hipHostFree(s.h_uv); if (s.h_ptr) (void)hipHostFree(s.h_ptr); if (s.h_res) (void)hipHostFree(s.h_res);
        if (s.h_mask) (void)hipHostFree(s.h_mask); if (s.h_nin) (void)hipHostFree(s.h_nin); if (s.h_stats) (void)hipHostFree(s.h_stats);
        if (up_done[b]) (void)hipEventDestroy(up_done[b]); if (compute_done[b]) (void)hipEventDestroy(compute_done[b]);
        if (kt0[b]) (void)hipEventDestroy(kt0[b]); if (kt1[b]) (void)hipEventDestroy(kt1[b]);
    }
    dmt.free(); frays.free();
    if (cs) (void)hipStreamDestroy(cs);
    return rc;
}

extern "C" int ssfm_ransac_last_kernel_ms(ssfm_ctx* ctx, double* ms) {
    if (!ctx || !ms) return SSFM_ERR_INVALID;
    *ms = ctx->ransac_kernel_ms;
    return SSFM_OK;
}

extern "C" int ssfm_ransac_batch_indexed(ssfm_ctx* ctx, int32_t num_frames, const int32_t* feat_ptr, const double* feat_rays, int32_t num_pairs,
                                         const int32_t* pair_frame0, const int32_t* pair_frame1, const int32_t* match_ptr, const int32_t* match_idx0,
                                         const int32_t* match_idx1, double sq_thresh, const ssfm_ransac_options* opt, double* E_out, double* R_out,
                                         uint8_t* inlier_mask, int32_t* num_inliers, double* scores, uint32_t* stats) {
    if (!ctx || num_frames <= 0 || !feat_ptr || !feat_rays || num_pairs <= 0 || !pair_frame0 || !pair_frame1 || !match_ptr || !match_idx0 || !match_idx1)
        return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed: bad arguments");
    for (int f = 0; f < num_frames; f++) if (feat_ptr[f + 1] < feat_ptr[f]) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed: feat_ptr must ascend");
    for (int p = 0; p < num_pairs; p++) {
        const int f0 = pair_frame0[p], f1 = pair_frame1[p];
        if (f0 < 0 || f0 >= num_frames || f1 < 0 || f1 >= num_frames || match_ptr[p + 1] < match_ptr[p]) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed: frame index out of range or match_ptr not ascending");
    }
    ssfm_ransac_options O; if (opt) O = *opt; else ssfm_ransac_default_options(&O);
    const RansacIndexed X{num_frames, feat_ptr, feat_rays, pair_frame0, pair_frame1, match_idx0, match_idx1};
    return ransac_batch_impl(ctx, num_pairs, match_ptr, nullptr, nullptr, sq_thresh, O, nullptr, E_out, R_out, inlier_mask, num_inliers, scores, stats, &X);
}

extern "C" int ssfm_ransac_batch(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v, double sq_thresh,
                                 const ssfm_ransac_options* opt, double* E_out, double* R_out, uint8_t* inlier_mask, int32_t* num_inliers,
                                 double* scores, uint32_t* stats) {
    if (!ctx || !pair_ptr || !u || !v || num_pairs <= 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch: bad arguments");
    ssfm_ransac_options O; if (opt) O = *opt; else ssfm_ransac_default_options(&O);
    return ransac_batch_impl(ctx, num_pairs, pair_ptr, u, v, sq_thresh, O, nullptr, E_out, R_out, inlier_mask, num_inliers, scores, stats);
}

// Every rank enters with the results of ITS pairs (ids = their global indices, lptr = their local CSR) and leaves with all of them: one sum
// all-reduce of a zero-filled table [E 9 | R 9 | score | num_inliers | iterations | LO runs] per pair + the inlier masks packed 32 per double
// (exact: one rank contributes each word) + an error flag, so that a rank whose local batch failed does not leave the others waiting.
static int sharded_exchange(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const std::vector<int>& ids, const std::vector<int>& lptr, int local_rc,
                            const std::vector<double>& lE, const std::vector<double>& lR, const std::vector<double>& lS, const std::vector<uint8_t>& lmask,
                            const std::vector<int>& lnin, const std::vector<uint32_t>& lst, double* E_out, double* R_out, uint8_t* inlier_mask, int32_t* num_inliers,
                            double* scores, uint32_t* stats) {
    const int nl = (int)ids.size();
    const int NT = planner_threads();          // packing / unpacking 500 mask bits per pair is host work of the same order as a rank's GPU time at 8 ranks: fork-join over the pairs
    // result table; every mask word belongs to exactly one pair's rank only if words do not straddle pairs: pack per pair
    std::vector<size_t> wptr(num_pairs + 1, 0);
    for (int p = 0; p < num_pairs; p++) wptr[p + 1] = wptr[p] + (size_t)(pair_ptr[p + 1] - pair_ptr[p] + 31) / 32;
    const size_t per = 22, n_tab = per * num_pairs + wptr[num_pairs] + 1;              // + 1: the error flag
    raw_vector<double> tab(n_tab);
    parallel_chunks((int64_t)n_tab, NT, [&](int, int64_t a0, int64_t a1) { std::memset(tab.data() + a0, 0, (size_t)(a1 - a0) * sizeof(double)); });
    tab[n_tab - 1] = (local_rc != SSFM_OK) ? 1.0 : 0.0;
    if (local_rc == SSFM_OK) parallel_chunks(nl, NT, [&](int, int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            const int p = ids[i]; double* t = &tab[per * (size_t)p];
            std::memcpy(t, &lE[9 * (size_t)i], 9 * sizeof(double)); std::memcpy(t + 9, &lR[9 * (size_t)i], 9 * sizeof(double));
            t[18] = lS[i]; t[19] = (double)lnin[i]; t[20] = (double)lst[2 * (size_t)i]; t[21] = (double)lst[2 * (size_t)i + 1];
            double* w = &tab[per * (size_t)num_pairs + wptr[p]];
            const int n = lptr[i + 1] - lptr[i];
            for (int k = 0; k < n; k += 32) { uint32_t bits = 0; for (int q = 0; q < 32 && k + q < n; q++) bits |= (uint32_t)(lmask[lptr[i] + k + q] != 0) << q; w[k / 32] = (double)bits; }
        }
    });
    const std::string local_err = ctx->err;
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf<double> dtab;
    SSFM_HIP_CHECK(ctx, upload(dtab, tab, ctx->stream));
    { const int rc = ctx_allreduce(ctx, dtab.p, n_tab, ncclSum); if (rc) { dtab.free(); return rc; } }
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(tab.data(), dtab.p, n_tab * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    dtab.free();
    if (local_rc != SSFM_OK) return fail(ctx, local_rc, local_err);
    if (tab[n_tab - 1] != 0.0) return fail(ctx, SSFM_ERR_COMM, "ssfm_ransac_batch_*_sharded: the local batch of another rank failed");
    parallel_chunks(num_pairs, NT, [&](int, int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; p++) {
            const double* t = &tab[per * (size_t)p];
            if (E_out) std::memcpy(E_out + 9 * (size_t)p, t, 9 * sizeof(double));
            if (R_out) std::memcpy(R_out + 9 * (size_t)p, t + 9, 9 * sizeof(double));
            if (scores) scores[p] = t[18];
            if (num_inliers) num_inliers[p] = (int32_t)t[19];
            if (stats) { stats[2 * (size_t)p] = (uint32_t)t[20]; stats[2 * (size_t)p + 1] = (uint32_t)t[21]; }
            if (inlier_mask) {
                const double* w = &tab[per * (size_t)num_pairs + wptr[p]];
                const int n = pair_ptr[p + 1] - pair_ptr[p]; uint8_t* m = inlier_mask + pair_ptr[p];
                for (int k = 0; k < n; k += 32) { const uint32_t bits = (uint32_t)w[k / 32]; const int lim = std::min(32, n - k); for (int q = 0; q < lim; q++) m[k + q] = (uint8_t)((bits >> q) & 1u); }
            }
        }
    });
    return SSFM_OK;
}

// Multi-GPU estimate_pairwise (BASELINE configs[3]): image pairs are independent, so rank r of the context's
// communicator takes pairs r, r + nranks, ... (round robin keeps neighbouring-frame pairs, which have the most correspondences,
// spread over the ranks), runs them as one local batch with the random streams of their global indices, and one sum all-reduce
// of a zero-filled result table hands every rank every pair: [E 9 | R 9 | score | num_inliers | iterations | LO runs] per pair + the
// inlier masks packed 32 per double (exact: one rank contributes each word).  Same results as ssfm_ransac_batch on one GPU, bit for bit.
// Argument checks run on the GLOBAL pair list before it is sharded, and a rank whose local batch fails still enters the collective
// (carrying an error flag in the table), so that every rank returns the error instead of some of them waiting for it forever.
extern "C" int ssfm_ransac_batch_sharded(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v,
                                         double sq_thresh, const ssfm_ransac_options* opt, double* E_out, double* R_out,
                                         uint8_t* inlier_mask, int32_t* num_inliers, double* scores, uint32_t* stats) {
    if (!ctx || !pair_ptr || !u || !v || num_pairs <= 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_sharded: bad arguments");
    ssfm_ransac_options O; if (opt) O = *opt; else ssfm_ransac_default_options(&O);
    const int nr = ctx->collective ? ctx->nranks : 1, rk = ctx->collective ? ctx->rank : 0;
    if (nr == 1 && !ctx->collective) return ransac_batch_impl(ctx, num_pairs, pair_ptr, u, v, sq_thresh, O, nullptr, E_out, R_out, inlier_mask, num_inliers, scores, stats);
    // the checks every rank must agree on
    for (int p = 0; p < num_pairs; p++) {
        const int c = pair_ptr[p + 1] - pair_ptr[p];
        if (c < 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_sharded: pair_ptr must ascend");
        if (O.mode == SSFM_RANSAC_FIXED_BUDGET && (size_t)6 * c * sizeof(double) > 150 * 1024) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_sharded: more than 3200 correspondences in one pair (fixed-budget mode)");
    }
    // local batch
    std::vector<int> ids, lptr(1, 0);
    for (int p = rk; p < num_pairs; p += nr) { ids.push_back(p); lptr.push_back(lptr.back() + pair_ptr[p + 1] - pair_ptr[p]); }
    const int nl = (int)ids.size(), ltotal = lptr.back();
    std::vector<double> lu((size_t)3 * ltotal), lv((size_t)3 * ltotal), lE((size_t)9 * nl), lR((size_t)9 * nl), lS(nl);
    std::vector<uint8_t> lmask(ltotal); std::vector<int> lnin(nl); std::vector<uint32_t> lst((size_t)2 * nl);
    for (int i = 0; i < nl; i++) {
        const size_t src = (size_t)3 * pair_ptr[ids[i]], cnt = (size_t)3 * (lptr[i + 1] - lptr[i]);
        if (cnt) { std::memcpy(&lu[(size_t)3 * lptr[i]], u + src, cnt * sizeof(double)); std::memcpy(&lv[(size_t)3 * lptr[i]], v + src, cnt * sizeof(double)); }
    }
    int local_rc = SSFM_OK;
    if (nl > 0) local_rc = ransac_batch_impl(ctx, nl, lptr.data(), lu.data(), lv.data(), sq_thresh, O, ids.data(), lE.data(), lR.data(), lmask.data(), lnin.data(), lS.data(), lst.data());
    return sharded_exchange(ctx, num_pairs, pair_ptr, ids, lptr, local_rc, lE, lR, lS, lmask, lnin, lst, E_out, R_out, inlier_mask, num_inliers, scores, stats);
}

// The same for the indexed form (per-frame feature rays + per-pair match lists: what estimate_pairwise holds, and 6x less host-to-device traffic):
// every rank uploads the feature rays of all frames (2000 frames x 1000 rays = 48 MB at BASELINE configs[3]) and the match lists of ITS pairs.
extern "C" int ssfm_ransac_batch_indexed_sharded(ssfm_ctx* ctx, int32_t num_frames, const int32_t* feat_ptr, const double* feat_rays, int32_t num_pairs,
                                                 const int32_t* pair_frame0, const int32_t* pair_frame1, const int32_t* match_ptr, const int32_t* match_idx0,
                                                 const int32_t* match_idx1, double sq_thresh, const ssfm_ransac_options* opt, double* E_out, double* R_out,
                                                 uint8_t* inlier_mask, int32_t* num_inliers, double* scores, uint32_t* stats) {
    if (!ctx) return SSFM_ERR_INVALID;
    const int nr = ctx->collective ? ctx->nranks : 1, rk = ctx->collective ? ctx->rank : 0;
    if (nr == 1 && !ctx->collective)
        return ssfm_ransac_batch_indexed(ctx, num_frames, feat_ptr, feat_rays, num_pairs, pair_frame0, pair_frame1, match_ptr, match_idx0, match_idx1, sq_thresh, opt, E_out, R_out,
                                         inlier_mask, num_inliers, scores, stats);
    if (num_frames <= 0 || !feat_ptr || !feat_rays || num_pairs <= 0 || !pair_frame0 || !pair_frame1 || !match_ptr || !match_idx0 || !match_idx1)
        return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed_sharded: bad arguments");
    ssfm_ransac_options O; if (opt) O = *opt; else ssfm_ransac_default_options(&O);
    // the checks every rank must agree on (the global list)
    for (int f = 0; f < num_frames; f++) if (feat_ptr[f + 1] < feat_ptr[f]) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed_sharded: feat_ptr must ascend");
    for (int p = 0; p < num_pairs; p++) {
        const int f0 = pair_frame0[p], f1 = pair_frame1[p], c = match_ptr[p + 1] - match_ptr[p];
        if (f0 < 0 || f0 >= num_frames || f1 < 0 || f1 >= num_frames || c < 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed_sharded: frame index out of range or match_ptr not ascending");
        if (O.mode == SSFM_RANSAC_FIXED_BUDGET && (size_t)6 * c * sizeof(double) > 150 * 1024) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_indexed_sharded: more than 3200 correspondences in one pair (fixed-budget mode)");
    }
    std::vector<int> ids, lptr(1, 0), lf0, lf1;
    for (int p = rk; p < num_pairs; p += nr) { ids.push_back(p); lptr.push_back(lptr.back() + match_ptr[p + 1] - match_ptr[p]); lf0.push_back(pair_frame0[p]); lf1.push_back(pair_frame1[p]); }
    const int nl = (int)ids.size(), ltotal = lptr.back();
    std::vector<int> li0(std::max(ltotal, 1)), li1(std::max(ltotal, 1));
    for (int i = 0; i < nl; i++) {
        const size_t src = (size_t)match_ptr[ids[i]], cnt = (size_t)(lptr[i + 1] - lptr[i]);
        if (cnt) { std::memcpy(&li0[lptr[i]], match_idx0 + src, cnt * sizeof(int)); std::memcpy(&li1[lptr[i]], match_idx1 + src, cnt * sizeof(int)); }
    }
    std::vector<double> lE((size_t)9 * nl), lR((size_t)9 * nl), lS(nl);
    std::vector<uint8_t> lmask(std::max(ltotal, 1)); std::vector<int> lnin(nl); std::vector<uint32_t> lst((size_t)2 * nl);
    int local_rc = SSFM_OK;
    if (nl > 0) {
        const RansacIndexed X{num_frames, feat_ptr, feat_rays, lf0.data(), lf1.data(), li0.data(), li1.data()};
        local_rc = ransac_batch_impl(ctx, nl, lptr.data(), nullptr, nullptr, sq_thresh, O, ids.data(), lE.data(), lR.data(), lmask.data(), lnin.data(), lS.data(), lst.data(), &X);
    }
    return sharded_exchange(ctx, num_pairs, match_ptr, ids, lptr, local_rc, lE, lR, lS, lmask, lnin, lst, E_out, R_out, inlier_mask, num_inliers, scores, stats);
}

// parity probe: the minimal solver on given 3-point samples.  Es: [S*36] (4 column-major 3x3 per sample), counts: [S]
static int solver_probe(ssfm_ctx* ctx, bool poly, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples, double* Es, int32_t* counts) {
    if (!ctx || S <= 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_spherical_solver_probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<double> du, dv, dE; DevBuf<int> ds, dc;
    std::vector<double> hu(u, u + (size_t)3 * n), hv(v, v + (size_t)3 * n); std::vector<int> hs(samples, samples + (size_t)3 * S);
    SSFM_HIP_CHECK(ctx, upload(du, hu, st)); SSFM_HIP_CHECK(ctx, upload(dv, hv, st)); SSFM_HIP_CHECK(ctx, upload(ds, hs, st));
    SSFM_HIP_CHECK(ctx, dE.alloc((size_t)36 * S)); SSFM_HIP_CHECK(ctx, dc.alloc(S));
    if (poly) hipLaunchKernelGGL(k_solver_probe<true>, dim3((S + 63) / 64), dim3(64), 0, st, S, ds.p, du.p, dv.p, dE.p, dc.p);
    else hipLaunchKernelGGL(k_solver_probe<false>, dim3((S + 63) / 64), dim3(64), 0, st, S, ds.p, du.p, dv.p, dE.p, dc.p);
    std::vector<double> hE((size_t)36 * S);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hE.data(), dE.p, hE.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(counts, dc.p, S * sizeof(int), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (int s = 0; s < S; s++) for (int m = 0; m < 4; m++) rm_to_cm(&hE[36 * (size_t)s + 9 * m], Es + 36 * (size_t)s + 9 * m);
    du.free(); dv.free(); dE.free(); ds.free(); dc.free();
    return SSFM_OK;
}
extern "C" int ssfm_spherical_solver_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples,
                                           double* Es, int32_t* counts) { return solver_probe(ctx, false, n, u, v, S, samples, Es, counts); }
extern "C" int ssfm_spherical_solver_poly_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples,
                                                double* Es, int32_t* counts) { return solver_probe(ctx, true, n, u, v, S, samples, Es, counts); }

// find_best_focal_length_random without its random draw (examples/spherical_sfm_tools.cpp:1418-1496): the caller supplies the
// trial focals; costs[t] = loop_constraint_cost_fn(focals[t]); best_trial = first minimum; rotations_best = the sequential
// initialisation at that focal (column-major 3x3 per camera), ready for ssfm_posegraph_focal_solve (run_optimization, :1160-1188).
extern "C" int ssfm_focal_search(ssfm_ctx* ctx, int32_t n, int32_t E, const int32_t* index0, const int32_t* index1, const double* rel_rotations,
                                 int32_t inward, double focal_guess, int32_t num_trials, const double* focals, double* costs,
                                 int32_t* best_trial, double* rotations_best, double* rel_rotations_best) {
    if (!ctx || n <= 0 || E <= 0 || num_trials <= 0 || !index0 || !index1 || !rel_rotations || !focals)
        return fail(ctx, SSFM_ERR_INVALID, "ssfm_focal_search: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // Es[i] = make_spherical_essential_matrix(R_i, inward) (:1429-1433), row-major for the device
    std::vector<double> Es((size_t)9 * E); std::vector<int> e0(index0, index0 + E), e1(index1, index1 + E), chain(n, -1);
    for (int e = 0; e < E; e++) {
        double Rm[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rm[3 * i + j] = rel_rotations[9 * (size_t)e + i + 3 * j];
        double t[3] = {Rm[2], Rm[5], Rm[8] - 1.0}; if (inward) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
        double* Em = &Es[9 * (size_t)e];
        for (int j = 0; j < 3; j++) { Em[j] = t[1] * Rm[6 + j] - t[2] * Rm[3 + j]; Em[3 + j] = t[2] * Rm[j] - t[0] * Rm[6 + j]; Em[6 + j] = t[0] * Rm[3 + j] - t[1] * Rm[j]; }
        if (index1[e] >= 1 && index1[e] < n && index0[e] == index1[e] - 1 && chain[index1[e]] < 0) chain[index1[e]] = e;      // first match (k-1, k), :804-810
    }
    DevBuf<double> dEs, dF, dR, dX, dRot, dC; DevBuf<int> de0, de1, dch;
    std::vector<double> fv(focals, focals + num_trials);
    SSFM_HIP_CHECK(ctx, upload(dEs, Es, st)); SSFM_HIP_CHECK(ctx, upload(dF, fv, st)); SSFM_HIP_CHECK(ctx, upload(de0, e0, st));
    SSFM_HIP_CHECK(ctx, upload(de1, e1, st)); SSFM_HIP_CHECK(ctx, upload(dch, chain, st));
    SSFM_HIP_CHECK(ctx, dR.alloc((size_t)num_trials * E * 3)); SSFM_HIP_CHECK(ctx, dX.alloc((size_t)num_trials * n * 3));
    SSFM_HIP_CHECK(ctx, dRot.alloc((size_t)num_trials * n * 9)); SSFM_HIP_CHECK(ctx, dC.alloc(num_trials));
    hipLaunchKernelGGL(k_focal_trials, dim3(num_trials), dim3(256), 0, st, n, E, de0.p, de1.p, dch.p, dEs.p, inward, focal_guess, dF.p, dR.p, dX.p, dRot.p, dC.p);
    std::vector<double> hc(num_trials);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), dC.p, hc.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    int best = 0; for (int t = 1; t < num_trials; t++) if (hc[t] < hc[best]) best = t;                 // :1467-1474 (strict <, first minimum)
    if (costs) std::memcpy(costs, hc.data(), hc.size() * sizeof(double));
    if (best_trial) *best_trial = best;
    if (rotations_best) {
        std::vector<double> hr((size_t)n * 9);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), dRot.p + (size_t)best * n * 9, hr.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        for (int i = 0; i < n; i++) rm_to_cm(&hr[9 * (size_t)i], rotations_best + 9 * (size_t)i);
    }
    if (rel_rotations_best) {                              // the matches as transform_image_matches leaves them at the best focal
        std::vector<double> hr((size_t)E * 3);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), dR.p + (size_t)best * E * 3, hr.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        for (int e = 0; e < E; e++) { double Rm[9]; so3exp(&hr[3 * (size_t)e], Rm); rm_to_cm(Rm, rel_rotations_best + 9 * (size_t)e); }
    }
    dEs.free(); dF.free(); dR.free(); dX.free(); dRot.free(); dC.free(); de0.free(); de1.free(); dch.free();
    return SSFM_OK;
}
