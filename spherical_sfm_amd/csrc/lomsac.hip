// spherical_sfm_amd -- LocallyOptimizedMSAC with the reference's own sample trace, one workgroup per image pair.
//
//   ransac_lib::LocallyOptimizedMSAC<Matrix3d, ..., SphericalEstimator>::EstimateModel      include/RansacLib/ransac.h:128-275
//   LocalOptimization / LeastSquaresFit / GetInliers / ScoreModel / UpdateBestModel          include/RansacLib/ransac.h:277-428
//   UniformSampling (std::mt19937 seeded with random_seed_, DrawSample / ShuffleSample)      include/RansacLib/sampling.h:46-135
//   utils::RandomShuffleAndResize, utils::NumRequiredIterations                              include/RansacLib/utils.h:48-140
//   SphericalEstimator::{MinimalSolver, NonMinimalSolver, EvaluateModelOnPoint, LeastSquares} src/spherical_estimator.cpp:67-157
//   the per-pair tail of estimate_pairwise (inlier flags, acceptance, Decompose)            examples/spherical_sfm_tools.cpp:378-419
//
// The fixed-budget kernels of ransac.hip replace the reference's sequential sampling by a parallel one and agree with it
// statistically.  This kernel keeps the reference's control flow instead, draw for draw: both std::mt19937 streams (the sampler's and
// the local optimisation's) and libstdc++'s uniform_int_distribution are restated on the device, so a pair runs the same minimal
// samples, crowns the same best-so-far models at the same iterations, shuffles the same inlier lists into the same least-squares
// subsets and stops after the same number of iterations as a CPU build of the reference -- what differs is floating-point rounding.
// What is parallel: the rays sit in LDS; a chunk of iterations is evaluated at once, one lane per iteration (the sampler's stream does not
// depend on the models, so its draws can run ahead; the models of a lane are scored against every ray); the control flow then walks the
// chunk in order, and the rare events (a new best minimal model, the local optimisation at iteration lo_starting_iterations_, the
// final least squares) are workgroup-cooperative: ordered inlier compaction, block-wide scores, a block-wide Levenberg-Marquardt.
// 128 threads per pair: RansacLib never stops before min_num_iterations_ = 100 and usually stops there, so the first chunk is those
// 100 iterations.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include "ransac_device.h"
#include "sampson_lsq.h"

namespace ssfm {

constexpr int LO_T = 128;          // threads per pair
constexpr int LO_FIFO = 3 * LO_T + 64;   // pre-drawn sampler indices (three per iteration + spare for repeated indices)

struct LoOpts {
    double sq_thresh, thresh_mult, success_prob;
    unsigned min_it, max_it, lo_start;
    int num_lo_steps, num_lsq_it, min_sample_mult, non_min_mult, final_lsq, inward, min_num_inliers, fast_shuffle;
};

__device__ __forceinline__ unsigned num_required_iterations(double ratio, double pmiss, int ssize, unsigned mn, unsigned mx) {   // utils.h:110-140
    if (ratio <= 0.0) return mx;
    if (ratio >= 1.0) return mn;
    const double pn = 1.0 - pow(ratio, (double)ssize);
    if (pn >= 0.99999999999999) return mx;
    const double it = ceil(log(pmiss) / log(pn) + 0.5);
    const unsigned r = (it >= 4294967295.0) ? 4294967295u : (unsigned)it;
    return max(mn, min(r, mx));
}

// SphericalEstimator::NonMinimalSolver (src/spherical_estimator.cpp:86-108): the action-matrix solver on 4..9 rays, the candidate with
// the least Sampson sum over the sample wins (first of equals).  One thread runs it.
__device__ int nonminimal_solver_dev(const int* sample, int ns, const double* pu, const double* pv, double* Eout) {
    double uN[27], vN[27];
    for (int i = 0; i < 9; i++) for (int k = 0; k < 3; k++) { const int q = (i < ns) ? sample[i] : sample[0]; uN[3 * i + k] = pu[3 * q + k]; vN[3 * i + k] = pv[3 * q + k]; }
    double B[6][3], Es[36];
    spherical_nullspace<9>(uN, vN, ns, B);
    const int cnt = spherical_models_from_basis<false, true>(B, Es);
    if (cnt == 0) return 0;
    double best = INFINITY; int bi = 0;
    for (int m = 0; m < cnt; m++) { double sc = 0; for (int j = 0; j < ns; j++) sc += sampson_err(Es + 9 * m, pu + 3 * sample[j], pv + 3 * sample[j]); if (sc < best) { best = sc; bi = m; } }
    for (int k = 0; k < 9; k++) Eout[k] = Es[9 * bi + k];
    return 1;
}

struct LoShared {                    // static LDS of the trace kernel
    double score[LO_T]; double red[10 * (LO_T / 64)]; double sh[64]; double bc; double E[9];
    int sample[3 * LO_T]; int nm[LO_T]; int cnt[LO_T / 64]; int dr[64]; int flag;
};

// everything a pair's control flow needs, identical in every thread
struct LoState {
    const double* pu; const double* pv; int n; int* listA; int* listB; unsigned* mtR; int posR; LoShared* S; LoOpts o;
};

// LeastSquares on the first wave (ransac_device.h: wave_sampson_lsq), the result handed to the other wave through LDS
__device__ void lo_wave_lsq(const int* list, int cnt, const double* pu, const double* pv, bool inward, double* E, LoShared* S) {
    __syncthreads();                                           // the list is complete and nobody reads S->E any more
    if (threadIdx.x < 64) {
        wave_sampson_lsq(list, cnt, pu, pv, inward, E);
        if (threadIdx.x == 0) for (int k = 0; k < 9; k++) S->E[k] = E[k];
    }
    __syncthreads();
    for (int k = 0; k < 9; k++) E[k] = S->E[k];
    __syncthreads();
}

// LeastSquaresFit (ransac.h:409-420)
__device__ void lo_lsq_fit(LoState& st, double thresh, double* model) {
    const int ni = block_inlier_list(model, st.pu, st.pv, st.n, thresh, st.listB, st.S->cnt);
    if (ni < 3) return;
    const int sz = min(st.o.min_sample_mult * 3, ni);
    block_shuffle_resize(st.listB, ni, sz, st.mtR, st.posR, st.o.fast_shuffle != 0, st.S->dr, &st.S->flag);
    lo_wave_lsq(st.listB, sz, st.pu, st.pv, st.o.inward != 0, model, st.S);
}
__device__ __forceinline__ void lo_update(double sc, const double* m, double* best_sc, double* best) { if (sc < *best_sc) { *best_sc = sc; for (int k = 0; k < 9; k++) best[k] = m[k]; } }

// LocalOptimization (ransac.h:341-407)
__device__ void lo_local_optimization(LoState& st, double* best_min, double* score_best) {
    if (4 > st.n) return;                                                         // non_minimal_sample_size() > num_data
    const double thr = st.o.sq_thresh, mult = st.o.thresh_mult;
    double m_init[9]; for (int k = 0; k < 9; k++) m_init[k] = best_min[k];
    lo_lsq_fit(st, thr * mult, m_init);
    lo_update(block_msac_score(m_init, st.pu, st.pv, st.n, thr, st.S->red, &st.S->bc), m_init, score_best, best_min);
    if (st.o.num_lo_steps <= 0) return;                                           // (the base inlier set is only used by the LO steps)
    const int nb = block_inlier_list(m_init, st.pu, st.pv, st.n, thr * mult, st.listA, st.S->cnt);
    const int nonmin = max(4, min(3 * st.o.non_min_mult, nb / 2));
    for (int r = 0; r < st.o.num_lo_steps; r++) {
        for (int i = threadIdx.x; i < nb; i += blockDim.x) st.listB[i] = st.listA[i];
        __syncthreads();
        // RandomShuffleAndResize(nonmin, rng, &sample): resize() pads with zeros when the base set is smaller than the sample
        block_shuffle_resize(st.listB, nb, min(nonmin, nb), st.mtR, st.posR, st.o.fast_shuffle != 0, st.S->dr, &st.S->flag);
        const int ns = min(nonmin, 9);
        if (threadIdx.x == 0) {
            int smp[9]; for (int i = 0; i < ns; i++) smp[i] = (i < nb) ? st.listB[i] : 0;
            double Em[9]; const int ok = nonminimal_solver_dev(smp, ns, st.pu, st.pv, Em);
            st.S->flag = ok; for (int k = 0; k < 9; k++) st.S->E[k] = Em[k];
        }
        __syncthreads();
        const int ok = st.S->flag; double m[9]; for (int k = 0; k < 9; k++) m[k] = st.S->E[k];
        __syncthreads();
        if (!ok) continue;
        lo_update(block_msac_score(m, st.pu, st.pv, st.n, thr, st.S->red, &st.S->bc), m, score_best, best_min);
        lo_lsq_fit(st, thr, m);
        double th = mult * thr; const double upd = (mult - 1.0) * thr / (double)(st.o.num_lsq_it - 1);
        for (int i = 0; i < st.o.num_lsq_it; i++) {
            lo_lsq_fit(st, th, m);
            lo_update(block_msac_score(m, st.pu, st.pv, st.n, thr, st.S->red, &st.S->bc), m, score_best, best_min);
            th -= upd;
        }
    }
}

template <bool POLY, bool RAYS_LDS>
__global__ void __launch_bounds__(LO_T, 2)
k_lomsac_trace(const int* __restrict__ pair_ptr, const double* __restrict__ gu, const double* __restrict__ gv, LoOpts o,
               const unsigned* __restrict__ mt_seeded, int* __restrict__ glists /* RAYS_LDS ? unused : [2 * total] */,
               double* __restrict__ outE, double* __restrict__ outScore, double* __restrict__ outR, unsigned char* __restrict__ inlier_mask,
               int* __restrict__ num_inliers, unsigned* __restrict__ stats /* [pairs*2] iterations, LO runs; or null */) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ LoShared S;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int r0 = pair_ptr[pair], n = pair_ptr[pair + 1] - r0;
    const double MAXD = 1.79769313486231570815e308;
    // dynamic LDS: [rays u | rays v | listA | listB] (RAYS_LDS) then [mtS | mtR | fifo]
    double* su = lds; double* sv = lds + (size_t)3 * n;
    int* lA = reinterpret_cast<int*>(lds + (size_t)6 * n); int* lB = lA + n;
    unsigned* mtS = RAYS_LDS ? reinterpret_cast<unsigned*>(lB + n + (n & 1)) : reinterpret_cast<unsigned*>(lds);
    unsigned* mtR = mtS + 624; int* fifo = reinterpret_cast<int*>(mtR + 624);
    const double* pu; const double* pv; int* listA; int* listB;
    if (RAYS_LDS) {
        for (int i = tid; i < 3 * n; i += LO_T) { su[i] = gu[(size_t)3 * r0 + i]; sv[i] = gv[(size_t)3 * r0 + i]; }
        pu = su; pv = sv; listA = lA; listB = lB;
    } else { pu = gu + (size_t)3 * r0; pv = gv + (size_t)3 * r0; listA = glists + (size_t)2 * r0; listB = listA + n; }
    for (int i = tid; i < 624; i += LO_T) { const unsigned w = mt_seeded[i]; mtS[i] = w; mtR[i] = w; }
    __syncthreads();

    double best_model[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double best_score = MAXD; int best_num_inliers = 0; unsigned it = 0, lo_count = 0;
    if (n >= 3) {                                                                      // ransac.h:137-141
        LoState st; st.pu = pu; st.pv = pv; st.n = n; st.listA = listA; st.listB = listB; st.mtR = mtR; st.posR = 624; st.S = &S; st.o = o;
        int posS = 624, fifo_head = 0, fifo_cnt = 0;
        const bool draw = ((double)n / (double)(n - 3)) < 2.71828182845904523536;       // DrawBetterThanShuffle, sampling.h:66-75
        unsigned max_it = max(o.max_it, o.min_it);
        double best_min[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; double best_min_score = MAXD;
        auto refresh_inliers = [&]() {
            // GetInliers(best_model) -> best_num_inliers, inlier_ratio -> max_num_iterations  (ransac.h:169-176, 229-236)
            double c[1] = {0.0};
            for (int i = tid; i < n; i += LO_T) c[0] += (sampson_err(best_model, pu + 3 * i, pv + 3 * i) < o.sq_thresh) ? 1.0 : 0.0;
            block_sum<1>(c, S.red);
            if (tid == 0) S.bc = c[0];
            __syncthreads();
            best_num_inliers = (int)S.bc;
            __syncthreads();
        };
        bool done = false;
        while (!done && it < max_it) {
            // ---- chunk of iterations [it, it + cnt)
            unsigned cnt = min((unsigned)LO_T, max_it - it);
            if (it < o.min_it) cnt = min(cnt, o.min_it - it);                           // never fewer than min_num_iterations_ are run
            // phase A: the minimal samples of the chunk, in order (the sampler's stream is independent of everything else)
            if (draw) {
                for (unsigned c = 0; c < cnt; c++) {
                    int smp[3];
                    for (int i = 0; i < 3; i++) {
                        bool found = true;
                        while (found) {
                            if (fifo_head >= fifo_cnt) {
                                // refill: temper the next words of the stream in parallel; -1 marks a Lemire rejection (the draw is repeated)
                                __syncthreads();                       // every thread has read the last entry before it is overwritten
                                fifo_head = 0; fifo_cnt = 0;
                                while (fifo_cnt < LO_FIFO) {
                                    if (posS >= 624) { mt_twist(mtS); posS = 0; }
                                    const int seg = min(LO_FIFO - fifo_cnt, 624 - posS);
                                    for (int j = tid; j < seg; j += LO_T) { unsigned r; const bool ok = lemire_accept(mt_temper(mtS[posS + j]), (unsigned)n, &r); fifo[fifo_cnt + j] = ok ? (int)r : -1; }
                                    posS += seg; fifo_cnt += seg;
                                }
                                __syncthreads();
                            }
                            const int d = fifo[fifo_head++];
                            if (d < 0) continue;
                            smp[i] = d; found = false;
                            for (int j = 0; j < i; j++) if (smp[j] == d) { found = true; break; }
                        }
                    }
                    if (tid == 0) { S.sample[3 * c] = smp[0]; S.sample[3 * c + 1] = smp[1]; S.sample[3 * c + 2] = smp[2]; }
                }
            } else {
                // ShuffleSample (sampling.h:104-124): n = 3 takes (0,1,2) without a draw, n = 4 shuffles (0,1,2,3) and keeps three
                for (unsigned c = 0; c < cnt; c++) {
                    int p[4] = {0, 1, 2, 3};
                    if (n != 3) for (int i = 0; i < n - 1; i++) { const int idx = mt_uniform_int(mtS, posS, i, n - 1); const int t = p[i]; p[i] = p[idx]; p[idx] = t; }
                    if (tid == 0) { S.sample[3 * c] = p[0]; S.sample[3 * c + 1] = p[1]; S.sample[3 * c + 2] = p[2]; }
                }
            }
            __syncthreads();
            // phase B: one lane per iteration -- MinimalSolver + GetBestEstimatedModelId (ransac.h:184-195, 277-293)
            double myE[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; double myScore = MAXD; int myNm = 0;
            if ((unsigned)tid < cnt) {
                double u3[9], v3[9];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int q = S.sample[3 * tid + i];
#pragma unroll
                    for (int k = 0; k < 3; k++) { u3[3 * i + k] = pu[3 * q + k]; v3[3 * i + k] = pv[3 * q + k]; }
                }
                double B[6][3], Es[36];
                spherical_nullspace<3>(u3, v3, 3, B);
                myNm = spherical_models_from_basis<POLY, true>(B, Es);
                if (myNm == 4) {
                    // ScoreModel of the four candidates in ONE sweep over the rays: a ray is read from LDS once (a broadcast read) and
                    // the four independent chains fill the FP64 pipe; the quotient by hardware reciprocal + two Newton steps
                    double sc[4] = {0.0, 0.0, 0.0, 0.0};
                    for (int i = 0; i < n; i++) {
                        const double u0 = pu[3 * i], u1 = pu[3 * i + 1], u2 = pu[3 * i + 2], v0 = pv[3 * i], v1 = pv[3 * i + 1], v2 = pv[3 * i + 2];
#pragma unroll
                        for (int m = 0; m < 4; m++) {
                            const double* E = Es + 9 * m;
                            const double e0 = E[0] * u0 + E[1] * u1 + E[2] * u2, e1 = E[3] * u0 + E[4] * u1 + E[5] * u2, e2 = E[6] * u0 + E[7] * u1 + E[8] * u2;
                            const double f0 = E[0] * v0 + E[3] * v1 + E[6] * v2, f1 = E[1] * v0 + E[4] * v1 + E[7] * v2;
                            const double d = v0 * e0 + v1 * e1 + v2 * e2;
                            sc[m] += fmin((d * d) * fast_rcp(e0 * e0 + e1 * e1 + f0 * f0 + f1 * f1), o.sq_thresh);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < 4; m++) if (sc[m] < myScore) { myScore = sc[m]; for (int k = 0; k < 9; k++) myE[k] = Es[9 * m + k]; }
                }
            }
            S.score[tid] = myScore; S.nm[tid] = myNm;
            __syncthreads();
            // phase C: the control flow of EstimateModel over the chunk, in order
            for (unsigned c = 0; c < cnt; c++) {
                if (it >= max_it) { done = true; break; }
                if (it == o.lo_start && best_min_score < MAXD) {                        // ransac.h:160-177
                    ++lo_count;
                    lo_local_optimization(st, best_model, &best_score);
                    refresh_inliers();
                    max_it = num_required_iterations((double)best_num_inliers / (double)n, 1.0 - o.success_prob, 3, o.min_it, o.max_it);
                }
                const int nm = S.nm[c]; const double bl = S.score[c];
                if (nm > 0 && (bl < best_min_score || it == o.lo_start)) {              // ransac.h:197-237
                    const bool best_min_model = bl < best_min_score;
                    __syncthreads();
                    if (best_min_model) {
                        if ((unsigned)tid == c) for (int k = 0; k < 9; k++) S.E[k] = myE[k];
                        __syncthreads();
                        best_min_score = bl; for (int k = 0; k < 9; k++) best_min[k] = S.E[k];
                        lo_update(best_min_score, best_min, &best_score, best_model);
                    }
                    __syncthreads();
                    const bool run_lo = (it >= o.lo_start && best_min_score < MAXD);
                    if (best_min_model || run_lo) {
                        if (run_lo) {
                            ++lo_count;
                            double sc = best_min_score;
                            lo_local_optimization(st, best_min, &sc);
                            lo_update(sc, best_min, &best_score, best_model);
                        }
                        refresh_inliers();
                        max_it = num_required_iterations((double)best_num_inliers / (double)n, 1.0 - o.success_prob, 3, o.min_it, o.max_it);
                    }
                }
                ++it;
            }
            __syncthreads();
        }
        if (it <= o.lo_start && best_score < MAXD) {                                    // ransac.h:241-251
            ++lo_count;
            lo_local_optimization(st, best_model, &best_score);
            refresh_inliers();
        }
        if (o.final_lsq) {                                                              // ransac.h:253-270
            double refined[9]; for (int k = 0; k < 9; k++) refined[k] = best_model[k];
            // stats.inlier_indices is GetInliers(best_model) of the last update; LeastSquares on an empty list still rebuilds E from r
            const int ni = block_inlier_list(best_model, pu, pv, n, o.sq_thresh, listB, S.cnt);
            lo_wave_lsq(listB, ni, pu, pv, o.inward != 0, refined, &S);
            const double sc = block_msac_score(refined, pu, pv, n, o.sq_thresh, S.red, &S.bc);
            if (sc < best_score) { best_score = sc; for (int k = 0; k < 9; k++) best_model[k] = refined[k]; refresh_inliers(); }
        }
    }
    // ---- estimate_pairwise's tail: inlier flags of E (spherical_sfm_tools.cpp:388-392), acceptance and Decompose (:410-419)
    const bool have = (n >= 3) && best_score < MAXD;
    double cnt[1] = {0.0};
    for (int i = tid; i < n; i += LO_T) {
        const bool in = have && sampson_err(best_model, pu + 3 * i, pv + 3 * i) < o.sq_thresh;
        inlier_mask[r0 + i] = in ? 1 : 0; cnt[0] += in ? 1.0 : 0.0;
    }
    block_sum<1>(cnt, S.red);
    if (tid == 0) {
        const int nin = (int)cnt[0]; num_inliers[pair] = nin;
        double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (have && nin > o.min_num_inliers) { double r[3]; decompose_E_dev(best_model, o.inward != 0, r); so3exp(r, Rm); }
        for (int k = 0; k < 9; k++) { outR[9 * (size_t)pair + k] = Rm[k]; outE[9 * (size_t)pair + k] = best_model[k]; }
        outScore[pair] = best_score;
        if (stats) { stats[2 * (size_t)pair] = it; stats[2 * (size_t)pair + 1] = lo_count; }
    }
    (void)best_num_inliers;
}

// ---- deterministic probes of the pieces (tests/test_ransac_probes_gpu.py) -------------------------------------------------------
// what: 0 = LeastSquares(list, E) -> E;  1 = Decompose(E) -> r (angle-axis) ;  2 = NonMinimalSolver(list) -> E (flag in out[9]);
// 3 = LeastSquares through the one-wave variant; one workgroup per task, rays of ONE pair in global memory.  EST_OUT doubles per task.
constexpr int EST_OUT = 20;
__global__ void __launch_bounds__(LO_T)
k_estimator_probe(int what, int n, const double* __restrict__ u, const double* __restrict__ v, const int* __restrict__ task_ptr,
                  const int* __restrict__ lists, const double* __restrict__ Ein /* [tasks*9] row-major */, int inward, double* __restrict__ out /* [tasks*EST_OUT] */) {
    __shared__ double red[28 * (LO_T / 64)]; __shared__ double sh[64];
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x, l0 = task_ptr[t], cnt = task_ptr[t + 1] - l0;
    int* list = reinterpret_cast<int*>(lds);
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) list[i] = lists[l0 + i];
    __syncthreads();
    double E[9]; for (int k = 0; k < 9; k++) E[k] = Ein[9 * (size_t)t + k];
    double* o = out + EST_OUT * (size_t)t;
    if (what == 0 || what == 3) {
        // LeastSquares: 0 = the workgroup-cooperative fit, 3 = the one-wave fit k_lomsac_trace runs; o[10..19] = [r1; t1], iterations, status, costs
        double tr[10];
        if (what == 0) block_sampson_lsq(list, cnt, u, v, inward != 0, E, red, sh, tr);
        else if (threadIdx.x < 64) wave_sampson_lsq(list, cnt, u, v, inward != 0, E, tr);
        if (threadIdx.x == 0) { for (int k = 0; k < 9; k++) o[k] = E[k]; o[9] = 1.0; for (int k = 0; k < 10; k++) o[10 + k] = tr[k]; }
    } else if (what == 1) {
        if (threadIdx.x == 0) { double r[3]; decompose_E_dev(E, inward != 0, r); o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; so3exp(r, o + 3); }
    } else {
        if (threadIdx.x == 0) { double Em[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; const int ok = nonminimal_solver_dev(list, min(cnt, 9), u, v, Em); for (int k = 0; k < 9; k++) o[k] = Em[k]; o[9] = (double)ok; }
    }
}

// SphericalEstimator::MinimalSolver as the reference returns it: always four candidates per sample, the real parts of complex
// eigenvectors / quartic roots included (src/spherical_solvers.cpp:296, 629-640); lane per sample
template <bool POLY>
__global__ void k_minimal_all_probe(int S, int ns /* rays per sample, 3..9 */, const int* __restrict__ sample, const double* __restrict__ u, const double* __restrict__ v,
                                    double* __restrict__ Es, int* __restrict__ counts) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double B[6][3], E[36];
    if (ns == 3) {
        double u3[9], v3[9];
        for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { u3[3 * i + k] = u[3 * sample[3 * s + i] + k]; v3[3 * i + k] = v[3 * sample[3 * s + i] + k]; }
        spherical_nullspace<3>(u3, v3, 3, B);
    } else {
        double uN[27], vN[27];
        for (int i = 0; i < 9; i++) for (int k = 0; k < 3; k++) { const int q = sample[ns * s + ((i < ns) ? i : 0)]; uN[3 * i + k] = u[3 * q + k]; vN[3 * i + k] = v[3 * q + k]; }
        spherical_nullspace<9>(uN, vN, ns, B);
    }
    const int c = spherical_models_from_basis<POLY, true>(B, E);
    counts[s] = c;
    for (int k = 0; k < 36; k++) Es[36 * (size_t)s + k] = (k < 9 * c) ? E[k] : 0.0;
}
// SphericalEstimator::EvaluateModelOnPoint (src/spherical_estimator.cpp:67-78) of T models on all n rays: err[t*n + i]
__global__ void k_sampson_probe(int T, int n, const double* __restrict__ Es, const double* __restrict__ u, const double* __restrict__ v, double* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (i >= n || t >= T) return;
    double E[9]; for (int k = 0; k < 9; k++) E[k] = Es[9 * (size_t)t + k];
    err[(size_t)t * n + i] = sampson_err(E, u + 3 * (size_t)i, v + 3 * (size_t)i);
}

// so3exp / so3ln / AngleAxisToRotationMatrix / RotationMatrixToAngleAxis on the device, one lane per item (row a9)
__global__ void k_so3_probe(int what, int n, const double* __restrict__ in, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (what == 0) so3exp(in + 3 * (size_t)i, out + 9 * (size_t)i);
    else if (what == 1) so3ln(in + 9 * (size_t)i, out + 3 * (size_t)i);
    else if (what == 2) angle_axis_to_matrix(in + 3 * (size_t)i, out + 9 * (size_t)i);
    else matrix_to_angle_axis(in + 9 * (size_t)i, out + 3 * (size_t)i);
}

// std::mt19937(seed) + uniform_int_distribution<int>(lo[i], hi[i]) draws, through the cooperative device generator
__global__ void __launch_bounds__(LO_T)
k_mt_probe(const unsigned* __restrict__ seeded, int n, const int* __restrict__ lo, const int* __restrict__ hi, int* __restrict__ out, unsigned* __restrict__ raw_out, int nraw) {
    __shared__ unsigned s[624];
    for (int i = threadIdx.x; i < 624; i += blockDim.x) s[i] = seeded[i];
    __syncthreads();
    int pos = 624;
    for (int i = 0; i < nraw; i++) { const unsigned r = mt_next(s, pos); if (threadIdx.x == 0) raw_out[i] = r; }
    for (int i = 0; i < n; i++) { const int d = mt_uniform_int(s, pos, lo[i], hi[i]); if (threadIdx.x == 0) out[i] = d; }
}

}  // namespace ssfm
using namespace ssfm;

static void rm_to_cm(const double* rm, double* cm) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cm[i + 3 * j] = rm[3 * i + j]; }
static void cm_to_rm(const double* cm, double* rm) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rm[3 * i + j] = cm[i + 3 * j]; }

// One slab of pairs through the trace kernel; device buffers are the caller's.  Used by ransac.hip's batch driver.
namespace ssfm {
int lomsac_launch(ssfm_ctx* ctx, hipStream_t st, int num_pairs, int max_n, const int* d_pair_ptr, const double* d_u, const double* d_v, int total,
                  const ssfm_ransac_options& O, double sq_thresh, const unsigned* d_mt_seeded, int* d_lists, double* d_E, double* d_score, double* d_R,
                  unsigned char* d_mask, int* d_nin, unsigned* d_stats) {
    LoOpts o;
    o.sq_thresh = sq_thresh; o.thresh_mult = O.threshold_multiplier; o.success_prob = O.success_probability;
    o.min_it = O.min_num_iterations; o.max_it = O.max_num_iterations; o.lo_start = O.lo_starting_iterations;
    o.num_lo_steps = O.num_lo_steps; o.num_lsq_it = O.num_lsq_iterations; o.min_sample_mult = O.min_sample_multiplicator;
    o.non_min_mult = O.non_min_sample_multiplier; o.final_lsq = O.final_least_squares; o.inward = O.inward; o.min_num_inliers = O.min_num_inliers;
    o.fast_shuffle = O.fast_shuffle;
    const size_t fixed = (size_t)(2 * 624 + LO_FIFO) * 4;
    const size_t lds_rays = (size_t)6 * max_n * 8 + (size_t)(2 * max_n + 2) * 4 + fixed;
    const bool in_lds = lds_rays <= 150 * 1024;
    const size_t lds = in_lds ? lds_rays : fixed;
    (void)total;
#define SSFM_LO_LAUNCH(P, L)                                                                                                                   \
    do {                                                                                                                                       \
        if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_lomsac_trace<P, L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_lomsac_trace<P, L>), dim3(num_pairs), dim3(LO_T), lds, st, d_pair_ptr, d_u, d_v, o, d_mt_seeded, d_lists, d_E, d_score, d_R, d_mask, d_nin, d_stats); \
    } while (0)
    if (O.use_poly_solver) { if (in_lds) SSFM_LO_LAUNCH(true, true); else SSFM_LO_LAUNCH(true, false); }
    else { if (in_lds) SSFM_LO_LAUNCH(false, true); else SSFM_LO_LAUNCH(false, false); }
#undef SSFM_LO_LAUNCH
    SSFM_HIP_CHECK(ctx, hipGetLastError());
    return SSFM_OK;
}
bool lomsac_needs_global_lists(int max_n) {
    return (size_t)6 * max_n * 8 + (size_t)(2 * max_n + 2) * 4 + (size_t)(2 * 624 + LO_FIFO) * 4 > 150 * 1024;
}
}  // namespace ssfm

// ---- C ABI: probes -----------------------------------------------------------------------------------------------------------
static int estimator_probe(ssfm_ctx* ctx, int what, int32_t n, const double* u, const double* v, int32_t tasks, const int32_t* task_ptr,
                           const int32_t* lists, const double* E_cm, int32_t inward, double* out12) {
    if (!ctx || n <= 0 || !u || !v || tasks <= 0 || !task_ptr || !out12) return fail(ctx, SSFM_ERR_INVALID, "ssfm estimator probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int nl = task_ptr[tasks]; int maxc = 1;
    for (int t = 0; t < tasks; t++) maxc = std::max(maxc, task_ptr[t + 1] - task_ptr[t]);
    for (int i = 0; i < nl; i++) if (lists[i] < 0 || lists[i] >= n) return fail(ctx, SSFM_ERR_INVALID, "ssfm estimator probe: index out of range");
    std::vector<double> hu(u, u + (size_t)3 * n), hv(v, v + (size_t)3 * n), hE((size_t)9 * tasks, 0.0);
    if (E_cm) for (int t = 0; t < tasks; t++) cm_to_rm(E_cm + 9 * (size_t)t, &hE[9 * (size_t)t]);
    std::vector<int> hp(task_ptr, task_ptr + tasks + 1), hl(lists, lists + std::max(nl, 0)); if (hl.empty()) hl.push_back(0);
    DevBuf<double> du, dv, dE, dout; DevBuf<int> dp, dl;
    int rc = SSFM_OK;
    auto body = [&]() -> int {
        SSFM_HIP_CHECK(ctx, upload(du, hu, st)); SSFM_HIP_CHECK(ctx, upload(dv, hv, st)); SSFM_HIP_CHECK(ctx, upload(dE, hE, st));
        SSFM_HIP_CHECK(ctx, upload(dp, hp, st)); SSFM_HIP_CHECK(ctx, upload(dl, hl, st)); SSFM_HIP_CHECK(ctx, dout.alloc((size_t)EST_OUT * tasks));
        const size_t lds = (size_t)maxc * 4 + 16;
        if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_estimator_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_estimator_probe, dim3(tasks), dim3(LO_T), lds, st, what, n, du.p, dv.p, dp.p, dl.p, dE.p, inward, dout.p);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(out12, dout.p, (size_t)EST_OUT * tasks * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    rc = body();
    du.free(); dv.free(); dE.free(); dout.free(); dp.free(); dl.free();
    return rc;
}

extern "C" int ssfm_sampson_refine_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t tasks, const int32_t* task_ptr,
                                         const int32_t* lists, int32_t inward, double* E_inout) {
    std::vector<double> out((size_t)EST_OUT * std::max(tasks, 1));
    const int rc = estimator_probe(ctx, 0, n, u, v, tasks, task_ptr, lists, E_inout, inward, out.data());
    if (rc) return rc;
    for (int t = 0; t < tasks; t++) rm_to_cm(&out[EST_OUT * (size_t)t], E_inout + 9 * (size_t)t);
    return SSFM_OK;
}
// the same fit with its trace: variant 0 = workgroup-cooperative, 1 = the one-wave form k_lomsac_trace runs.  trace: [tasks*10] =
// [r1; t1] at the end (t1 is free in the reference's problem, src/spherical_estimator.cpp:140-144), LM iterations, status
// (0 converged, 1 iteration limit, 2 invalid steps, 3 evaluation failure), initial cost, final cost
extern "C" int ssfm_sampson_refine_probe_ex(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t tasks, const int32_t* task_ptr,
                                            const int32_t* lists, int32_t inward, int32_t variant, double* E_inout, double* trace) {
    if (variant != 0 && variant != 1) return fail(ctx, SSFM_ERR_INVALID, "ssfm_sampson_refine_probe_ex: variant is 0 or 1");
    std::vector<double> out((size_t)EST_OUT * std::max(tasks, 1));
    const int rc = estimator_probe(ctx, variant ? 3 : 0, n, u, v, tasks, task_ptr, lists, E_inout, inward, out.data());
    if (rc) return rc;
    for (int t = 0; t < tasks; t++) {
        rm_to_cm(&out[EST_OUT * (size_t)t], E_inout + 9 * (size_t)t);
        if (trace) for (int k = 0; k < 10; k++) trace[10 * (size_t)t + k] = out[EST_OUT * (size_t)t + 10 + k];
    }
    return SSFM_OK;
}
extern "C" int ssfm_decompose_probe(ssfm_ctx* ctx, int32_t tasks, const double* E, int32_t inward, double* r_out, double* R_out) {
    if (!E || tasks <= 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_decompose_probe: bad arguments");
    std::vector<double> out((size_t)EST_OUT * tasks); std::vector<int> ptr(tasks + 1, 0), lists(1, 0);
    const double dummy[3] = {0, 0, 1};
    const int rc = estimator_probe(ctx, 1, 1, dummy, dummy, tasks, ptr.data(), lists.data(), E, inward, out.data());
    if (rc) return rc;
    for (int t = 0; t < tasks; t++) {
        const double* o = &out[EST_OUT * (size_t)t];
        if (r_out) { r_out[3 * t] = o[0]; r_out[3 * t + 1] = o[1]; r_out[3 * t + 2] = o[2]; }
        if (R_out) rm_to_cm(o + 3, R_out + 9 * (size_t)t);
    }
    return SSFM_OK;
}
extern "C" int ssfm_nonminimal_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t tasks, const int32_t* task_ptr,
                                     const int32_t* lists, double* E_out, int32_t* ok_out) {
    std::vector<double> out((size_t)EST_OUT * std::max(tasks, 1));
    if (task_ptr) for (int t = 0; t < tasks; t++) { const int c = task_ptr[t + 1] - task_ptr[t]; if (c < 3 || c > 9) return fail(ctx, SSFM_ERR_INVALID, "ssfm_nonminimal_probe: samples hold 3..9 rays"); }
    const int rc = estimator_probe(ctx, 2, n, u, v, tasks, task_ptr, lists, nullptr, 0, out.data());
    if (rc) return rc;
    for (int t = 0; t < tasks; t++) { if (E_out) rm_to_cm(&out[EST_OUT * (size_t)t], E_out + 9 * (size_t)t); if (ok_out) ok_out[t] = (int)out[EST_OUT * (size_t)t + 9]; }
    return SSFM_OK;
}

extern "C" int ssfm_so3_probe(ssfm_ctx* ctx, int32_t what, int32_t n, const double* in, double* out) {
    if (!ctx || n <= 0 || !in || !out || what < 0 || what > 3) return fail(ctx, SSFM_ERR_INVALID, "ssfm_so3_probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool vec_in = (what == 0 || what == 2);
    const size_t nin = (size_t)(vec_in ? 3 : 9) * n, nout = (size_t)(vec_in ? 9 : 3) * n;
    std::vector<double> hin(nin), hout(nout);
    // matrices cross the ABI column-major; the kernels are row-major
    if (vec_in) std::memcpy(hin.data(), in, nin * sizeof(double)); else for (int i = 0; i < n; i++) cm_to_rm(in + 9 * (size_t)i, &hin[9 * (size_t)i]);
    DevBuf<double> din, dout;
    auto body = [&]() -> int {
        SSFM_HIP_CHECK(ctx, upload(din, hin, st)); SSFM_HIP_CHECK(ctx, dout.alloc(nout));
        hipLaunchKernelGGL(k_so3_probe, dim3((n + 63) / 64), dim3(64), 0, st, what, n, din.p, dout.p);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hout.data(), dout.p, nout * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    const int rc = body();
    din.free(); dout.free();
    if (rc) return rc;
    if (vec_in) for (int i = 0; i < n; i++) rm_to_cm(&hout[9 * (size_t)i], out + 9 * (size_t)i); else std::memcpy(out, hout.data(), nout * sizeof(double));
    return SSFM_OK;
}

extern "C" int ssfm_mt19937_probe(ssfm_ctx* ctx, uint32_t seed, int32_t n, const int32_t* lo, const int32_t* hi, int32_t* draws, int32_t nraw, uint32_t* raw) {
    if (!ctx || n < 0 || nraw < 0 || (n > 0 && (!lo || !hi || !draws)) || (nraw > 0 && !raw)) return fail(ctx, SSFM_ERR_INVALID, "ssfm_mt19937_probe: bad arguments");
    for (int i = 0; i < n; i++) if (hi[i] < lo[i]) return fail(ctx, SSFM_ERR_INVALID, "ssfm_mt19937_probe: empty range");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<unsigned> seeded(624); mt_seed_host(seed, seeded.data());
    std::vector<int> hlo(lo, lo + n), hhi(hi, hi + n); if (hlo.empty()) { hlo.push_back(0); hhi.push_back(0); }
    DevBuf<unsigned> ds, draw; DevBuf<int> dlo, dhi, dout;
    auto body = [&]() -> int {
        SSFM_HIP_CHECK(ctx, upload(ds, seeded, st)); SSFM_HIP_CHECK(ctx, upload(dlo, hlo, st)); SSFM_HIP_CHECK(ctx, upload(dhi, hhi, st));
        SSFM_HIP_CHECK(ctx, dout.alloc(std::max(n, 1))); SSFM_HIP_CHECK(ctx, draw.alloc(std::max(nraw, 1)));
        hipLaunchKernelGGL(k_mt_probe, dim3(1), dim3(LO_T), 0, st, ds.p, n, dlo.p, dhi.p, dout.p, draw.p, nraw);
        if (n > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(draws, dout.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
        if (nraw > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(raw, draw.p, (size_t)nraw * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    const int rc = body();
    ds.free(); draw.free(); dlo.free(); dhi.free(); dout.free();
    return rc;
}

extern "C" int ssfm_minimal_solver_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples, int32_t use_poly_solver,
                                         double* Es, int32_t* counts) {
    if (!ctx || n <= 0 || !u || !v || S <= 0 || !samples || !Es || !counts) return fail(ctx, SSFM_ERR_INVALID, "ssfm_minimal_solver_probe: bad arguments");
    for (int i = 0; i < 3 * S; i++) if (samples[i] < 0 || samples[i] >= n) return fail(ctx, SSFM_ERR_INVALID, "ssfm_minimal_solver_probe: index out of range");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<double> hu(u, u + (size_t)3 * n), hv(v, v + (size_t)3 * n), hE((size_t)36 * S); std::vector<int> hs(samples, samples + (size_t)3 * S);
    DevBuf<double> du, dv, dE; DevBuf<int> ds, dc;
    auto body = [&]() -> int {
        SSFM_HIP_CHECK(ctx, upload(du, hu, st)); SSFM_HIP_CHECK(ctx, upload(dv, hv, st)); SSFM_HIP_CHECK(ctx, upload(ds, hs, st));
        SSFM_HIP_CHECK(ctx, dE.alloc((size_t)36 * S)); SSFM_HIP_CHECK(ctx, dc.alloc(S));
        if (use_poly_solver) hipLaunchKernelGGL(k_minimal_all_probe<true>, dim3((S + 63) / 64), dim3(64), 0, st, S, 3, ds.p, du.p, dv.p, dE.p, dc.p);
        else hipLaunchKernelGGL(k_minimal_all_probe<false>, dim3((S + 63) / 64), dim3(64), 0, st, S, 3, ds.p, du.p, dv.p, dE.p, dc.p);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hE.data(), dE.p, hE.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(counts, dc.p, S * sizeof(int), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    const int rc = body();
    du.free(); dv.free(); dE.free(); ds.free(); dc.free();
    if (rc) return rc;
    for (int s = 0; s < S; s++) for (int m = 0; m < 4; m++) rm_to_cm(&hE[36 * (size_t)s + 9 * m], Es + 36 * (size_t)s + 9 * m);
    return SSFM_OK;
}

extern "C" int ssfm_sampson_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t T, const double* Es, double* errors) {
    if (!ctx || n <= 0 || !u || !v || T <= 0 || !Es || !errors) return fail(ctx, SSFM_ERR_INVALID, "ssfm_sampson_probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<double> hu(u, u + (size_t)3 * n), hv(v, v + (size_t)3 * n), hE((size_t)9 * T);
    for (int t = 0; t < T; t++) cm_to_rm(Es + 9 * (size_t)t, &hE[9 * (size_t)t]);
    DevBuf<double> du, dv, dE, derr;
    auto body = [&]() -> int {
        SSFM_HIP_CHECK(ctx, upload(du, hu, st)); SSFM_HIP_CHECK(ctx, upload(dv, hv, st)); SSFM_HIP_CHECK(ctx, upload(dE, hE, st)); SSFM_HIP_CHECK(ctx, derr.alloc((size_t)T * n));
        hipLaunchKernelGGL(k_sampson_probe, dim3((n + 255) / 256, T), dim3(256), 0, st, T, n, dE.p, du.p, dv.p, derr.p);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(errors, derr.p, (size_t)T * n * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    const int rc = body();
    du.free(); dv.free(); dE.free(); derr.free();
    return rc;
}

// ---- C ABI: the reference's estimator interface for ONE pair, rays resident on the device ---------------------------------------
// One entry point per virtual of sphericalsfm::Estimator<Matrix3d> / EssentialEstimator (include/sphericalsfm/estimator.h:7-29) as
// SphericalEstimator implements them (src/spherical_estimator.cpp:67-164).  This is the single-pair boundary a host-side RANSAC driver
// (ransac_lib::LocallyOptimizedMSAC, include/RansacLib/ransac.h:128) calls into; thousands of pairs go through ssfm_ransac_batch instead.
struct ssfm_estimator {
    ssfm_ctx* ctx = nullptr; int n = 0, poly = 0, inward = 0;
    DevBuf<double> u, v, E, out; DevBuf<int> idx, cnt;
    double* h_d = nullptr; int* h_i = nullptr;          // pinned staging: [max(n, 48) doubles], [max(n, 9) + 4 ints] (a sample may repeat indices: up to 9 entries on a pair with n < 9)
};
extern "C" int ssfm_estimator_create(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t use_poly_solver, int32_t inward, ssfm_estimator** out) {
    if (!ctx || n < 0 || (n > 0 && (!u || !v)) || !out) return fail(ctx, SSFM_ERR_INVALID, "ssfm_estimator_create: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ssfm_estimator* e = new ssfm_estimator; e->ctx = ctx; e->n = n; e->poly = use_poly_solver; e->inward = inward;
    auto body = [&]() -> int {
        std::vector<double> hu(u, u + (size_t)3 * n), hv(v, v + (size_t)3 * n); if (hu.empty()) { hu.assign(3, 0.0); hv.assign(3, 0.0); }
        SSFM_HIP_CHECK(ctx, upload(e->u, hu, ctx->stream)); SSFM_HIP_CHECK(ctx, upload(e->v, hv, ctx->stream));
        SSFM_HIP_CHECK(ctx, e->E.alloc(36)); SSFM_HIP_CHECK(ctx, e->out.alloc(std::max(n, 48))); SSFM_HIP_CHECK(ctx, e->idx.alloc(std::max(n, 9) + 4)); SSFM_HIP_CHECK(ctx, e->cnt.alloc(4));
        SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&e->h_d, (size_t)std::max(n, 48) * sizeof(double), hipHostMallocDefault));
        SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&e->h_i, (size_t)(std::max(n, 9) + 4) * sizeof(int), hipHostMallocDefault));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return SSFM_OK;
    };
    const int rc = body();
    if (rc) { ssfm_estimator_destroy(e); return rc; }
    *out = e;
    return SSFM_OK;
}
extern "C" void ssfm_estimator_destroy(ssfm_estimator* e) {
    if (!e) return;
    (void)hipSetDevice(e->ctx->device); (void)hipStreamSynchronize(e->ctx->stream);
    e->u.free(); e->v.free(); e->E.free(); e->out.free(); e->idx.free(); e->cnt.free();
    if (e->h_d) (void)hipHostFree(e->h_d); if (e->h_i) (void)hipHostFree(e->h_i);
    delete e;
}
static int est_check_sample(ssfm_estimator* e, const int32_t* sample, int32_t m, int lo, int hi, const char* who) {
    if (!e || !sample || m < lo || m > hi) return fail(e ? e->ctx : nullptr, SSFM_ERR_INVALID, std::string(who) + ": bad sample");
    for (int i = 0; i < m; i++) if (sample[i] < 0 || sample[i] >= e->n) return fail(e->ctx, SSFM_ERR_INVALID, std::string(who) + ": index out of range");
    return SSFM_OK;
}
// uploads [0, m | sample] as the one-task CSR the probe kernel reads
static int est_upload_list(ssfm_estimator* e, const int32_t* sample, int32_t m) {
    e->h_i[0] = 0; e->h_i[1] = m; for (int i = 0; i < m; i++) e->h_i[2 + i] = sample[i];
    SSFM_HIP_CHECK(e->ctx, hipMemcpyAsync(e->cnt.p, e->h_i, 2 * sizeof(int), hipMemcpyHostToDevice, e->ctx->stream));
    if (m) SSFM_HIP_CHECK(e->ctx, hipMemcpyAsync(e->idx.p, e->h_i + 2, (size_t)m * sizeof(int), hipMemcpyHostToDevice, e->ctx->stream));
    return SSFM_OK;
}
extern "C" int ssfm_estimator_minimal_solver(ssfm_estimator* e, const int32_t* sample, int32_t sample_size, double* Es, int32_t* num_models) {
    { const int rc = est_check_sample(e, sample, sample_size, 3, 9, "ssfm_estimator_minimal_solver"); if (rc) return rc; }
    ssfm_ctx* ctx = e->ctx; hipStream_t st = ctx->stream;
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < sample_size; i++) e->h_i[i] = sample[i];
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(e->idx.p, e->h_i, (size_t)sample_size * sizeof(int), hipMemcpyHostToDevice, st));
    if (e->poly) hipLaunchKernelGGL(k_minimal_all_probe<true>, dim3(1), dim3(64), 0, st, 1, sample_size, e->idx.p, e->u.p, e->v.p, e->E.p, e->cnt.p);
    else hipLaunchKernelGGL(k_minimal_all_probe<false>, dim3(1), dim3(64), 0, st, 1, sample_size, e->idx.p, e->u.p, e->v.p, e->E.p, e->cnt.p);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(e->h_d, e->E.p, 36 * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(e->h_i + sample_size, e->cnt.p, sizeof(int), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    const int c = e->h_i[sample_size];
    for (int m = 0; m < c; m++) rm_to_cm(e->h_d + 9 * m, Es + 9 * m);
    if (num_models) *num_models = c;
    return SSFM_OK;
}
static int est_task(ssfm_estimator* e, int what, const int32_t* sample, int32_t m, const double* E_cm, double* out12) {
    ssfm_ctx* ctx = e->ctx; hipStream_t st = ctx->stream;
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    { const int rc = est_upload_list(e, sample, m); if (rc) return rc; }
    if (E_cm) { cm_to_rm(E_cm, e->h_d); SSFM_HIP_CHECK(ctx, hipMemcpyAsync(e->E.p, e->h_d, 9 * sizeof(double), hipMemcpyHostToDevice, st)); }
    const size_t lds = (size_t)std::max(m, 1) * 4 + 16;
    if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_estimator_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_estimator_probe, dim3(1), dim3(LO_T), lds, st, what, e->n, e->u.p, e->v.p, e->cnt.p, e->idx.p, e->E.p, e->inward, e->out.p);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(e->h_d + 16, e->out.p, EST_OUT * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (int k = 0; k < EST_OUT; k++) out12[k] = e->h_d[16 + k];
    return SSFM_OK;
}
extern "C" int ssfm_estimator_non_minimal_solver(ssfm_estimator* e, const int32_t* sample, int32_t sample_size, double* E, int32_t* ok) {
    { const int rc = est_check_sample(e, sample, sample_size, 3, 9, "ssfm_estimator_non_minimal_solver"); if (rc) return rc; }
    double o[EST_OUT]; const int rc = est_task(e, 2, sample, sample_size, nullptr, o); if (rc) return rc;
    if (ok) *ok = (int)o[9];
    if (o[9] != 0.0) rm_to_cm(o, E);
    return SSFM_OK;
}
extern "C" int ssfm_estimator_least_squares(ssfm_estimator* e, const int32_t* sample, int32_t sample_size, double* E) {
    { const int rc = est_check_sample(e, sample, sample_size, 0, e ? e->n : 0, "ssfm_estimator_least_squares"); if (rc) return rc; }
    if (!E) return fail(e->ctx, SSFM_ERR_INVALID, "ssfm_estimator_least_squares: E is null");
    double o[EST_OUT]; const int rc = est_task(e, 0, sample, sample_size, E, o); if (rc) return rc;
    rm_to_cm(o, E);
    return SSFM_OK;
}
extern "C" int ssfm_estimator_decompose(ssfm_estimator* e, const double* E, double* R, double* t) {
    if (!e || !E) return fail(e ? e->ctx : nullptr, SSFM_ERR_INVALID, "ssfm_estimator_decompose: bad arguments");
    const int32_t none = 0; double o[EST_OUT]; const int rc = est_task(e, 1, &none, 0, E, o); if (rc) return rc;
    if (R) rm_to_cm(o + 3, R);
    if (t) { const double s = e->inward ? -1.0 : 1.0; t[0] = s * o[3 + 2]; t[1] = s * o[3 + 5]; t[2] = s * (o[3 + 8] - 1.0); }   // src/spherical_utils.cpp:43-49: t = R.col(2) - e_z, negated if inward
    return SSFM_OK;
}
extern "C" int ssfm_estimator_evaluate_model(ssfm_estimator* e, const double* E, double* errors) {
    if (!e || !E || !errors) return fail(e ? e->ctx : nullptr, SSFM_ERR_INVALID, "ssfm_estimator_evaluate_model: bad arguments");
    if (e->n == 0) return SSFM_OK;
    ssfm_ctx* ctx = e->ctx; hipStream_t st = ctx->stream;
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    cm_to_rm(E, e->h_d);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(e->E.p, e->h_d, 9 * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_sampson_probe, dim3((e->n + 255) / 256, 1), dim3(256), 0, st, 1, e->n, e->E.p, e->u.p, e->v.p, e->out.p);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(e->h_d, e->out.p, (size_t)e->n * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    std::memcpy(errors, e->h_d, (size_t)e->n * sizeof(double));
    return SSFM_OK;
}
