// spherical_sfm_amd -- block-banded Cholesky preconditioner for the reduced camera system.
//
// The reference solves the reduced system with a direct sparse Cholesky (Ceres SPARSE_SCHUR,
// src/sfm.cpp:273).  A camera ring has long-wavelength modes that block-Jacobi PCG needs ~10^3
// iterations for (measured: profiles/r01_notes.md), so PCG is preconditioned with an exact Cholesky of S
// restricted to a block band in Cuthill-McKee order.  When the band covers the whole structure (always the
// case for the ordering built in ba_flatten.h) the preconditioner is S^-1 itself and PCG is a one-step
// refinement that also checks the residual.
//
// Layout: band[(i*(b+1) + d)*DC*DC + ...] = block (i, i-d) of the permuted matrix, d = 0..b, row-major DCxDC.
// One 1024-lane workgroup runs the right-looking factorisation; the DCxDC panel of the current block column
// sits in LDS, the trailing window is updated in place (L2-resident).  Right-hand sides ride along as border
// rows, so the forward substitution costs no extra pass.
#pragma once
#include <hip/hip_runtime.h>
#include "ba_kernels.h"

namespace ssfm {

// 1/sqrt(d) from the hardware estimate (v_rsq_f64) plus two Newton steps: full double precision with a
// dependent chain of ~10 instructions instead of the ~50 of an IEEE sqrt followed by an IEEE divide.
// Barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, which would expose the
// latency of the global prefetches / write-backs that the LDS-resident kernels deliberately leave in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Band row(s) of camera c from its row of S.  The workgroup owns band row i = pos[c] (its blocks all come from S row c): clear it,
// then scatter.  A separator camera of a twisted component owns a second row i2 behind the reversed segment: blocks whose column
// lies there go to that row.  S may hold both triangles (pose graphs): only blocks inside the band of a row are taken.
// MERGE (DC = 3): two consecutive camera rows share one 6x6 block row R = i >> 1 (upper / lower half u = i & 1); the camera owns its
// three scalar rows of every block of R, except that the off-diagonal 3x3 parts of the DIAGONAL block both belong to the lower
// camera (it mirrors its (1,0) part into (0,1)); an even camera without partner (pair_dummy) also owns the partner's rows: identity.
// first half: the camera's band row(s) cleared (no barrier inside)
template <int DC, bool MERGE>
__device__ __forceinline__ void band_rows_zero(int i, int i2, int b, bool dummy, double* __restrict__ band) {
    constexpr int BB = DC * DC;
    const int tid = threadIdx.x, W = b + 1;
    if (!MERGE) {
        double* row = band + (size_t)i * W * BB;
        double* row2 = band + (size_t)max(i2, 0) * W * BB;
        for (int e = tid; e < W * BB; e += blockDim.x) { row[e] = 0.0; if (i2 >= 0) row2[e] = 0.0; }
    } else {
        constexpr int BM = 4 * BB, D2 = 2 * DC;                    // 6x6 blocks of 3x3 parts
        const int R = i >> 1, u = i & 1, R2 = max(i2, 0) >> 1, u2 = max(i2, 0) & 1;
        double* row = band + (size_t)R * W * BM;
        double* row2 = band + (size_t)R2 * W * BM;
        for (int e = tid; e < W * DC * D2; e += blockDim.x) {      // this camera's DC rows of every block
            const int d = e / (DC * D2), q = e - d * (DC * D2), r = q / D2, col = q - r * D2;
            if (!(d == 0 && u == 0 && col >= DC && !dummy)) row[(size_t)d * BM + (u * DC + r) * D2 + col] = 0.0;
            if (i2 >= 0 && !(d == 0 && u2 == 0 && col >= DC)) row2[(size_t)d * BM + (u2 * DC + r) * D2 + col] = 0.0;
            if (dummy) row[(size_t)d * BM + (DC + r) * D2 + col] = (d == 0 && col == DC + r) ? 1.0 : 0.0;       // the empty partner slot: identity
        }
        if (tid < BB) {                                            // part (0,1) of the diagonal block belongs to the lower camera
            const int r = tid / DC, col = tid - r * DC;
            if (u == 1) row[r * D2 + DC + col] = 0.0;
            if (i2 >= 0 && u2 == 1) row2[r * D2 + DC + col] = 0.0;
        }
    }
}
// second half (behind a barrier): the blocks of S's row c scattered into the band row(s).  col_pos (optional) = pos[col_idx[.]] precomputed: one dependent gather less
template <int DC, bool MERGE>
__device__ __forceinline__ void band_rows_scatter(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const double* __restrict__ S_val,
                                                  const int* __restrict__ pos, int i, int i2, int c, int b, double* __restrict__ band, const int* __restrict__ col_pos = nullptr) {
    constexpr int BB = DC * DC;
    const int rb = row_ptr[c], nnb = row_ptr[c + 1] - rb, tid = threadIdx.x, W = b + 1;
    if (!MERGE) {
        double* row = band + (size_t)i * W * BB;
        double* row2 = band + (size_t)max(i2, 0) * W * BB;
        for (int idx = tid; idx < nnb * BB; idx += blockDim.x) {
            const int s = rb + idx / BB, e = idx % BB;
            const int k = col_pos ? col_pos[s] : pos[col_idx[s]];
            if (k <= i) { if (i - k <= b) row[(size_t)(i - k) * BB + e] = S_val[(size_t)s * BB + e]; }
            else if (i2 > k && i2 - k <= b) row2[(size_t)(i2 - k) * BB + e] = S_val[(size_t)s * BB + e];
        }
    } else {
        constexpr int BM = 4 * BB, D2 = 2 * DC;
        const int R = i >> 1, u = i & 1, R2 = max(i2, 0) >> 1, u2 = max(i2, 0) & 1;
        double* row = band + (size_t)R * W * BM;
        double* row2 = band + (size_t)R2 * W * BM;
        for (int idx = tid; idx < nnb * BB; idx += blockDim.x) {
            const int s = rb + idx / BB, e = idx % BB, a = e / DC, a2 = e - a * DC;
            const int k = col_pos ? col_pos[s] : pos[col_idx[s]], C = k >> 1, v = k & 1;
            const double val = S_val[(size_t)s * BB + e];
            if (C < R) { if (R - C <= b) row[(size_t)(R - C) * BM + (u * DC + a) * D2 + v * DC + a2] = val; }
            else if (C == R) { if (v <= u) { row[(u * DC + a) * D2 + v * DC + a2] = val; if (v < u) row[(v * DC + a2) * D2 + u * DC + a] = val; } }
            else if (i2 >= 0 && R2 > C && R2 - C <= b) row2[(size_t)(R2 - C) * BM + (u2 * DC + a) * D2 + v * DC + a2] = val;
        }
    }
}
// Round 5, rings (ba_flatten.h: band_plan, ring_wrap_table): the coupling blocks between the first arc of a ring and its LAST separator are stored in S in the row of
// the separator camera (it is eliminated later), but the band wants them in the row of the arc camera c, relative to the separator's copy slot in front of the arc:
// (band row of c) - (copy row of the row camera) = d <= b, the block TRANSPOSED.  Camera c's own workgroup places them (its row was zeroed by itself).
template <int DC, bool MERGE>
__device__ __forceinline__ void band_rows_wrap(const int* __restrict__ wrap_ptr, const int* __restrict__ wrap_blk, const int* __restrict__ wrap_row2,
                                               const double* __restrict__ S_val, int i, int c, int b, double* __restrict__ band) {
    constexpr int BB = DC * DC;
    const int w0 = wrap_ptr[c], nw = wrap_ptr[c + 1] - w0, tid = threadIdx.x, W = b + 1;
    for (int idx = tid; idx < nw * BB; idx += blockDim.x) {
        const int q = w0 + idx / BB, e = idx % BB, a = e / DC, a2 = e - a * DC;     // S block (row camera components a, this camera's components a2)
        const double val = S_val[(size_t)wrap_blk[q] * BB + e];
        const int k = wrap_row2[q];
        if (!MERGE) band[((size_t)i * W + (i - k)) * BB + a2 * DC + a] = val;
        else { constexpr int BM = 4 * BB, D2 = 2 * DC; const int R = i >> 1, u = i & 1, C = k >> 1, v = k & 1;
               band[((size_t)R * W + (R - C)) * BM + (u * DC + a2) * D2 + v * DC + a] = val; }
    }
}
template <int DC, bool MERGE>
__device__ __forceinline__ void band_rows_fill(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const double* __restrict__ S_val,
                                               const int* __restrict__ pos, int i, int i2, int c, int b, bool dummy, double* __restrict__ band) {
    band_rows_zero<DC, MERGE>(i, i2, b, dummy, band);
    __syncthreads();
    band_rows_scatter<DC, MERGE>(row_ptr, col_idx, S_val, pos, i, i2, c, b, band);
}

// S (block-CSR, camera order) -> band storage (permuted)
template <int DC, bool MERGE>
__global__ void k_band_gather(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const double* __restrict__ S_val,
                              const int* __restrict__ pos, const int* __restrict__ pos2, const unsigned char* __restrict__ pair_dummy, int Nc, int b,
                              double* __restrict__ band, const int* __restrict__ wrap_ptr = nullptr, const int* __restrict__ wrap_blk = nullptr,
                              const int* __restrict__ wrap_row2 = nullptr) {
    const int c = blockIdx.x;
    band_rows_fill<DC, MERGE>(row_ptr, col_idx, S_val, pos, pos[c], pos2 ? pos2[c] : -1, c, b, pair_dummy && pair_dummy[c], band);
    if (wrap_ptr) band_rows_wrap<DC, MERGE>(wrap_ptr, wrap_blk, wrap_row2, S_val, pos[c], c, b, band);
}
// k_finalize_S + k_band_gather + k_band_permute_rhs in one launch for the BA path with the banded preconditioner (one workgroup per
// camera, which owns its diagonal block, its band row and its slice of the right-hand sides): LM diagonal on the diagonal block
// (in S_val too: the residual check multiplies by S), gradient max-norm, band row, permuted [rhs | S_fc]; workgroup 0 also folds the
// focal sums and writes the focal row.  The block-Jacobi inverse of k_finalize_S is only needed by preconditioner 1 and is not built here.
// (the default kernel keeps __restrict__ on the arrays the fold variant writes through *_w)
template <bool R, class T> struct RestrictIf { typedef T* type; };
template <class T> struct RestrictIf<true, T> { typedef T* __restrict__ type; };
// FOLD (round 6): the variant that first folds the partial blocks of the atomics-free Gram emission (its 37 KB of LDS and its registers stay out of the default kernel)
template <int DC, bool MERGE, bool FOLD = false>
__global__ void __launch_bounds__(256)
k_finalize_gather(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const int* __restrict__ diag_slot,
                  const double* __restrict__ scale_cam, const double* __restrict__ scale_f, typename RestrictIf<!FOLD, const double>::type Udiag,      // (Udiag, gcraw, Sfc, rhs, S_val: no __restrict__ in the variant --
                  typename RestrictIf<!FOLD, const double>::type gcraw, double radius, double min_diag, double max_diag, int Nc, const int* __restrict__ pos,            //  the fold below writes them through *_w)
                  const int* __restrict__ pos2, const unsigned char* __restrict__ pair_dummy, int Nb, int b, typename RestrictIf<!FOLD, double>::type S_val, typename RestrictIf<!FOLD, double>::type rhs,
                  typename RestrictIf<!FOLD, const double>::type Sfc, double* __restrict__ Sff,
                  double* __restrict__ band, double* __restrict__ Y, double* __restrict__ scal,
                  double2* __restrict__ clear = nullptr, size_t clear_len2 = 0, const int* __restrict__ col_pos = nullptr,
                  const int* __restrict__ wrap_ptr = nullptr, const int* __restrict__ wrap_blk = nullptr, const int* __restrict__ wrap_row2 = nullptr,
                  // round 6, atomics-free Gram emission (ba_flatten.h: fold lists): this camera's row of S and its vectors are the sums, in list order, of the
                  // partial blocks / vector stretches k_schur_gram's tasks stored; nullptr: S_val ... hold the accumulated values already
                  const double* __restrict__ part = nullptr, const int* __restrict__ fold_slot_ptr = nullptr, const int* __restrict__ fold_slot_src = nullptr,
                  double* __restrict__ Udiag_w = nullptr,
                  double* __restrict__ gcraw_w = nullptr, double* __restrict__ Sfc_w = nullptr,
                  // deterministic mode without a decode launch: the four focal sums come straight from the long accumulators (det_acc.h), which are cleared here
                  long long* __restrict__ lacc = nullptr) {
    constexpr int BB = DC * DC; constexpr int off = (DC == 6) ? 0 : 3;
    const int c = blockIdx.x, tid = threadIdx.x;
    // deterministic mode: the limbs of the four focal sums (wave 1 of workgroup 0 folds them at the end) are requested NOW -- 32 independent loads that fly under
    // everything else; loaded where they are used they were 5 us on the launch's critical path
    constexpr int LA_KS[4] = {SC_FJJ, SC_FWW, SC_FJR, SC_FWG};
    long long lv[4][LA_STRIDE];
    // (with a fixed focal length the sums are not used -- S_ff = 1, rho = 0 -- and nothing was added to their limbs: Jf is scaled by zero)
    const bool la_wave = FOLD && lacc && c == 0 && tid >= 64 && tid < 128 && scale_f[0] > 0.0;
    if (la_wave) {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int j = 0; j < LA_STRIDE; j++) lv[k][j] = lacc[((size_t)(tid - 64) * SC_TOTAL + LA_KS[k]) * LA_STRIDE + j];
    }
    if constexpr (FOLD) {
        // Everything this row needs -- the partial blocks of its slots, then the vector stretches of its camera -- is one list of sources in a fixed-stride table per
        // camera (ba_flatten.h: GRAM_FOLD_*): ONE round of index loads, ONE round of independent value loads into LDS, then every entry of S / of the four vectors is
        // summed over its own sub-range in list order from LDS.  (A thread walking its entry's list through memory: 19 us per launch, a dependent round trip per source;
        // CSR lists found through row_ptr: 23 us, four dependent round trips -- these small launches pay ~2.5 us for each.)
        constexpr int SL = (BB > 5 * DC) ? BB : 5 * DC;                                            // doubles per source in LDS: a block, or a vector stretch (diag U | rhs 1 | rhs 2 | Jc^T r | S_fc)
        __shared__ double fold_buf[GRAM_FOLD_SRCS * SL];
        __shared__ int fold_tab[GRAM_FOLD_STRIDE];
        const int* __restrict__ tab = fold_slot_src + (size_t)c * GRAM_FOLD_STRIDE;
        if (tid < GRAM_FOLD_STRIDE) fold_tab[tid] = tab[tid];
        const int nnb0 = row_ptr[c + 1] - row_ptr[c], rb0 = row_ptr[c];
        __syncthreads();
        const int nq = fold_tab[0];
        // (all of a thread's loads in flight before the first LDS store: the plain loop waited for every load in turn -- eleven dependent round trips, 15 us)
        constexpr int FOLD_LD = (GRAM_FOLD_SRCS * SL + 255) / 256;                                 // <= 18 values per thread
        {
            double fv[FOLD_LD];
#pragma unroll
            for (int u = 0; u < FOLD_LD; u++) { const int idx = min(tid + u * 256, max(nq * SL - 1, 0)); fv[u] = part[(size_t)fold_tab[GRAM_FOLD_HEAD + idx / SL] + idx % SL]; }      // (reads past a 3-dof block's 9 entries stay inside the buffer's slack)
#pragma unroll
            for (int u = 0; u < FOLD_LD; u++) { const int idx = tid + u * 256; if (idx < nq * SL) fold_buf[idx] = fv[u]; }
        }
        __syncthreads();
        const int nent = nnb0 * BB + 4 * DC;                                                       // entries this workgroup produces: blocks, then diag U | rhs | Jc^T r | S_fc
#pragma unroll
        for (int u = 0; u < 4; u++) {                                                              // <= 4 x 256 entries: rows of <= 27 blocks (the host checks)
            const int idx = tid + u * 256;
            if (idx < nent) {
                const bool vec = idx >= nnb0 * BB;
                const int j = vec ? nnb0 : idx / BB, e = vec ? idx - nnb0 * BB : idx % BB;
                const int a = fold_tab[1 + j], b = fold_tab[2 + j];
                double t = 0.0;
                if (!vec) { for (int q = a; q < b; q++) t += fold_buf[q * SL + e]; S_val[(size_t)rb0 * BB + idx] = t; }
                else { const int v = e / DC, ca = e - v * DC;
                       for (int q = a; q < b; q++) { const double* pv = fold_buf + q * SL; t += (v == 0) ? pv[ca] : (v == 1) ? (pv[DC + ca] + pv[2 * DC + ca]) : (v == 2) ? pv[3 * DC + ca] : pv[4 * DC + ca]; }
                       (v == 0 ? Udiag_w : v == 1 ? rhs : v == 2 ? gcraw_w : Sfc_w)[c * DC + ca] = t; }
            }
        }
        __syncthreads();
    }
    // the accumulation zone of the NEXT iteration (nothing has read it since the iteration before this one ended) is cleared here, a slice
    // per workgroup: the separate memset was a launch of its own on the critical path behind k_publish (4.7 us per iteration)
    if (clear) {
        const size_t per = (clear_len2 + gridDim.x - 1) / gridDim.x, lo = (size_t)c * per, hi = (lo + per < clear_len2) ? lo + per : clear_len2;
        for (size_t e = lo + tid; e < hi; e += blockDim.x) clear[e] = make_double2(0.0, 0.0);
    }
    const int i = pos[c], i2 = pos2[c], rb = row_ptr[c];     // pos / pos2: band rows in camera units (second row: twisted separators)
    band_rows_zero<DC, MERGE>(i, i2, b, MERGE && pair_dummy[c], band);      // (round 4: in front of the damping, ONE barrier for both; the scatter reads precomputed column rows)
    double gmax = 0.0;
    if (tid < DC) {
        double* blk = S_val + ((size_t)rb + diag_slot[c]) * BB;
        const double s = scale_cam[c * 6 + off + tid];
        blk[tid * DC + tid] += (s > 0.0) ? fmin(fmax(Udiag[c * DC + tid], min_diag), max_diag) / radius : 1.0;
        if (s > 0.0) gmax = fabs(gcraw[c * DC + tid] / s);
        Y[(size_t)i * DC + tid] = rhs[c * DC + tid];
        Y[(size_t)Nb * DC + (size_t)i * DC + tid] = Sfc[c * DC + tid];
        if (i2 >= 0) { Y[(size_t)i2 * DC + tid] = 0.0; Y[(size_t)Nb * DC + (size_t)i2 * DC + tid] = 0.0; }
        if (MERGE && pair_dummy[c]) { Y[(size_t)(i + 1) * DC + tid] = 0.0; Y[(size_t)Nb * DC + (size_t)(i + 1) * DC + tid] = 0.0; }
    }
    if (tid < 64) { gmax = wave_max(gmax); if (tid == 0 && gmax > 0.0) atomic_max_nonneg(&scal[(size_t)(c & (SC_NSLOT - 1)) * SC_TOTAL + SC_GMAX], gmax); }
    __syncthreads();                                               // damped diagonal block visible to the whole workgroup
    band_rows_scatter<DC, MERGE>(row_ptr, col_idx, S_val, pos, i, i2, c, b, band, col_pos);
    if (wrap_ptr) band_rows_wrap<DC, MERGE>(wrap_ptr, wrap_blk, wrap_row2, S_val, i, c, b, band);
    if (c == 0 && tid >= 64 && tid < 128) {                        // wave 1 of workgroup 0: focal row from the replicas of the focal sums
        const int l = tid - 64;
        const double* sl = scal + (size_t)(l & (SC_NSLOT - 1)) * SC_TOTAL;
        double fjj, fww, fjr, fwg;
        if (FOLD && la_wave) {
            // (the limbs were loaded at the top of the kernel; load / clear / load / ... in turn made every limb a dependent round trip)
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int j = 0; j < LA_STRIDE; j++) if (lv[k][j] != 0) lacc[((size_t)l * SC_TOTAL + LA_KS[k]) * LA_STRIDE + j] = 0;
            // integer sums over the 64 replicas through LDS: lane t < 32 adds column t = (sum k, limb j) of the 64 rows (192 64-bit butterflies took 3 us)
            __shared__ long long la_buf[64 * 33];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int j = 0; j < LA_STRIDE; j++) la_buf[l * 33 + k * LA_STRIDE + j] = lv[k][j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            long long colsum = 0;
            if (l < 32) for (int r = 0; r < 64; r++) colsum += la_buf[r * 33 + l];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (l < 32) la_buf[l] = colsum;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double fs[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { long long tot[LA_STRIDE];
#pragma unroll
                                          for (int j = 0; j < LA_STRIDE; j++) tot[j] = la_buf[k * LA_STRIDE + j];
                                          fs[k] = lacc_value(tot, tot[LA_NL]); }
            fjj = fs[0]; fww = fs[1]; fjr = fs[2]; fwg = fs[3];
        } else if (lacc) { fjj = fww = 1.0; fjr = fwg = 0.0; } else {      // (lacc without la_wave: the focal length is fixed, the values are not used)
 fjj = wave_sum(sl[SC_FJJ]); fww = wave_sum(sl[SC_FWW]); fjr = wave_sum(sl[SC_FJR]); fwg = wave_sum(sl[SC_FWG]); }
        if (l == 0) {
            const double sf = scale_f[0];
            if (sf > 0.0) {
                Sff[0] = fjj + fmin(fmax(fjj, min_diag), max_diag) / radius - fww;
                rhs[Nc * DC] = fjr - fwg;
                atomic_max_nonneg(&scal[SC_GMAX], fabs(fjr / sf));
            } else { Sff[0] = 1.0; rhs[Nc * DC] = 0.0; }
        }
    }
}

template <int DC>
__global__ void k_band_permute_rhs(const double* __restrict__ rhs, const double* __restrict__ Sfc, const int* __restrict__ pos,
                                   const int* __restrict__ pos2, const unsigned char* __restrict__ pair_dummy, int Nc, int Nb, double* __restrict__ Y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Nc * DC) return;
    const int c = t / DC, a = t - c * DC;
    Y[pos[c] * DC + a] = rhs[t];
    Y[(size_t)Nb * DC + pos[c] * DC + a] = Sfc[t];
    const int i2 = pos2 ? pos2[c] : -1;
    if (i2 >= 0) { Y[i2 * DC + a] = 0.0; Y[(size_t)Nb * DC + i2 * DC + a] = 0.0; }
    if (pair_dummy && pair_dummy[c]) { Y[(pos[c] + 1) * DC + a] = 0.0; Y[(size_t)Nb * DC + (pos[c] + 1) * DC + a] = 0.0; }   // empty partner slot of a merged pair
}

// Factorise in place (lower), store inverse diagonal factors, forward-substitute NR right-hand sides Y[r][N*DC].
template <int DC, int NR>
__global__ void __launch_bounds__(1024)
k_band_chol(double* __restrict__ band, double* __restrict__ Linv_out, double* __restrict__ Y, const int* __restrict__ pairs,
            int N, int b, int* __restrict__ fail_flag) {
    constexpr int BB = DC * DC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sLinv = lds;                // BB
    double* sDiag = lds + BB;           // BB
    double* sY = sDiag + BB;            // NR*DC
    double* sPanel = sY + NR * DC;      // b*BB
    const int W = b + 1, n = N * DC, tid = threadIdx.x;
    for (int j = 0; j < N; j++) {
        const int nb = min(b, N - 1 - j);
        // ---- A: diagonal block
        if (tid < BB) sDiag[tid] = band[((size_t)j * W) * BB + tid];
        __syncthreads();
        if (tid == 0) {
            double L[DC][DC], Li[DC][DC];
#pragma unroll
            for (int r = 0; r < DC; r++)
#pragma unroll
                for (int c = 0; c < DC; c++) { L[r][c] = 0.0; Li[r][c] = 0.0; }
            bool bad = false;
#pragma unroll
            for (int c = 0; c < DC; c++) {
                double d = sDiag[c * DC + c];
#pragma unroll
                for (int k = 0; k < c; k++) d -= L[c][k] * L[c][k];
                if (!(d > 0.0)) { bad = true; d = 1.0; }
                d = sqrt(d); L[c][c] = d;
#pragma unroll
                for (int r = c + 1; r < DC; r++) {
                    double s = sDiag[r * DC + c];
#pragma unroll
                    for (int k = 0; k < c; k++) s -= L[r][k] * L[c][k];
                    L[r][c] = s / d;
                }
            }
#pragma unroll
            for (int c = 0; c < DC; c++) {
                Li[c][c] = 1.0 / L[c][c];
#pragma unroll
                for (int r = c + 1; r < DC; r++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = c; k < r; k++) s -= L[r][k] * Li[k][c];
                    Li[r][c] = s / L[r][r];
                }
            }
#pragma unroll
            for (int r = 0; r < DC; r++)
#pragma unroll
                for (int c = 0; c < DC; c++) { sLinv[r * DC + c] = Li[r][c]; Linv_out[(size_t)j * BB + r * DC + c] = Li[r][c]; band[((size_t)j * W) * BB + r * DC + c] = L[r][c]; }
            if (bad) *fail_flag = 1;
        }
        __syncthreads();
        // ---- B: panel rows  L_ij = A_ij L_jj^-T  (one lane per block row), and y_j = L_jj^-1 r_j
        for (int t = tid; t < nb * DC; t += 1024) {
            const int blk = t / DC, a = t - blk * DC;
            const size_t addr = (((size_t)(j + 1 + blk)) * W + (blk + 1)) * BB + a * DC;
            double A[DC];
#pragma unroll
            for (int k = 0; k < DC; k++) A[k] = band[addr + k];
#pragma unroll
            for (int c = 0; c < DC; c++) { double s = 0.0;
#pragma unroll
                for (int k = 0; k <= c; k++) s += A[k] * sLinv[c * DC + k];
                band[addr + c] = s; sPanel[blk * BB + a * DC + c] = s; }
        }
        if (tid >= 1024 - NR * 64 && (tid & 63) == 0) {
            const int r = (tid - (1024 - NR * 64)) >> 6;
            const size_t addr = (size_t)r * n + (size_t)j * DC;
            double v[DC];
#pragma unroll
            for (int k = 0; k < DC; k++) v[k] = Y[addr + k];
#pragma unroll
            for (int a = 0; a < DC; a++) { double s = 0.0;
#pragma unroll
                for (int k = 0; k <= a; k++) s += sLinv[a * DC + k] * v[k];
                Y[addr + a] = s; sY[r * DC + a] = s; }
        }
        __syncthreads();
        // ---- C: trailing window  A_ik -= L_ij L_kj^T  (one lane per (pair, row)), r_k -= L_kj y_j
        const int work = (nb * (nb + 1) / 2) * DC;
        for (int t = tid; t < work; t += 1024) {
            const int pr = t / DC, a = t - pr * DC;
            const int pk = pairs[pr]; const int ir = pk & 0xffff, kr = pk >> 16;     // 1-based offsets from j
            const double* Li_ = sPanel + (ir - 1) * BB + a * DC;
            const double* Lk_ = sPanel + (kr - 1) * BB;
            double la[DC];
#pragma unroll
            for (int m = 0; m < DC; m++) la[m] = Li_[m];
            double* dst = band + (((size_t)(j + ir)) * W + (ir - kr)) * BB + a * DC;
#pragma unroll
            for (int c = 0; c < DC; c++) { double s = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) s += la[m] * Lk_[c * DC + m];
                dst[c] -= s; }
        }
        for (int t = tid; t < nb * DC * NR; t += 1024) {
            const int r = t / (nb * DC), q = t - r * nb * DC, kr = q / DC, a = q - kr * DC;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) s += sPanel[kr * BB + a * DC + m] * sY[r * DC + m];
            Y[(size_t)r * n + (size_t)(j + 1 + kr) * DC + a] -= s;
        }
        __syncthreads();
    }
}

// forward substitution with an existing factor: Y <- L^-1 Y  (NR right-hand sides)
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_band_fwd(const double* __restrict__ band, const double* __restrict__ Linv, double* __restrict__ Y, int N, int b) {
    constexpr int BB = DC * DC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;                       // ring of the last b solutions: [b][NR*DC]
    double* sPart = sX + (size_t)b * NR * DC;   // [b][NR*DC] partial products
    double* sAcc = sPart + (size_t)b * NR * DC; // NR*DC
    const int W = b + 1, n = N * DC, tid = threadIdx.x;
    for (int j = 0; j < N; j++) {
        const int nb = min(b, j);
        for (int t = tid; t < nb * DC * NR; t += blockDim.x) {
            const int r = t / (nb * DC), q = t - r * nb * DC, d = q / DC + 1, a = q - (d - 1) * DC;
            const double* blk = band + (((size_t)j) * W + d) * BB + a * DC;       // row a of block (j, j-d)
            const double* x = sX + ((size_t)((j - d) % b) * NR + r) * DC;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) s += blk[m] * x[m];
            sPart[((size_t)(d - 1) * NR + r) * DC + a] = s;
        }
        __syncthreads();
        if (tid < NR * DC) {
            const int r = tid / DC, a = tid - r * DC;
            double s = Y[(size_t)r * n + (size_t)j * DC + a];
            for (int d = 0; d < nb; d++) s -= sPart[((size_t)d * NR + r) * DC + a];
            sAcc[tid] = s;
        }
        __syncthreads();
        if (tid < NR * DC) {
            const int r = tid / DC, a = tid - r * DC;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < DC; k++) if (k <= a) s += Linv[(size_t)j * BB + a * DC + k] * sAcc[r * DC + k];
            Y[(size_t)r * n + (size_t)j * DC + a] = s;
            if (b > 0) sX[((size_t)(j % b) * NR + r) * DC + a] = s;
        }
        __syncthreads();
    }
}

// backward substitution: Y <- L^-T Y
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_band_back(const double* __restrict__ band, const double* __restrict__ Linv, double* __restrict__ Y, int N, int b) {
    constexpr int BB = DC * DC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;
    double* sPart = sX + (size_t)b * NR * DC;
    double* sAcc = sPart + (size_t)b * NR * DC;
    const int W = b + 1, n = N * DC, tid = threadIdx.x;
    for (int j = N - 1; j >= 0; j--) {
        const int nb = min(b, N - 1 - j);
        for (int t = tid; t < nb * DC * NR; t += blockDim.x) {
            const int r = t / (nb * DC), q = t - r * nb * DC, d = q / DC + 1, a = q - (d - 1) * DC;
            const double* blk = band + (((size_t)(j + d)) * W + d) * BB;          // block (j+d, j); need column a of it
            const double* x = sX + ((size_t)((j + d) % b) * NR + r) * DC;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) s += blk[m * DC + a] * x[m];
            sPart[((size_t)(d - 1) * NR + r) * DC + a] = s;
        }
        __syncthreads();
        if (tid < NR * DC) {
            const int r = tid / DC, a = tid - r * DC;
            double s = Y[(size_t)r * n + (size_t)j * DC + a];
            for (int d = 0; d < nb; d++) s -= sPart[((size_t)d * NR + r) * DC + a];
            sAcc[tid] = s;
        }
        __syncthreads();
        if (tid < NR * DC) {
            const int r = tid / DC, a = tid - r * DC;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < DC; k++) if (k >= a) s += Linv[(size_t)j * BB + k * DC + a] * sAcc[r * DC + k];   // L_jj^-T
            Y[(size_t)r * n + (size_t)j * DC + a] = s;
            if (b > 0) sX[((size_t)(j % b) * NR + r) * DC + a] = s;
        }
        __syncthreads();
    }
}

// Arrow combine for the focal border:  [B s; s^T sigma] [y; phi] = [r; rho]
//   v = B^-1 r (V, permuted), u = B^-1 s (U, permuted), phi = (rho - s.v)/(sigma - s.u), y = v - u phi
// Writes the un-permuted result out[c*DC+a], out[n] = phi.
template <int DC>
__global__ void __launch_bounds__(1024)
k_band_combine(const double* __restrict__ V, const double* __restrict__ U, const double* __restrict__ Sfc, const double* __restrict__ Sff,
               const double* __restrict__ rho_ptr, const int* __restrict__ pos, int Nc, double* __restrict__ out) {
    __shared__ double red[2 * 16];
    __shared__ double sphi;
    const int n = Nc * DC;
    double acc[2] = {0, 0};
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int c = t / DC, a = t - c * DC; const int pi = pos[c] * DC + a;
        acc[0] += Sfc[t] * V[pi]; acc[1] += Sfc[t] * U[pi];
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) sphi = (rho_ptr[0] - acc[0]) / (Sff[0] - acc[1]);
    __syncthreads();
    const double phi = sphi;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int c = t / DC, a = t - c * DC; const int pi = pos[c] * DC + a;
        out[t] = V[pi] - U[pi] * phi;
    }
    if (threadIdx.x == 0) out[n] = phi;
}

// ---- refinement PCG around the banded preconditioner (single-workgroup vector kernels) -----------------
// r = b - q ; flags
template <int DC>
__global__ void __launch_bounds__(1024)
k_ref_residual(const double* __restrict__ bvec, const double* __restrict__ q, const double* __restrict__ x, const double* __restrict__ Sfc,
               const double* __restrict__ Sff, int Nc, double tol2, double* __restrict__ r, double* __restrict__ pcg) {
    __shared__ double red[2 * 16];
    __shared__ double sqf;
    const int n = Nc * DC;
    double a0[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) a0[0] += Sfc[i] * x[i];
    block_sum<1>(a0, red);
    if (threadIdx.x == 0) sqf = Sff[0] * x[n] + a0[0];
    __syncthreads();
    double a2[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        const double qi = (i < n) ? q[i] : sqf;
        const double ri = bvec[i] - qi; r[i] = ri;
        a2[0] += ri * ri; a2[1] += bvec[i] * bvec[i];
    }
    block_sum<2>(a2, red);
    if (threadIdx.x == 0) {
        pcg[PCG_RR] = a2[0]; pcg[PCG_BN2] = a2[1]; pcg[PCG_ITERS] = 0.0; pcg[PCG_BREAKDOWN] = 0.0;
        pcg[PCG_DONE] = (a2[0] <= tol2 * a2[1]) ? 1.0 : 0.0;
    }
}
// direction: rz = r.z ; p = z + (rz/rz_old) p  (first = 1: p = z)
static __global__ void __launch_bounds__(1024)
k_ref_direction(const double* __restrict__ r, const double* __restrict__ z, int n1, int first, double* __restrict__ p, double* __restrict__ pcg) {
    __shared__ double red[16];
    __shared__ double sbeta;
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n1; i += blockDim.x) acc[0] += r[i] * z[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) { sbeta = first ? 0.0 : acc[0] / pcg[PCG_RZ]; pcg[PCG_RZ] = acc[0]; }
    __syncthreads();
    const double beta = sbeta;
    for (int i = threadIdx.x; i < n1; i += blockDim.x) p[i] = z[i] + (first ? 0.0 : beta * p[i]);
}
// step: q_f, alpha = rz / p.q ; x += alpha p ; r -= alpha q ; rr ; done?
template <int DC>
__global__ void __launch_bounds__(1024)
k_ref_step(const double* __restrict__ Sfc, const double* __restrict__ Sff, int Nc, double tol2, const double* __restrict__ p,
           const double* __restrict__ q, const double* __restrict__ pqpart, double* __restrict__ x, double* __restrict__ r, double* __restrict__ pcg) {
    __shared__ double red[2 * 16];
    __shared__ double sh[2];
    const int n = Nc * DC;
    const double pf = p[n];
    double acc[2] = {0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc[0] += Sfc[i] * p[i];
    for (int c = threadIdx.x; c < Nc; c += blockDim.x) acc[1] += pqpart[c];
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) { const double qf = Sff[0] * pf + acc[0]; sh[0] = qf; sh[1] = acc[1] + pf * qf; }
    __syncthreads();
    const double qf = sh[0], pq = sh[1];
    if (!(pq > 0.0)) { if (threadIdx.x == 0) { pcg[PCG_DONE] = 1.0; pcg[PCG_BREAKDOWN] = 1.0; } return; }
    const double alpha = pcg[PCG_RZ] / pq;
    double a1[1] = {0.0};
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        const double qi = (i < n) ? q[i] : qf;
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * qi; r[i] = ri; a1[0] += ri * ri;
    }
    block_sum<1>(a1, red);
    if (threadIdx.x == 0) { pcg[PCG_RR] = a1[0]; pcg[PCG_ITERS] += 1.0; if (a1[0] <= tol2 * pcg[PCG_BN2]) pcg[PCG_DONE] = 1.0; }
}

}  // namespace ssfm

// =====================================================================================================
// Substitution kernels for factors produced by k_band_chol_v2 (band_kernels2.h): one workgroup per connected component of
// the camera graph (comp_ptr: contiguous position ranges from the Cuthill-McKee pass).  The forward kernel serves the PCG
// refinement path; the back kernel is the fallback for bands too wide for the single-wave k_band_back_v2.
// =====================================================================================================
namespace ssfm {

// back substitution Y <- L^-T Y per component; next column of L prefetched into registers while the current step runs
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_band_back_lds(const double* __restrict__ band, const double* __restrict__ Linv, double* __restrict__ Y, const int* __restrict__ comp_ptr,
                int N, int b) {
    constexpr int BB = DC * DC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;                               // ring [b][NR*DC]
    double* sPart = sX + (size_t)b * NR * DC;       // [b][NR*DC]
    double* sAcc = sPart + (size_t)b * NR * DC;     // NR*DC
    const int W = b + 1, n = N * DC, tid = threadIdx.x;
    const int r0 = comp_ptr[blockIdx.x], r1 = comp_ptr[blockIdx.x + 1];
    const int d = tid / DC + 1, a = tid - (d - 1) * DC;     // this lane's (offset, column) of the L column, valid if tid < b*DC
    const bool has = tid < b * DC;
    double col[DC], li[DC], yv = 0.0;
    auto fetch = [&](int j) {
#pragma unroll
        for (int m = 0; m < DC; m++) col[m] = (has && j >= r0 && j + d < r1) ? band[(((size_t)(j + d)) * W + d) * BB + m * DC + a] : 0.0;
        if (tid < NR * DC && j >= r0) {
            const int aa = tid % DC;
#pragma unroll
            for (int k = 0; k < DC; k++) li[k] = (k >= aa) ? Linv[(size_t)j * BB + k * DC + aa] : 0.0;      // column aa of L_jj^-1 = row of L^-T
            yv = Y[(size_t)(tid / DC) * n + (size_t)j * DC + aa];
        }
    };
    fetch(r1 - 1);
    int jm = (b > 0) ? (r1 - 1) % b : 0;                     // ring slot of row j, kept incrementally
    for (int j = r1 - 1; j >= r0; j--, jm = (jm == 0) ? max(b - 1, 0) : jm - 1) {
        const int nb = min(b, r1 - 1 - j);
        double cur[DC], curli[DC]; const double cury = yv;
#pragma unroll
        for (int m = 0; m < DC; m++) { cur[m] = col[m]; curli[m] = li[m]; }
        fetch(j - 1);                                   // in flight during this step
        if (has && d <= nb) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int xs = (jm + d >= b) ? jm + d - b : jm + d;
                const double* x = sX + ((size_t)xs * NR + r) * DC;
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) s += cur[m] * x[m];
                sPart[((size_t)(d - 1) * NR + r) * DC + a] = s;
            }
        }
        lds_barrier();
        if (tid < NR * DC) {
            const int r = tid / DC, aa = tid - r * DC;
            double s = cury;
            for (int dd = 0; dd < nb; dd++) s -= sPart[((size_t)dd * NR + r) * DC + aa];
            sAcc[tid] = s;
        }
        lds_barrier();
        if (tid < NR * DC) {
            const int r = tid / DC, aa = tid - r * DC;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < DC; k++) s += curli[k] * sAcc[r * DC + k];
            Y[(size_t)r * n + (size_t)j * DC + aa] = s;
            if (b > 0) sX[((size_t)jm * NR + r) * DC + aa] = s;
        }
        lds_barrier();
    }
}

// forward substitution Y <- L^-1 Y per component with the same prefetch scheme (PCG refinement applications)
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_band_fwd_lds(const double* __restrict__ band, const double* __restrict__ Linv, double* __restrict__ Y, const int* __restrict__ comp_ptr,
               int N, int b) {
    constexpr int BB = DC * DC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;
    double* sPart = sX + (size_t)b * NR * DC;
    double* sAcc = sPart + (size_t)b * NR * DC;
    const int W = b + 1, n = N * DC, tid = threadIdx.x;
    const int r0 = comp_ptr[blockIdx.x], r1 = comp_ptr[blockIdx.x + 1];
    const int d = tid / DC + 1, a = tid - (d - 1) * DC;
    const bool has = tid < b * DC;
    double row[DC], li[DC], yv = 0.0;
    auto fetch = [&](int j) {
#pragma unroll
        for (int m = 0; m < DC; m++) row[m] = (has && j < r1 && j - d >= r0) ? band[(((size_t)j) * W + d) * BB + a * DC + m] : 0.0;
        if (tid < NR * DC && j < r1) {
            const int aa = tid % DC;
#pragma unroll
            for (int k = 0; k < DC; k++) li[k] = (k <= aa) ? Linv[(size_t)j * BB + aa * DC + k] : 0.0;
            yv = Y[(size_t)(tid / DC) * n + (size_t)j * DC + aa];
        }
    };
    fetch(r0);
    int jm = (b > 0) ? r0 % b : 0;
    for (int j = r0; j < r1; j++, jm = (jm + 1 >= b) ? 0 : jm + 1) {
        const int nb = min(b, j - r0);
        double cur[DC], curli[DC]; const double cury = yv;
#pragma unroll
        for (int m = 0; m < DC; m++) { cur[m] = row[m]; curli[m] = li[m]; }
        fetch(j + 1);
        if (has && d <= nb) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int xs = (jm - d < 0) ? jm - d + b : jm - d;
                const double* x = sX + ((size_t)xs * NR + r) * DC;
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) s += cur[m] * x[m];
                sPart[((size_t)(d - 1) * NR + r) * DC + a] = s;
            }
        }
        lds_barrier();
        if (tid < NR * DC) {
            const int r = tid / DC, aa = tid - r * DC;
            double s = cury;
            for (int dd = 0; dd < nb; dd++) s -= sPart[((size_t)dd * NR + r) * DC + aa];
            sAcc[tid] = s;
        }
        lds_barrier();
        if (tid < NR * DC) {
            const int r = tid / DC, aa = tid - r * DC;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < DC; k++) s += curli[k] * sAcc[r * DC + k];
            Y[(size_t)r * n + (size_t)j * DC + aa] = s;
            if (b > 0) sX[((size_t)jm * NR + r) * DC + aa] = s;
        }
        lds_barrier();
    }
}

}  // namespace ssfm
