// spherical_sfm_amd -- third-generation block-banded Cholesky (6x6 blocks): the trailing window lives in REGISTERS.
//
// k_band_chol_v2 (band_kernels2.h) keeps the (b+1)^2-block window in LDS; its step was bound by the LDS traffic of the
// trailing update (read-modify-write of 55 blocks + 36 operand reads per 54 multiply-adds: ~135 KB per step) and by the
// 1.7k-cycle factor-and-invert of the next diagonal block on the look-ahead wave (s_memtime stamps).
//
// Here a block (i, i-d) of the band sits at ring slot (i mod R, d), R = b+1, for its whole life in the window, so ONE LANE
// owns it: 36 accumulators that never move.  Per step a live lane reads the two panel blocks it needs (72 doubles as 36
// ds_read_b128 from a transposed, bank-padded copy of the panel) and issues 216 multiply-adds -- no window traffic at all.
// When a block's column is the next pivot its lane drops it into sA (the next panel's input); when its row leaves the
// window the lane takes the block of the row that enters from a staging row the loader wave filled one step earlier.
// Only the diagonal blocks (the dependent chain) stay in LDS.  The diagonal block is factored by Gauss-Jordan on [A | I]
// with one lane per row: G = D^-1/2 L1^-1 falls out of the elimination, without the separate inverse pass.
//
// Roles (physical wave p runs on SIMD p mod 4): wave 0 look-ahead factor, alone on SIMD 0 with the loader (p = 4); the
// trailing waves on SIMDs 1..3 first; one wave for the other diagonal blocks + right-hand sides; one writer.
#pragma once
#include "band_kernels2.h"

namespace ssfm {

constexpr int C3_PS = 38;          // doubles per panel block in LDS (36 padded: ten blocks 304 B apart spread over 8 of 8 16-byte bank groups)
typedef double c3_d2 __attribute__((ext_vector_type(2)));

// Lane r (< 6) enters with row r of the SPD block in row[] (other lanes: zeros); on exit lane r holds row r of G = L^-1
// (g[c] = G[r][c], zero above the diagonal).  Gaussian elimination on [A | I]: M A = D L1^T with M = L1^-1, G = D^-1/2 M.
__device__ __forceinline__ bool wave_ldl_inverse6(double (&row)[6], double (&g)[6]) {
    const int lane = threadIdx.x & 63;
    double dl = 1.0; bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; c++) g[c] = (lane == c) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double d = lane_bcast(row[c], c);
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        if (lane == c) dl = d;
        if (c < 5) {
            const double rinv = fast_rcp(d);
            const double f = (lane > c) ? row[c] * rinv : 0.0;
#pragma unroll
            for (int c2 = c + 1; c2 < 6; c2++) row[c2] -= f * lane_bcast(row[c2], c);
#pragma unroll
            for (int c2 = 0; c2 < c; c2++) g[c2] -= f * lane_bcast(g[c2], c);
            g[c] = (lane == c) ? 1.0 : -f;
        }
    }
    const double rs = fast_rsqrt(dl);
#pragma unroll
    for (int c = 0; c < 6; c++) g[c] *= rs;
    return ok;
}

// LDS doubles of k_band_chol_v3
inline size_t chol3_lds_doubles(int b, int NR) {
    const size_t R = b + 1, BB = 36, RWP = R * BB + (size_t)NR * 6;
    return R * BB + 2 * (size_t)b * C3_PS + (size_t)b * BB + 2 * RWP + R * NR * 6 + NR * 6 + 2 * BB;
}
inline int chol3_trailing_waves(int b) { return ((b + 1) * b + 63) / 64; }

//   band  [N][b+1][36]  in/out: block d of row i = (i, i-d); off-diagonal blocks leave as L, diagonal blocks are left alone
//   Ginv  [N][36]       out: L_jj^-1 (row-major, lower)
//   Y     [NR][N*6]     in/out: right-hand sides -> L^-1 Y
// Same tables and meaning as k_band_chol_v2: pivots [piv_lo, piv_hi), window to win_hi, merge_from, await2 / signal / flags.
// TW trailing waves (64 TW >= (b+1) b), PRE loader registers per lane (64 PRE >= (b+1) 36 + 6 NR).  blockDim.x = 64 (TW + 4).
#ifdef CHOL3_STAMPS
__device__ long long* g_chol3_stamps;
#define CHOL3_STAMPJ(jv_, slot_) do { if (blockIdx.x == 0 && (jv_) == r0 + 10 && lane == 0) g_chol3_stamps[role * 8 + (slot_)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CHOL3_STAMPJ(jv_, slot_) do { } while (0)
#endif
#define CHOL3_STAMP(slot_) CHOL3_STAMPJ(j, slot_)
template <int NR, int TW, int PRE>
__global__ void __launch_bounds__(64 * (TW + 4))
k_band_chol_v3(double* __restrict__ band, double* __restrict__ Ginv, double* __restrict__ Y,
               const int* __restrict__ piv_lo, const int* __restrict__ piv_hi, const int* __restrict__ win_hi,
               const int* __restrict__ merge_from, int N, int b, int* __restrict__ fail_flag,
               const int* __restrict__ await2 = nullptr, const int* __restrict__ signal = nullptr, int* __restrict__ flags = nullptr, int seq = 0) {
    constexpr int DC = 6, BB = 36, PS = C3_PS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int R = b + 1, RW = R * BB, RWP = RW + NR * DC;
    double* sDg = lds;                                      // [R][BB]      diagonal blocks of the window rows (ring)
    double* sP = sDg + (size_t)R * BB;                      // [b][PS]      panel, row-major: X_k[a][m]
    double* sPT = sP + (size_t)b * PS;                      // [b][PS]      panel, transposed: X_k[m][a] at m*6 + a
    double* sA = sPT + (size_t)b * PS;                      // [b][BB]      blocks of the NEXT pivot column, as their lanes left them
    double* sStage = sA + (size_t)b * BB;                   // [2][RWP]     image of the row that enters the window (band row + right-hand sides)
    double* sYr = sStage + (size_t)2 * RWP;                 // [R][NR][DC]  right-hand-side rows of the window
    double* sYj = sYr + (size_t)R * NR * DC;                // [NR][DC]     final y_j
    double* sG = sYj + NR * DC;                             // [BB]         inverse factor of the current diagonal block
    const int n = N * DC, nt = blockDim.x, lane = threadIdx.x & 63, tid = threadIdx.x, nw = TW + 4;
    // role of this physical wave: 0 look-ahead | 1..TW trailing | TW+1 diagonal blocks + right-hand sides | TW+2 writer | TW+3 loader
    const int pw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int role;
    if (pw == 0) role = 0;
    else if (pw == 4) role = TW + 3;
    else if ((pw & 3) != 0) role = 1 + (pw - 1) - ((pw - 1) >> 2);
    else role = 1 + ((nw - 1) - ((nw - 1) >> 2)) + ((pw >> 2) - 2);
    const int r0 = piv_lo[blockIdx.x], r1 = piv_hi[blockIdx.x], re = win_hi[blockIdx.x];
    const int sig = signal ? signal[blockIdx.x] : -1, aw = await2 ? await2[blockIdx.x] : -1;
    if (r0 >= r1) { if (sig >= 0 && tid == 0) __hip_atomic_store(flags + sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (aw >= 0) {
        if (tid == 0) { while (__hip_atomic_load(flags + aw, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(8);
                        while (__hip_atomic_load(flags + aw + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(8); }
        __syncthreads(); __threadfence();
    }
    const int mf = merge_from ? merge_from[blockIdx.x] : -1;
    const int jm0 = r0 % R;
    // ---- trailing lanes: slot t = rho * b + (d - 1) of the ring; acc = block (i, i - d) of the row i currently at ring slot rho
    const int ti = role - 1, t = ti * 64 + lane;
    const bool is_tr = role >= 1 && role <= TW, valid = is_tr && t < R * b;
    const int rho = valid ? t / b : 0, d = valid ? t - rho * b + 1 : 1;
    double acc[6][6];
    if (is_tr) {
        int rel = rho - jm0; if (rel < 0) rel += R;
        const int i = r0 + rel;
        const bool in_win = valid && i < re && d <= rel;
        const double* src = band + ((size_t)(in_win ? i : r0) * R + (in_win ? d : 0)) * BB;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++) acc[a][c] = src[a * 6 + c];
        if (mf >= 0) {
            // separator of a twisted component: block (s, s-d) also takes the transpose of the copy's block (b-1-s+d, d)
            const double* src2 = band + ((size_t)(mf + b - 1 - (in_win ? rel - d : 0)) * R + (in_win ? d : 0)) * BB;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 6; c++) acc[a][c] += src2[c * 6 + a];
        }
        if (!in_win) {
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 6; c++) acc[a][c] = 0.0;
        } else if (rel == d) {                              // column r0: input of the first panel
            double* dst = sA + (size_t)(rel - 1) * BB;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 6; c += 2) *reinterpret_cast<c3_d2*>(dst + a * 6 + c) = c3_d2{acc[a][c], acc[a][c + 1]};
        }
    } else {
        // everyone else: diagonal blocks + right-hand sides of the first window, and the image of row r0 + R
        const int ntr = TW * 64, ot = (role == 0) ? lane : (role - TW) * 64 + lane, on = nt - ntr;      // dense index over the non-trailing threads
        const int nrow0 = min(r0 + R, re) - r0;
        for (int idx = ot; idx < nrow0 * BB; idx += on) {
            const int s = idx / BB, e = idx - s * BB;
            double v = band[(size_t)(r0 + s) * RW + e];
            if (mf >= 0) { const int a = e / DC, a2 = e - a * DC; v += band[(size_t)(mf + b - 1 - s) * RW + a2 * DC + a]; }
            sDg[(size_t)((r0 + s) % R) * BB + e] = v;
        }
        for (int idx = ot; idx < nrow0 * NR * DC; idx += on) {
            const int s = idx / (NR * DC), e = idx - s * (NR * DC);
            double v = Y[(size_t)(e / DC) * n + (size_t)(r0 + s) * DC + (e % DC)];
            if (mf >= 0) v += Y[(size_t)(e / DC) * n + (size_t)(mf + b - 1 - s) * DC + (e % DC)];
            sYr[(size_t)((r0 + s) % R) * NR * DC + e] = v;
        }
        if (r0 + R < re) {
            double* img = sStage + (size_t)(r0 & 1) * RWP;
            for (int e = ot; e < RWP; e += on) {
                const int q = e - RW;
                img[e] = (e < RW) ? band[(size_t)(r0 + R) * RW + e] : Y[(size_t)(q / DC) * n + (size_t)(r0 + R) * DC + (q % DC)];
            }
        }
    }
    __syncthreads();
    if (role == 0) {                                        // factor the first diagonal block
        double row[6], g[6];
        const double* D0 = sDg + (size_t)jm0 * BB;
#pragma unroll
        for (int c = 0; c < 6; c++) row[c] = (lane < 6) ? D0[lane * 6 + c] : 0.0;
        if (!wave_ldl_inverse6(row, g) && lane == 0) *fail_flag = 1;
        if (lane < 6) {
#pragma unroll
            for (int c = 0; c < 6; c += 2) *reinterpret_cast<c3_d2*>(sG + lane * 6 + c) = c3_d2{g[c], g[c + 1]};
        }
    }
    __syncthreads();
    // ---- phase B, every role: panel X_k = A_k G^T (both layouts) and y_j = G y_j
    auto phaseB = [&](int jm, int nb) {
        for (int e = tid; e < nb * BB; e += nt) {
            const int k = e / BB, rc = e - k * BB, a = rc / DC, c = rc - a * DC;
            const double* A = sA + (size_t)k * BB + a * DC;
            const double* Gc = sG + c * DC;
            double x = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) x += A[m] * Gc[m];
            sP[k * PS + a * DC + c] = x;
            sPT[k * PS + c * DC + a] = x;
        }
        if (tid >= nt - 64 && tid < nt - 64 + NR * DC) {
            const int q = tid - (nt - 64), r = q / DC, c = q - r * DC;
            const double* yr = sYr + (size_t)jm * NR * DC + r * DC;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) s += sG[c * DC + m] * yr[m];
            sYj[q] = s;
        }
    };
    if (role == 0) {
        // ---- look-ahead: D(j+1) -= X_1 X_1^T in row layout straight from the panel, factor, inverse
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            CHOL3_STAMP(0);
            phaseB(jm, nb);
            CHOL3_STAMP(1);
            lds_barrier();
            CHOL3_STAMP(2);
            if (j + 1 < r1) {
                int s1 = jm + 1; if (s1 >= R) s1 -= R;
                const int lr = min(lane, 5);
                const double* drow = sDg + (size_t)s1 * BB + lr * 6;
                double row[6], g[6], xr[6];
#pragma unroll
                for (int c = 0; c < 6; c += 2) { const c3_d2 v = *reinterpret_cast<const c3_d2*>(drow + c); row[c] = v.x; row[c + 1] = v.y;
                                                 const c3_d2 w = *reinterpret_cast<const c3_d2*>(sP + lr * 6 + c); xr[c] = w.x; xr[c + 1] = w.y; }
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double xc[6];
#pragma unroll
                    for (int m = 0; m < 6; m += 2) { const c3_d2 w = *reinterpret_cast<const c3_d2*>(sP + c * 6 + m); xc[m] = w.x; xc[m + 1] = w.y; }
#pragma unroll
                    for (int m = 0; m < 6; m++) row[c] -= xr[m] * xc[m];
                }
                if (lane >= 6) {
#pragma unroll
                    for (int c = 0; c < 6; c++) row[c] = 0.0;
                }
                CHOL3_STAMP(5);
                if (!wave_ldl_inverse6(row, g) && lane == 0) *fail_flag = 1;
                if (lane < 6) {
#pragma unroll
                    for (int c = 0; c < 6; c += 2) *reinterpret_cast<c3_d2*>(sG + lane * 6 + c) = c3_d2{g[c], g[c + 1]};
                }
            }
            CHOL3_STAMP(3);
            lds_barrier();
            CHOL3_STAMP(4);
        }
    } else if (is_tr) {
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            CHOL3_STAMP(0);
            phaseB(jm, nb);
            CHOL3_STAMP(1);
            lds_barrier();
            CHOL3_STAMP(2);
            int rel = rho - jm; if (rel < 0) rel += R;
            const int kr = rel - d;
            const bool live = valid && kr >= 1 && j + rel < re;
            if (live) {
                const double* pi = sPT + (size_t)(rel - 1) * PS;
                const double* pk = sPT + (size_t)(kr - 1) * PS;
#pragma unroll
                for (int m = 0; m < 6; m++) {
                    double la[6], lk[6];
#pragma unroll
                    for (int u = 0; u < 6; u += 2) { const c3_d2 v = *reinterpret_cast<const c3_d2*>(pi + m * 6 + u); la[u] = v.x; la[u + 1] = v.y;
                                                     const c3_d2 w = *reinterpret_cast<const c3_d2*>(pk + m * 6 + u); lk[u] = w.x; lk[u + 1] = w.y; }
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int c = 0; c < 6; c++) acc[a][c] -= la[a] * lk[c];
                }
                if (kr == 1) {                               // its column is the next pivot: hand the block to the next panel
                    double* dst = sA + (size_t)(rel - 2) * BB;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int c = 0; c < 6; c += 2) *reinterpret_cast<c3_d2*>(dst + a * 6 + c) = c3_d2{acc[a][c], acc[a][c + 1]};
                }
            } else if (valid && rel == 0 && j + R < re) {    // the pivot row leaves: its slot takes block (j + R, j + R - d)
                const double* src = sStage + (size_t)(j & 1) * RWP + (size_t)d * BB;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int c = 0; c < 6; c += 2) { const c3_d2 v = *reinterpret_cast<const c3_d2*>(src + a * 6 + c); acc[a][c] = v.x; acc[a][c + 1] = v.y; }
                if (d == b) {                                // block (j + R, j + 1) enters as an input of the very next panel
                    double* dst = sA + (size_t)(b - 1) * BB;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int c = 0; c < 6; c += 2) *reinterpret_cast<c3_d2*>(dst + a * 6 + c) = c3_d2{acc[a][c], acc[a][c + 1]};
                }
            }
            CHOL3_STAMP(3);
            lds_barrier();
            CHOL3_STAMP(4);
        }
    } else if (role == TW + 1) {
        // ---- the other diagonal blocks of the window (lower triangle computed, both halves written) + right-hand sides
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            CHOL3_STAMP(0);
            phaseB(jm, nb);
            CHOL3_STAMP(1);
            lds_barrier();
            CHOL3_STAMP(2);
            const int k0 = (j + 1 < r1) ? 1 : 0;             // block (j+1, j+1) belongs to the look-ahead wave while it is a pivot
            for (int q = lane; q < (nb - k0) * 21; q += 64) {
                const int kk = q / 21, tr = q - kk * 21, k = kk + k0;
                int a = 0; while ((a + 1) * (a + 2) / 2 <= tr) a++;
                const int c = tr - a * (a + 1) / 2;
                int s = jm + 1 + k; if (s >= R) s -= R;
                const double* X = sP + (size_t)k * PS;
                double v = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) v += X[a * 6 + m] * X[c * 6 + m];
                double* D = sDg + (size_t)s * BB;
                const double nv = D[a * 6 + c] - v;
                D[a * 6 + c] = nv; D[c * 6 + a] = nv;
            }
            for (int qq = lane; qq < nb * DC; qq += 64) {
                const int kr = qq / DC + 1, a = qq - (kr - 1) * DC;
                int sk = jm + kr; if (sk >= R) sk -= R;
                const double* Lk_ = sP + (size_t)(kr - 1) * PS + a * DC;
#pragma unroll
                for (int r = 0; r < NR; r++) { double v = 0.0;
#pragma unroll
                    for (int m = 0; m < DC; m++) v += Lk_[m] * sYj[r * DC + m];
                    sYr[(size_t)sk * NR * DC + r * DC + a] -= v; }
            }
            CHOL3_STAMP(3);
            lds_barrier();
            CHOL3_STAMP(4);
        }
    } else if (role == TW + 2) {
        // ---- writer: panel, y_j and G to global memory (stores only, never waited on)
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            CHOL3_STAMP(0);
            phaseB(jm, nb);
            if (lane < BB) Ginv[(size_t)j * BB + lane] = sG[lane];      // before the look-ahead wave replaces it
            CHOL3_STAMP(1);
            lds_barrier();
            CHOL3_STAMP(2);
            for (int e = lane; e < nb * BB; e += 64) { const int k = e / BB, rc = e - k * BB; band[((size_t)(j + 1 + k) * R + (k + 1)) * BB + rc] = sP[k * PS + rc]; }
            if (lane < NR * DC) Y[(size_t)(lane / DC) * n + (size_t)j * DC + (lane % DC)] = sYj[lane];
            CHOL3_STAMP(3);
            lds_barrier();
            CHOL3_STAMP(4);
        }
    } else {
        // ---- loader: the image of row j + 1 + R goes to the staging buffer during step j (its lanes take it during step j + 1);
        // loads two rows ahead, unconditional from clamped addresses; diagonal block + right-hand sides of row j + R move on to their ring slot
        double preA[PRE], preB[PRE];
#define CHOL3_ISSUE(pre_, jn_)                                                                                        \
        do {                                                                                                          \
            const int jc_ = min((jn_), re - 1);                                                                       \
            _Pragma("unroll") for (int u = 0; u < PRE; u++) {                                                         \
                const int e = min(lane + u * 64, RWP - 1);                                                            \
                const int q = max(e - RW, 0);                                                                         \
                const double* src = (e < RW) ? band + (size_t)jc_ * RW + e : Y + (size_t)(q / DC) * n + (size_t)jc_ * DC + (q % DC); \
                pre_[u] = *src;                                                                                       \
            }                                                                                                         \
        } while (0)
#define CHOL3_STEP(pre_, j_)                                                                                          \
        do {                                                                                                          \
            const int nb = min(b, re - 1 - (j_));                                                                     \
            CHOL3_STAMPJ((j_), 0);                                                                                           \
            phaseB(jm, nb);                                                                                           \
            CHOL3_STAMPJ((j_), 1);                                                                                           \
            lds_barrier();                                                                                            \
            CHOL3_STAMPJ((j_), 2);                                                                                           \
            if ((j_) + R < re) {                                                                                      \
                const double* img = sStage + (size_t)((j_) & 1) * RWP;                                                \
                if (lane < BB) sDg[(size_t)jm * BB + lane] = img[lane];                                               \
                else if (lane < BB + NR * DC) sYr[(size_t)jm * NR * DC + (lane - BB)] = img[RW + lane - BB];          \
            }                                                                                                         \
            double* nxt = sStage + (size_t)(((j_) + 1) & 1) * RWP;                                                    \
            _Pragma("unroll") for (int u = 0; u < PRE; u++) { const int e = lane + u * 64; if (e < RWP) nxt[e] = pre_[u]; } \
            CHOL3_ISSUE(pre_, (j_) + R + 3);                                                                          \
            CHOL3_STAMPJ((j_), 3);                                                                                           \
            lds_barrier();                                                                                            \
            CHOL3_STAMPJ((j_), 4);                                                                                           \
            jm = (jm + 1 == R) ? 0 : jm + 1;                                                                          \
        } while (0)
        CHOL3_ISSUE(preA, r0 + R + 1);
        CHOL3_ISSUE(preB, r0 + R + 2);
        int jm = jm0;
        for (int j = r0; j < r1; j += 2) {
            CHOL3_STEP(preA, j);
            if (j + 1 < r1) CHOL3_STEP(preB, j + 1);
        }
#undef CHOL3_ISSUE
#undef CHOL3_STEP
    }
    // ---- epilogue of a segment: rows [r1, re) = the separator behind it, reduced by this segment, go back to the band
    if (re > r1) {
        __syncthreads();
        if (is_tr) {
            int rel = rho - (r1 % R); if (rel < 0) rel += R;
            const int i = r1 + rel;
            if (valid && i < re && d <= rel) {
                double* dst = band + ((size_t)i * R + d) * BB;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int c = 0; c < 6; c++) dst[a * 6 + c] = acc[a][c];
            }
        } else {
            const int ot = (role == 0) ? lane : (role - TW) * 64 + lane, on = nt - TW * 64;
            for (int idx = ot; idx < (re - r1) * BB; idx += on) {
                const int s = idx / BB, e = idx - s * BB;
                band[(size_t)(r1 + s) * RW + e] = sDg[(size_t)((r1 + s) % R) * BB + e];
            }
            for (int idx = ot; idx < (re - r1) * NR * DC; idx += on) {
                const int s = idx / (NR * DC), e = idx - s * (NR * DC);
                Y[(size_t)(e / DC) * n + (size_t)(r1 + s) * DC + (e % DC)] = sYr[(size_t)((r1 + s) % R) * NR * DC + e];
            }
        }
    }
    if (sig >= 0) {
        __threadfence(); __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace ssfm
