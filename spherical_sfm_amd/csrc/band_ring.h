// spherical_sfm_amd -- long camera rings in their own circular order: the separator CYCLE solved by cyclic reduction (round 5).
// (Part of the replacement of Ceres' sparse Cholesky on the reduced camera system, SPARSE_SCHUR, src/sfm.cpp:276-279: still an exact direct solve.)
//
// ba_flatten.h (band_plan, RingComp) lays a ring of cameras out as   [copy of S_{m-1}] | A_0 | S_0 | A_1 | S_1 | ... | A_{m-1} | S_{m-1}   with a band of half-width
// b = the ring's reach (half of what the Cuthill-McKee fold needs).  Steps 1-3 and 5 of band_sub.h run on it unchanged -- every arc A_k is a segment with the
// separator S_k behind it (window continuation) and S_{k-1} in front of it (spike; for A_0 the copy slot of S_{m-1}) -- and leave, per separator s,
//     D_s  (Q x Q, Q = b DC)   its diagonal block after the elimination of both neighbouring arcs        (k_sub_sep_assemble_mfma -> Dd)
//     t_s                      its right-hand sides                                                     (-> tt)
//     E_s  (Q x Q)             the coupling (rows S_s, columns S_{s-1}) through the arc A_s: E_s(i, c) = Z[c][first scalar row of S_s + i]   (k_sub_spike_fwd)
// i.e. a block-tridiagonal system that closes on itself: M[s][s] = D_s, M[s][s-1] = E_s, indices mod m.
//
// Cyclic reduction: of the p separators still active (a cycle) every second one is eliminated at once -- each has two active neighbours u, w that are NOT eliminated
// in the same step --, which couples u and w directly (fill) and leaves a cycle of ceil(p / 2); two nodes are eliminated one after the other, the last one is the
// root.  Depth ceil(log2 m) + 2 instead of m / 2 + 2 for a chain taken from both ends, and every block is half the size of the folded chain's (2.5th power: 5.6x).
// One node's elimination (one workgroup, k_ring_cr_elim), v with neighbours u_j:
//     A   = D_v - sum over earlier eliminated neighbours x of F_{x,v} F_{x,v}^T          (the pending Schur updates of v, GATHERED: no atomics anywhere)
//     t   = t_v - sum F_{x,v} w_x
//     B_j = M[u_j][v]    = the original coupling E (from Z) and / or the fill -F_{x,u_j} F_{x,v}^T through an earlier eliminated x
//     [A; B_0; B_1; t^T]  ->  tall right-looking Cholesky on the first Q columns:   L = chol(A),  F_j = B_j L^-T,  w = L^-1 t      (one pass: the B and t rows ride along as panel rows)
// and on the way back (k_ring_cr_back):  x_v = L^-T (w - sum_j F_j^T x_{u_j}).
// Every sum has a fixed order: the result does not depend on the schedule (the replicated multi-rank solve relies on that).
// The products of two stored F blocks run on the matrix cores with the operands straight from global memory (v_mfma_f64_16x16x4, the layout of k_sub_sep_assemble_mfma).
#pragma once
#include "band_sub.h"
#include "ring_schedule.h"

namespace ssfm {

inline size_t ring_elim_lds_bytes(int Q, int NR) { return (size_t)(3 * Q + NR) * (size_t)(Q | 1) * sizeof(double); }

// ---- elimination of one separator --------------------------------------------------------------------------------------------------------
// one 16x16 tile of P R^T (P, R: [Q][Q] row-major in global memory), added to acc: lane (li, lk) loads P[r0 + li][k0 + 4 lk ..+3] and R[c0 + li][...] -- 32 contiguous
// bytes per operand and trip -- and holds C[lk + 4 q][li] in acc[q] (the accumulator layout of v_mfma_f64_16x16x4, band_sub.h 3b)
__device__ __forceinline__ v4d_t ring_tile_abt(const double* __restrict__ P, const double* __restrict__ R, int r0, int c0, int Q, int li, int lk, v4d_t acc) {
    const double* pa = P + (size_t)min(r0 + li, Q - 1) * Q;
    const double* pb = R + (size_t)min(c0 + li, Q - 1) * Q;
    constexpr int DEPTH = 3;                                       // trips of 16 k in flight together
    for (int k0 = 0; k0 < Q; k0 += 16 * DEPTH) {
        double a[DEPTH][4], bb[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int k = k0 + 16 * d + 4 * lk;
#pragma unroll
            for (int u = 0; u < 4; u++) { const bool in = k + u < Q; const int kk = in ? k + u : 0; const double av = pa[kk], bv = pb[kk]; a[d][u] = in ? av : 0.0; bb[d][u] = in ? bv : 0.0; }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][u], bb[d][u], acc, 0, 0, 0);
    }
    return acc;
}

template <int DC, int NR>
__global__ void __launch_bounds__(1024)
k_ring_cr_elim(const int* __restrict__ rec, const int* __restrict__ pend, int rec0, const double* __restrict__ Z, const double* __restrict__ Dd,
               const double* __restrict__ tt, double* __restrict__ crL, double* __restrict__ crF, double* __restrict__ crW, int N, int b, int* __restrict__ fail_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NB = DC;
    const int Q = b * DC, LD = Q | 1, n = N * DC, tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int* r = rec + (size_t)(rec0 + blockIdx.x) * RING_REC;
    const int v = r[0], nn = r[1], pl = r[2], ph = r[3];
    double* T = lds;                                                // rows [0, Q): A; [Q + j Q, ..): B_j; [3 Q, 3 Q + NR): t^T
    const size_t QQ = (size_t)Q * Q;
    // ---- phase 0: D_v, the couplings that sit in Z, t_v
    for (int e = tid; e < Q * Q; e += nt) {
        const int i = e / Q, c = e - i * Q;
        T[i * LD + c] = (c <= i) ? Dd[((size_t)v * Q + i) * Q + c] : 0.0;
        for (int j = 0; j < nn; j++) {
            const int* q = r + 8 + 16 * j;
            double val = 0.0;
            for (int t = 0; t < q[2]; t++) {
                const int* z = q + 4 + 6 * t;
                if (z[0] == 0) val += z[2] ? Z[(size_t)i * n + (size_t)z[1] * DC + c] : Z[(size_t)c * n + (size_t)z[1] * DC + i];
            }
            T[(Q + j * Q + i) * LD + c] = val;
        }
    }
    for (int e = tid; e < NR * Q; e += nt) { const int rr = e / Q, c = e - rr * Q; T[(3 * Q + rr) * LD + c] = tt[(size_t)v * NR * Q + e]; }
    __syncthreads();
    // ---- phase 1a: t -= F_{x,v} w_x over the pending updates, one wave per row, in list order
    for (int i = wave; i < Q && ph > pl; i += nw) {
        double acc[NR];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) acc[rr] = 0.0;
        for (int p = pl; p < ph; p++) {
            const int x = pend[2 * p], sl = pend[2 * p + 1];
            const double* F = crF + ((size_t)x * 2 + sl) * QQ + (size_t)i * Q;
            const double* w = crW + (size_t)x * NR * Q;
            double part[NR];
#pragma unroll
            for (int rr = 0; rr < NR; rr++) part[rr] = 0.0;
            for (int k = lane; k < Q; k += 64) { const double f = F[k];
#pragma unroll
                for (int rr = 0; rr < NR; rr++) part[rr] += f * w[rr * Q + k]; }
#pragma unroll
            for (int rr = 0; rr < NR; rr++) acc[rr] += wave_sum(part[rr]);
        }
        if (lane == 0) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) T[(3 * Q + rr) * LD + i] -= acc[rr];
        }
    }
    // ---- phase 1b: products of stored F blocks on the matrix cores; every output tile belongs to one wave, which sums its terms in list order
    {
        const int li = lane & 15, lk = lane >> 4, TQ = (Q + 15) / 16, ntl = TQ * (TQ + 1) / 2;
        const int ntask = ntl + nn * TQ * TQ;
        for (int task = wave; task < ntask; task += nw) {
            v4d_t acc = {0.0, 0.0, 0.0, 0.0};
            int I = 0, J = 0, rowbase = 0; bool lower = false, any = false;
            if (task < ntl) {
                while ((I + 1) * (I + 2) / 2 <= task) I++;
                J = task - I * (I + 1) / 2; lower = true;
                for (int p = pl; p < ph; p++) { const double* F = crF + ((size_t)pend[2 * p] * 2 + pend[2 * p + 1]) * QQ; acc = ring_tile_abt(F, F, 16 * I, 16 * J, Q, li, lk, acc); any = true; }
            } else {
                const int t2 = task - ntl, j = t2 / (TQ * TQ), u = t2 - j * TQ * TQ;
                I = u / TQ; J = u - I * TQ; rowbase = Q + j * Q;
                const int* q = r + 8 + 16 * j;
                for (int t = 0; t < q[2]; t++) {
                    const int* z = q + 4 + 6 * t;
                    if (z[0] == 1) { acc = ring_tile_abt(crF + ((size_t)z[1] * 2 + z[2]) * QQ, crF + ((size_t)z[1] * 2 + z[3]) * QQ, 16 * I, 16 * J, Q, li, lk, acc); any = true; }
                }
            }
            if (any) {
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const int row = 16 * I + lk + 4 * qq, col = 16 * J + li;
                    if (row < Q && col < Q && (!lower || col <= row)) T[(rowbase + row) * LD + col] -= acc[qq];
                }
            }
        }
    }
    // ---- phase 2: tall right-looking Cholesky, NB columns per pass; rows [Q, 3 Q + NR) are panel rows only
    const int tx = tid & 255, ty = tid >> 8, nty = nt >> 8;
    const int R_all = 3 * Q + NR;
    const bool row_ok = tx < Q || (tx < 3 * Q && (tx - Q) / Q < nn) || (tx >= 3 * Q && tx < R_all);
    for (int c0 = 0; c0 < Q; c0 += NB) {
        __syncthreads();
        if (wave == 0) {
            double row[NB], g[NB];
#pragma unroll
            for (int c = 0; c < NB; c++) row[c] = (lane < NB) ? T[(c0 + max(lane, c)) * LD + c0 + min(lane, c)] : ((lane == c) ? 1.0 : 0.0);
            if (!wave_chol_inverse<NB>(row, g) && lane == 0) *fail_flag = 1;
            if (lane < NB) {
#pragma unroll
                for (int rr = 0; rr < NB; rr++) if (rr >= lane) T[(c0 + rr) * LD + c0 + lane] = g[rr];
            }
        }
        __syncthreads();
        if (ty == 0 && row_ok && tx >= c0 + NB) {                  // panel: L(i, c0..) = A'(i, c0..) G^T
            double* Pr = T + tx * LD + c0;
            double ev[NB], pv[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) ev[k] = Pr[k];
#pragma unroll
            for (int k = 0; k < NB; k++) { double a = 0.0;
#pragma unroll
                for (int m = 0; m <= k; m++) a += ev[m] * T[(c0 + k) * LD + c0 + m];
                pv[k] = a; }
#pragma unroll
            for (int k = 0; k < NB; k++) Pr[k] = pv[k];
        }
        __syncthreads();
        if (row_ok && tx >= c0 + NB) {
            const double* Pr = T + tx * LD + c0;
            double pv[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) pv[k] = Pr[k];
            const int cend = (tx < Q) ? tx : Q - 1;                // triangle rows: columns up to the diagonal; B and t rows: all columns
            for (int cp = c0 + NB + ty; cp <= cend; cp += nty) {
                const double* Lr = T + cp * LD + c0;
                double val = T[tx * LD + cp];
#pragma unroll
                for (int k = 0; k < NB; k++) val -= pv[k] * Lr[k];
                T[tx * LD + cp] = val;
            }
        }
    }
    __syncthreads();
    // ---- phase 3: L (diagonal blocks hold G = L_blk^-1), F_j, w
    for (int e = tid; e < Q * Q; e += nt) {
        const int i = e / Q, c = e - i * Q;
        crL[(size_t)v * QQ + e] = (c <= i) ? T[i * LD + c] : 0.0;
        for (int j = 0; j < nn; j++) crF[((size_t)v * 2 + j) * QQ + e] = T[(Q + j * Q + i) * LD + c];
    }
    for (int e = tid; e < NR * Q; e += nt) { const int rr = e / Q, c = e - rr * Q; crW[(size_t)v * NR * Q + e] = T[(3 * Q + rr) * LD + c]; }
}

// ---- back substitution of one separator: x_v = L^-T (w - sum_j F_j^T x_{u_j}) -> Y rows of v (and of its copy slot) ---------------------------------
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_ring_cr_back(const int* __restrict__ rec, int rec0, const double* __restrict__ crL, const double* __restrict__ crF, const double* __restrict__ crW,
               double* __restrict__ Y, int N, int b) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NB = DC;
    const int Q = b * DC, LD = Q | 1, n = N * DC, tid = threadIdx.x, nt = blockDim.x;
    const int* r = rec + (size_t)(rec0 + blockIdx.x) * RING_REC;
    const int v = r[0], nn = r[1], copy = r[4], p0 = r[5];
    const size_t QQ = (size_t)Q * Q;
    double* sL = lds;                      // [Q][LD]
    double* sv = sL + (size_t)Q * LD;      // [NR][Q]
    double* sx = sv + NR * Q;              // [2][NR][Q]  neighbours' solutions
    double* so = sx + 2 * NR * Q;          // [NR][Q]     x_v
    for (int e = tid; e < Q * Q; e += nt) { const int i = e / Q, c = e - i * Q; sL[i * LD + c] = crL[(size_t)v * QQ + e]; }
    for (int e = tid; e < NR * Q; e += nt) sv[e] = crW[(size_t)v * NR * Q + e];
    for (int j = 0; j < nn; j++) {
        const int pj = r[8 + 16 * j + 1];
        for (int e = tid; e < NR * Q; e += nt) { const int rr = e / Q, i = e - rr * Q; sx[(j * NR + rr) * Q + i] = Y[(size_t)rr * n + (size_t)pj * DC + i]; }
    }
    __syncthreads();
    for (int e = tid; e < NR * Q; e += nt) {
        const int rr = e / Q, k = e - rr * Q;
        double acc = 0.0;
        for (int j = 0; j < nn; j++) {
            const double* F = crF + ((size_t)v * 2 + j) * QQ;
            const double* xj = sx + (j * NR + rr) * Q;
            for (int i = 0; i < Q; i++) acc += F[(size_t)i * Q + k] * xj[i];
        }
        sv[e] -= acc;
    }
    __syncthreads();
    for (int c0 = Q - NB; c0 >= 0; c0 -= NB) {
        if (tid < NB * NR) {
            const int rr = tid / NB, k = tid - rr * NB;
            double acc = 0.0;
            for (int m = k; m < NB; m++) acc += sL[(c0 + m) * LD + c0 + k] * sv[rr * Q + c0 + m];     // (G^T v_blk)_k
            so[rr * Q + c0 + k] = acc;
        }
        __syncthreads();
        for (int e = tid; e < NR * c0; e += nt) {
            const int rr = e / c0, i = e - rr * c0;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NB; k++) acc += sL[(c0 + k) * LD + i] * so[rr * Q + c0 + k];
            sv[rr * Q + i] -= acc;
        }
        __syncthreads();
    }
    for (int e = tid; e < NR * Q; e += nt) {
        const int rr = e / Q, i = e - rr * Q;
        Y[(size_t)rr * n + (size_t)p0 * DC + i] = so[e];
        if (copy >= 0) Y[(size_t)rr * n + (size_t)copy * DC + i] = so[e];
    }
}
inline size_t ring_back_lds_bytes(int Q, int NR) { return ((size_t)Q * (Q | 1) + (size_t)4 * NR * Q) * sizeof(double); }

}  // namespace ssfm
