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
// One node's elimination (one workgroup, ring_elim_node), v with neighbours u_j:
//     A   = D_v - PL[v] - PR[v]          PL / PR: the Schur updates earlier eliminations on v's left / right have left for it (each has ONE writer per step, steps are
//     t   = t_v - tL[v] - tR[v]          launches: no atomics, a fixed order of every sum -- the replicated multi-rank solve relies on that)
//     B_j = M[u_j][v]                    the original coupling E (from Z) and / or the fill an earlier elimination stored in ringE, either possibly transposed
//     [A; B_0; B_1; t^T]  ->  tall right-looking Cholesky on the first Q columns:   L = chol(A),  F_j = B_j L^-T,  w = L^-1 t      (one pass: the B and t rows ride along as panel rows)
//     then, while F_0, F_1, w still sit in LDS, what the neighbours will need:  P{side}[u_j] (+)= F_j F_j^T,  t{side}[u_j] (+)= F_j w,  ringE[v] = -F_1 F_0^T
//     (16x16 tiles of v_mfma_f64_16x16x4 with the operands read from LDS)
// and on the way back (ring_back_node):  x_v = L^-T (w - sum_j F_j^T x_{u_j}).
#pragma once
#include "band_sub.h"
#include "ring_schedule.h"

namespace ssfm {

constexpr int RING_BACK_P = 8;                    // threads that share one entry of F^T x in ring_back_node
inline size_t ring_elim_lds_bytes(int Q, int NR) { return (size_t)(3 * Q + NR) * (size_t)(Q | 1) * sizeof(double); }

// one 16x16 tile of P R^T with P, R = rows of the LDS matrix T (row stride LD, odd: the sixteen rows of an operand read fall into sixteen different banks); K = Q columns.
// lane (li, lk) reads P[r0 + li][k0 + 4 lk .. + 3] and R[c0 + li][...] and holds C[lk + 4 q][li] in acc[q] (the layout of v_mfma_f64_16x16x4, band_sub.h 3b)
__device__ __forceinline__ v4d_t ring_tile_lds(const double* __restrict__ P, const double* __restrict__ R, int r0, int c0, int Q, int LD, int li, int lk) {
    const double* pa = P + (size_t)min(r0 + li, Q - 1) * LD;
    const double* pb = R + (size_t)min(c0 + li, Q - 1) * LD;
    v4d_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < Q; k0 += 16) {
        const int k = k0 + 4 * lk;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = k + u < Q; const int kk = in ? k + u : 0;
            const double av = pa[kk], bv = pb[kk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? av : 0.0, in ? bv : 0.0, acc, 0, 0, 0);
        }
    }
    return acc;
}

// The elimination of one separator by the whole workgroup (blockDim.x threads, a multiple of 256; LDS: ring_elim_lds_bytes).  r = its record (ring_schedule.h).
template <int DC, int NR>
__device__ __forceinline__ void ring_elim_node(const int* __restrict__ r, const double* __restrict__ Z, const double* __restrict__ Dd, const double* __restrict__ tt,
                                               double* __restrict__ crL, double* __restrict__ crF, double* __restrict__ crW, double* __restrict__ crP, double* __restrict__ crT,
                                               double* __restrict__ crE, const int N, const int b, int* __restrict__ fail_flag, double* __restrict__ T,
                                               long long* __restrict__ stamps = nullptr,        // stamps: SSFM_RING_STAMPS timing study (ba_handle.h), null otherwise
                                               const int role = 0, const int nroles = 1) {      // r05ag: nroles workgroups per node (below, phase 3)
    const long long ts0 = stamps ? wall_clock64() : 0;
    const int Q = b * DC, LD = Q | 1, n = N * DC, tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int v = r[0], nn = r[1], hasL = r[2], hasR = r[3];
    const size_t QQ = (size_t)Q * Q;                              // T rows [0, Q): A; [Q + j Q, ..): B_j; [3 Q, 3 Q + NR): t^T
    // ---- phase 0: everything this elimination reads from global memory, in one sweep of independent loads
    {
        const double* D = Dd + (size_t)v * QQ; const double* PL = crP + ((size_t)v * 2) * QQ; const double* PR = PL + QQ;
        for (int e = tid; e < Q * Q; e += nt) {
            const int i = e / Q, c = e - i * Q;
            double val = 0.0;
            if (c <= i) { val = D[e]; if (hasL) val -= PL[e]; if (hasR) val -= PR[e]; }
            T[i * LD + c] = val;
        }
        for (int e = tid; e < NR * Q; e += nt) {
            const int rr = e / Q, c = e - rr * Q;
            double val = tt[(size_t)v * NR * Q + e];
            if (hasL) val -= crT[((size_t)v * 2) * NR * Q + e];
            if (hasR) val -= crT[((size_t)v * 2 + 1) * NR * Q + e];
            T[(3 * Q + rr) * LD + c] = val;
        }
        for (int j = 0; j < nn; j++) {
            const int* q = r + 8 + 16 * j;
            for (int t = 0; t < q[2]; t++) {
                const int* z = q + 4 + 5 * t;
                const bool first = t == 0;
                // consecutive threads walk the contiguous index of the source; the LDS row stride is odd, so the transposed writes do not collide either
                if (z[0] == 0) {
                    const size_t base = (size_t)z[1] * DC;
                    if (z[2]) { for (int e = tid; e < Q * Q; e += nt) { const int i = e / Q, c = e - i * Q; const double val = Z[(size_t)i * n + base + c]; double* d = T + (Q + j * Q + i) * LD + c; *d = first ? val : *d + val; } }
                    else      { for (int e = tid; e < Q * Q; e += nt) { const int c = e / Q, i = e - c * Q; const double val = Z[(size_t)c * n + base + i]; double* d = T + (Q + j * Q + i) * LD + c; *d = first ? val : *d + val; } }
                } else {
                    const double* E = crE + (size_t)z[1] * QQ;
                    if (z[2]) { for (int e = tid; e < Q * Q; e += nt) { const int c = e / Q, i = e - c * Q; const double val = E[e]; double* d = T + (Q + j * Q + i) * LD + c; *d = first ? val : *d + val; } }
                    else      { for (int e = tid; e < Q * Q; e += nt) { const int i = e / Q, c = e - i * Q; const double val = E[e]; double* d = T + (Q + j * Q + i) * LD + c; *d = first ? val : *d + val; } }
                }
                __syncthreads();                                    // (a second term adds to what the first one wrote, through another thread mapping)
            }
        }
    }
    const long long ts1 = stamps ? wall_clock64() : 0;
    // ---- phase 2: tall right-looking Cholesky, NB columns per pass; rows [Q, 3 Q + NR) are panel rows only.
    // (Measured and dropped, profiles/r05_notes.md r05h: a look-ahead wave that factors the next diagonal block during the trailing update -- 33.9 against 34.2 us per
    //  elimination at Q = 42; four rows per thread in the trailing update + all loads of phase 0 issued up front -- 36.7 us.  Phase stamps at Q = 42 / 78: loads 6.9 / 9.5 us,
    //  this phase 12.6 / 41.9 us, stores + the neighbours' products 9.2 / 14.2 us.)
    // Sixteen columns per pass on the matrix cores (r05p; before: DC columns per pass, lane per row, 12.6 / 41.9 us of the 34 / 74 us at Q = 42 / 78): wave 0 factors and
    // inverts the 16x16 diagonal block (wave_ldl_inverse16_mfma, band_sub.h), the panel rows become A G^T and the trailing tiles lose P_I P_J^T, both as 16x16x4 tiles
    // with the operands read from LDS.  Row tiles sit on the 16-grid of the tall matrix (the last pass: from row Q on); rows of an absent neighbour ride along unread.
    constexpr int TB = 16;
    const int li = lane & 15, lk = lane >> 4;
    const int R_all = 3 * Q + NR, TQ = (Q + TB - 1) / TB, RT = (R_all + TB - 1) / TB;
    const int dead0 = Q + nn * Q, dead1 = 3 * Q;                    // [dead0, dead1): B rows of neighbours that do not exist
    auto tile16 = [&](const double* pa, const double* pb, int nk) { // sum_k pa[k] pb[k], k < nk <= 16: pa / pb = this lane's operand rows at the pass's first column
        v4d_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = 4 * lk + u; const bool in = k < nk;
            const double av = pa[in ? k : 0], bv = pb[in ? k : 0];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? av : 0.0, in ? bv : 0.0, acc, 0, 0, 0);
        }
        return acc;
    };
    auto factor_diag = [&](int c0, int nJ) {                        // wave 0: the diagonal block becomes G = L_blk^-1 (lower triangle, in place)
        v4d_t Sd, Gd;
#pragma unroll
        for (int q = 0; q < 4; q++) { const int rr = lk + 4 * q; Sd[q] = (rr < nJ && li < nJ) ? T[(c0 + max(rr, li)) * LD + c0 + min(rr, li)] : ((rr == li) ? 1.0 : 0.0); }
        if (!wave_ldl_inverse16_mfma(Sd, Gd) && lane == 0) *fail_flag = 1;
#pragma unroll
        for (int q = 0; q < 4; q++) { const int rr = lk + 4 * q; if (rr >= li && rr < nJ) T[(c0 + rr) * LD + c0 + li] = Gd[q]; }
    };
    auto trailing_tile = [&](int c0, int nJ, int I, int Jc) {
        const int r0 = TB * I;
        if (r0 >= dead0 && r0 + TB <= dead1) return;
        const v4d_t acc = tile16(T + (size_t)min(r0 + li, R_all - 1) * LD + c0, T + (size_t)min(TB * Jc + li, Q - 1) * LD + c0, nJ);
        const int col = TB * Jc + li;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = r0 + lk + 4 * q;
            if (row < R_all && col < Q && (col <= row || row >= Q)) T[(size_t)row * LD + col] -= acc[q];     // triangle rows: up to the diagonal (the upper part stays zero: G^T above)
        }
    };
    __syncthreads();
    if (wave == 0) factor_diag(0, min(TB, Q));
    for (int J = 0; J < TQ; J++) {
        const int c0 = TB * J, nJ = min(TB, Q - c0);
        __syncthreads();                                            // G_J is there (wave 0: above, or during the previous pass's trailing update)
        const int r_start = (J + 1 < TQ) ? c0 + TB : Q;            // first panel row
        const int npt = (R_all - r_start + TB - 1) / TB;
        for (int t = wave; t < npt; t += nw) {                      // panel: L(rows, block J) = A'(rows, block J) G^T, in place (a wave reads its whole tile before it stores)
            const int r0 = r_start + TB * t;
            if (r0 >= dead0 && r0 + TB <= dead1) continue;
            const v4d_t acc = tile16(T + (size_t)min(r0 + li, R_all - 1) * LD + c0, T + (size_t)(c0 + min(li, nJ - 1)) * LD + c0, nJ);
#pragma unroll
            for (int q = 0; q < 4; q++) { const int row = r0 + lk + 4 * q; if (row < R_all && li < nJ) T[(size_t)row * LD + c0 + li] = acc[q]; }
        }
        __syncthreads();
        if (J + 1 < TQ) {
            // trailing update: tiles (I, Jc), Jc > J, I >= Jc down to the last row tile.  Look-ahead: wave 0 takes the next diagonal tile first and factors it at once --
            // nothing else touches that tile in this pass, the others only read block column J -- so the 16 dependent steps of the next block leave the critical path
            if (wave == 0) { trailing_tile(c0, nJ, J + 1, J + 1); factor_diag(c0 + TB, min(TB, Q - c0 - TB)); }
            else {
                const int ncol = TQ - 1 - J;
                int ntask = -1; for (int jc = 0; jc < ncol; jc++) ntask += RT - (J + 1 + jc);          // without the diagonal tile (task 0 of column J + 1)
                for (int task = wave - 1; task < ntask; task += nw - 1) {
                    int Jc = J + 1, t2 = task + 1;
                    while (t2 >= RT - Jc) { t2 -= RT - Jc; Jc++; }
                    trailing_tile(c0, nJ, Jc + t2, Jc);
                }
            }
        }
    }
    __syncthreads();
    const long long ts2 = stamps ? wall_clock64() : 0;
    // ---- phase 3: L (diagonal blocks hold G = L_blk^-1), F_j, w for the way back; the neighbours' Schur updates and the fill between them, from the F rows in LDS
    // Several workgroups per node (k_ring_cr_elim: gridDim.y) all run phases 0-2 -- the same loads and the same arithmetic, hence the same F rows in every LDS -- and
    // share this phase: workgroup 0 stores the factor, the others' tiles of the neighbours' products are dealt round robin.  Every output still has ONE writer.  (The
    // products are half of an elimination's flops and sat on one compute unit while 250 idled: 13.9 of 55 us at Q = 78.)
    if (role == 0) {
    for (int e = tid; e < Q * Q; e += nt) {
        const int i = e / Q, c = e - i * Q;
        crL[(size_t)v * QQ + e] = (c <= i) ? T[i * LD + c] : 0.0;
        for (int j = 0; j < nn; j++) crF[((size_t)v * 2 + j) * QQ + e] = T[(Q + j * Q + i) * LD + c];
    }
    for (int e = tid; e < NR * Q; e += nt) { const int rr = e / Q, c = e - rr * Q; crW[(size_t)v * NR * Q + e] = T[(3 * Q + rr) * LD + c]; }
    }
    if (nn > 0) {
        const int li = lane & 15, lk = lane >> 4, TQ = (Q + 15) / 16, ntl = TQ * (TQ + 1) / 2;
        const int ntask = nn * ntl + (r[6] ? TQ * TQ : 0);
        for (int task = wave + role * nw; task < ntask; task += nroles * nw) {
            if (task < nn * ntl) {                                 // P{side}[u_j] (+)= F_j F_j^T, lower tiles
                const int j = task / ntl, t2 = task - j * ntl;
                int I = 0; while ((I + 1) * (I + 2) / 2 <= t2) I++;
                const int J = t2 - I * (I + 1) / 2;
                const double* F = T + (size_t)(Q + j * Q) * LD;
                const v4d_t acc = ring_tile_lds(F, F, 16 * I, 16 * J, Q, LD, li, lk);
                const int* q = r + 8 + 16 * j;
                double* P = crP + ((size_t)q[0] * 2 + q[3]) * QQ;
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const int row = 16 * I + lk + 4 * qq, col = 16 * J + li;
                    if (row < Q && col <= row) { double* d = P + (size_t)row * Q + col; *d = q[14] ? *d + acc[qq] : acc[qq]; }
                }
            } else {                                               // ringE[v] = -F_1 F_0^T  (rows: neighbour 1, columns: neighbour 0)
                const int u = task - nn * ntl, I = u / TQ, J = u - I * TQ;
                const v4d_t acc = ring_tile_lds(T + (size_t)(2 * Q) * LD, T + (size_t)Q * LD, 16 * I, 16 * J, Q, LD, li, lk);
                double* E = crE + (size_t)v * QQ;
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const int row = 16 * I + lk + 4 * qq, col = 16 * J + li;
                    if (row < Q && col < Q) E[(size_t)row * Q + col] = -acc[qq];
                }
            }
        }
        if (role == 0)
        for (int e = tid; e < nn * NR * Q; e += nt) {             // t{side}[u_j] (+)= F_j w
            const int j = e / (NR * Q), e2 = e - j * NR * Q, rr = e2 / Q, i = e2 - rr * Q;
            const double* Fr = T + (size_t)(Q + j * Q + i) * LD; const double* w = T + (size_t)(3 * Q + rr) * LD;
            double a = 0.0;
            for (int k = 0; k < Q; k++) a += Fr[k] * w[k];
            const int* q = r + 8 + 16 * j;
            double* d = crT + ((size_t)q[0] * 2 + q[3]) * NR * Q + e2;
            *d = q[14] ? *d + a : a;
        }
    }
    if (stamps && tid == 0 && role == 0) { long long* d = stamps + 4 * (size_t)v; d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = wall_clock64(); }
}

// ---- back substitution of one separator by the whole workgroup: x_v = L^-T (w - sum_j F_j^T x_{u_j}) -> Y rows of v (and of its copy slot); LDS: ring_back_lds_bytes
template <int DC, int NR>
__device__ __forceinline__ void ring_back_node(const int* __restrict__ r, const double* __restrict__ crL, const double* __restrict__ crF, const double* __restrict__ crW,
                                               double* __restrict__ Y, const int N, const int b, double* __restrict__ lds) {
    const int Q = b * DC, LD = Q | 1, n = N * DC, tid = threadIdx.x, nt = blockDim.x;
    const int v = r[0], nn = r[1], copy = r[4], p0 = r[5];
    const size_t QQ = (size_t)Q * Q;
    double* sL = lds;                      // [Q][LD]
    double* sv = sL + (size_t)Q * LD;      // [NR][Q]
    double* sx = sv + NR * Q;              // [2][NR][Q]  neighbours' solutions
    double* so = sx + 2 * NR * Q;          // [NR][Q]     x_v
    for (int j = 0; j < nn; j++) {
        const int pj = r[8 + 16 * j + 1];
        for (int e = tid; e < NR * Q; e += nt) { const int rr = e / Q, i = e - rr * Q; sx[(j * NR + rr) * Q + i] = Y[(size_t)rr * n + (size_t)pj * DC + i]; }
    }
    for (int e = tid; e < Q * Q; e += nt) { const int i = e / Q, c = e - i * Q; sL[i * LD + c] = crL[(size_t)v * QQ + e]; }
    __syncthreads();
    // F_j^T x_{u_j}: every entry (rr, k) is a sum over the Q rows of F, read from global memory column-wise (coalesced over k).  One thread per entry walked Q dependent-free
    // but sequential loads per neighbour (20 of the 26 us of a back substitution at Q = 78); now RING_BACK_P threads share an entry -- each a slice of the rows, the
    // slices in LDS, added in slice order (r05ad)
    {
        const int items = NR * Q;
        int P = 1; if (Q >= 40) while (2 * P <= RING_BACK_P && 2 * P * items <= nt) P *= 2;      // (small blocks: the extra barrier costs more than the shorter walk saves -- 4000-node pose graph, Q = 24: 19.0 -> 20.3 ms)
        double* spart = so + NR * Q;           // [RING_BACK_P][NR * Q]
        const int rows_per = (Q + P - 1) / P;
        for (int t = tid; t < P * items; t += nt) {
            const int part = t / items, e = t - part * items, rr = e / Q, k = e - rr * Q;
            const int i0 = part * rows_per, i1 = min(Q, i0 + rows_per);
            double acc = 0.0;
            for (int j = 0; j < nn; j++) {
                const double* F = crF + ((size_t)v * 2 + j) * QQ;
                const double* xj = sx + (j * NR + rr) * Q;
#pragma unroll 8
                for (int i = i0; i < i1; i++) acc += F[(size_t)i * Q + k] * xj[i];
            }
            spart[part * items + e] = acc;
        }
        __syncthreads();
        for (int e = tid; e < items; e += nt) {
            double acc = 0.0;
            for (int part = 0; part < P; part++) acc += spart[part * items + e];
            sv[e] = crW[(size_t)v * NR * Q + e] - acc;
        }
    }
    __syncthreads();
    for (int J = (Q + 15) / 16 - 1; J >= 0; J--) {                  // blocks of sixteen columns, the last one shorter (ring_elim_node)
        const int c0 = 16 * J, nJ = min(16, Q - c0);
        if (tid < nJ * NR) {
            const int rr = tid / nJ, k = tid - rr * nJ;
            double acc = 0.0;
            for (int m = k; m < nJ; m++) acc += sL[(c0 + m) * LD + c0 + k] * sv[rr * Q + c0 + m];     // (G^T v_blk)_k
            so[rr * Q + c0 + k] = acc;
        }
        __syncthreads();
        for (int e = tid; e < NR * c0; e += nt) {
            const int rr = e / c0, i = e - rr * c0;
            double acc = 0.0;
            for (int k = 0; k < nJ; k++) acc += sL[(c0 + k) * LD + i] * so[rr * Q + c0 + k];
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
inline size_t ring_back_lds_bytes(int Q, int NR) { return ((size_t)Q * (Q | 1) + (size_t)(4 + RING_BACK_P) * NR * Q) * sizeof(double); }

// one step of the reduction: the eliminations of the step, one workgroup each
template <int DC, int NR>
__global__ void __launch_bounds__(1024)
k_ring_cr_elim(const int* __restrict__ rec, int rec0, const double* __restrict__ Z, const double* __restrict__ Dd, const double* __restrict__ tt,
               double* __restrict__ crL, double* __restrict__ crF, double* __restrict__ crW, double* __restrict__ crP, double* __restrict__ crT, double* __restrict__ crE,
               int N, int b, int* __restrict__ fail_flag, long long* __restrict__ stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    ring_elim_node<DC, NR>(rec + (size_t)(rec0 + blockIdx.x) * RING_REC, Z, Dd, tt, crL, crF, crW, crP, crT, crE, N, b, fail_flag, lds, stamps, (int)blockIdx.y, (int)gridDim.y);
}
template <int DC, int NR>
__global__ void __launch_bounds__(1024)
k_ring_cr_back(const int* __restrict__ rec, int rec0, const double* __restrict__ crL, const double* __restrict__ crF, const double* __restrict__ crW,
               double* __restrict__ Y, int N, int b) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    ring_back_node<DC, NR>(rec + (size_t)(rec0 + blockIdx.x) * RING_REC, crL, crF, crW, Y, N, b, lds);
}
// The last few separators of every ring (from RING_TAIL_P active ones on): their eliminations, the root and their back substitutions by ONE workgroup per ring in ONE
// launch, one after the other -- at that point a step holds one or two eliminations per ring and a launch per step would cost more than the work.  tail_ptr: records of
// ring g = [tail_ptr[g], tail_ptr[g + 1]) in elimination order.  Blocks written by an earlier elimination of the same workgroup are read back through global memory:
// a WORKGROUP-scope fence + barrier between two nodes (the waves of a workgroup share one L1, which writes through: an agent-scope __threadfence would
// write the L2 back for other XCDs that do not take part, r05ab).
template <int DC, int NR>
__global__ void __launch_bounds__(1024)
k_ring_cr_tail(const int* __restrict__ rec, const int* __restrict__ tail_ptr, const double* __restrict__ Z, const double* __restrict__ Dd, const double* __restrict__ tt,
               double* __restrict__ crL, double* __restrict__ crF, double* __restrict__ crW, double* __restrict__ crP, double* __restrict__ crT, double* __restrict__ crE,
               double* __restrict__ Y, int N, int b, int* __restrict__ fail_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int q0 = tail_ptr[blockIdx.x], q1 = tail_ptr[blockIdx.x + 1];
    for (int q = q0; q < q1; q++) {
        ring_elim_node<DC, NR>(rec + (size_t)q * RING_REC, Z, Dd, tt, crL, crF, crW, crP, crT, crE, N, b, fail_flag, lds);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads();
    }
    for (int q = q1 - 1; q >= q0; q--) {
        ring_back_node<DC, NR>(rec + (size_t)q * RING_REC, crL, crF, crW, Y, N, b, lds);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads();
    }
}

}  // namespace ssfm
