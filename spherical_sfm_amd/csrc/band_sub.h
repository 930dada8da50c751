// spherical_sfm_amd -- substructured block-band Cholesky: the factorisation of ONE long connected component spread over
// many workgroups.  (Replaces Ceres' sparse Cholesky on the reduced camera system, SPARSE_SCHUR, src/sfm.cpp:276-279.)
//
// A component in Cuthill-McKee order is a block band of half-width b.  b consecutive block rows cut it in two, so the order
//     seg_0 | sep_0 | seg_1 | sep_1 | ... | seg_{P-1}          (every sep = b block rows)
// makes the segments independent of each other; eliminating all segments first and the separators last is an exact
// factorisation in a different (nested-dissection-like) elimination order:
//   1. k_band_chol_v2 on every segment (band_kernels2.h), window continued into the separator BEHIND the segment: that
//      separator receives its factor rows L(sep, seg), its Schur update and its share of the forward substitution for free.
//   2. k_sub_spike_fwd: the coupling to the separator IN FRONT of a segment fills in along the whole segment ("spike"):
//      Z = L_seg^-1 C_left, b*DC right-hand sides, one wave each (the forward twin of k_band_back_v2), continued into the
//      separator behind, where it leaves the coupling block E between the two separators.
//   3. k_sub_sep_assemble: D_j = (reduced inner blocks of sep_j) - Z^T Z,  t_j = y(sep_j) - Z^T y(seg_{j+1}).
//   4. k_sub_sep_chain: block-tridiagonal chain over the separators of a component (dense b*DC blocks, LDS resident).
//   5. k_sub_apply_left: y(seg) -= Z x(sep in front); then k_band_back_v2 continued FROM the separator behind (given rows).
// scripts/lab/substructure_proto.py is the numpy statement of the same algebra.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "band_kernels2.h"

namespace ssfm {

// ---- host: segment / separator tables ----------------------------------------------------------------------------------
struct BandSub {
    bool enabled = false;
    int nseg = 0, nsep = 0, nchain = 0, nleft = 0;
    std::vector<int> seg_lo, seg_hi, seg_wend;      // pivots [lo, hi), window end (hi, or hi + b with a separator behind)
    std::vector<int> left_segs;                     // segments with a separator in front
    std::vector<int> sep_lo, sep_rseg;              // first row of the separator; the segment behind it
    std::vector<int> chain_ptr;                     // separators of chain c: [chain_ptr[c], chain_ptr[c+1])  (one chain per cut component)
};

// Rough cost model in microseconds (MI355X measurements of the kernels involved): a factorisation step, a separator step of the
// chain, and the fixed cost of the extra launches.  Segments must be at least b + 1 rows.
constexpr int SUB_MIN_ROWS = 256;                  // shorter components stay on one workgroup (measured break-even, profiles/r01_notes.md)
inline int sub_choose_segments(int rows, int b, int dc) {
    if (rows < SUB_MIN_ROWS) return 1;
    const double t_step = 0.55 + 0.085 * b, t_sep = 40.0 * (b * dc / 114.0) * (b * dc / 114.0) + 6.0, t_fixed = 30.0;
    int best = 1; double best_t = rows * t_step * 1.3;                      // factorisation + back substitution of the plain path
    for (int P = 2; P <= 64; P++) {
        const int m = (rows - (P - 1) * b) / P;
        if (m < b + 1) break;
        const double t = (m + b) * t_step * 1.5 + (P - 1) * t_sep + t_fixed;
        if (t < best_t) { best_t = t; best = P; }
    }
    return best;
}

inline void sub_build(const std::vector<int>& comp_ptr, int b, int dc, BandSub& S) {
    S = BandSub();
    const char* env = std::getenv("SSFM_BAND_SEGMENTS");               // 1 = never cut; P >= 2 = cut every component that can take it into P
    const int forced = env ? std::atoi(env) : 0;
    if (b < 1 || b * dc > 114 || forced == 1) return;                  // separator blocks must fit the chain kernel's LDS (see k_sub_sep_chain)
    S.chain_ptr.assign(1, 0);
    for (size_t c = 0; c + 1 < comp_ptr.size(); c++) {
        const int c0 = comp_ptr[c], rows = comp_ptr[c + 1] - c0;
        int P = forced >= 2 ? forced : sub_choose_segments(rows, b, dc);
        while (P > 1 && (rows - (P - 1) * b) / P < b + 1) P--;
        const int m_total = rows - (P - 1) * b;
        int pos = c0;
        for (int i = 0; i < P; i++) {
            const int m = m_total / P + (i < m_total % P ? 1 : 0);
            if (i > 0) S.left_segs.push_back((int)S.seg_lo.size());
            S.seg_lo.push_back(pos); S.seg_hi.push_back(pos + m); S.seg_wend.push_back(i + 1 < P ? pos + m + b : pos + m);
            pos += m;
            if (i + 1 < P) { S.sep_lo.push_back(pos); S.sep_rseg.push_back((int)S.seg_lo.size()); pos += b; }
        }
        if (P > 1) { S.enabled = true; S.chain_ptr.push_back((int)S.sep_lo.size()); }
    }
    S.nseg = (int)S.seg_lo.size(); S.nsep = (int)S.sep_lo.size(); S.nchain = (int)S.chain_ptr.size() - 1; S.nleft = (int)S.left_segs.size();
    if (!S.enabled) S = BandSub();
}

// ---- 2. spike: Z(:, q) = L_seg^-1 C_left(:, q), one wave per (segment, column) ---------------------------------------------
// Column q = (separator row r0 - b + q / DC, component q % DC).  C_left lives in the band rows of the segment's first b rows
// (blocks whose column lies in front of r0).  Right-looking: task t = (d-1)*DC + a (d = 1..b) owns the pending sum of row
// k + (d-1), component a; lanes carry tasks t = lane and t = lane + 64.  Rows [r1, re) take no pivot: they receive -sum = E.
// Z: [b*DC][N*DC], row index = global scalar row.
constexpr int SPIKE_PD = 4;
template <int DC>
__global__ void __launch_bounds__(64)
k_sub_spike_fwd(const double* __restrict__ band, const double* __restrict__ Ginv, double* __restrict__ Z, const int* __restrict__ seg_lo,
                const int* __restrict__ seg_hi, const int* __restrict__ seg_wend, const int* __restrict__ left_segs, int N, int b) {
    constexpr int BB = DC * DC;
    const int W = b + 1, n = N * DC, lane = threadIdx.x;
    const int seg = left_segs[blockIdx.x], q = blockIdx.y;
    const int r0 = seg_lo[seg], r1 = seg_hi[seg], re = seg_wend[seg];
    const int cs = q / DC, cc = q - cs * DC;
    double* z = Z + (size_t)q * n;
    const int T = b * DC;
    const int t0 = min(lane, T - 1), t1 = min(lane + 64, T - 1);
    const int d0 = t0 / DC + 1, a0 = t0 - (d0 - 1) * DC, d1 = t1 / DC + 1, a1 = t1 - (d1 - 1) * DC;
    const bool has0 = lane < T, has1 = lane + 64 < T;
    const int lc = min(lane, DC - 1);
    struct Stage { double row0[DC], row1[DC], g[DC], cv; };
    Stage st[SPIKE_PD];
    auto fetch = [&](int k, Stage& s) {    // row a of L(k+d, k) for this lane's tasks; row `lane` of G_k; C_left(k, q)[lane]
        const int kc = min(k, re - 1);
        const int k0 = min(kc + d0, re - 1), k1 = min(kc + d1, re - 1);
        const int dl = min(kc - r0 + b - cs, b);
#pragma unroll
        for (int m = 0; m < DC; m++) {
            s.row0[m] = band[((size_t)k0 * W + d0) * BB + a0 * DC + m];
            s.row1[m] = band[((size_t)k1 * W + d1) * BB + a1 * DC + m];
            s.g[m] = Ginv[(size_t)kc * BB + lc * DC + m];                 // G[lane][m], zero for m > lane
        }
        s.cv = band[((size_t)kc * W + dl) * BB + lc * DC + cc];
    };
#pragma unroll
    for (int u = 0; u < SPIKE_PD; u++) fetch(r0 + u, st[u]);
    double acc0 = 0.0, acc1 = 0.0;
    for (int kb = r0; kb < re; kb += SPIKE_PD) {
#pragma unroll
        for (int u = 0; u < SPIKE_PD; u++) {
            const int k = kb + u;
            if (k >= re) break;
            double c0[DC], c1[DC], cg[DC];
            const bool v0 = has0 && k + d0 < re, v1 = has1 && k + d1 < re;
            const double cv = (k - r0 + b - cs <= b) ? st[u].cv : 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) { c0[m] = v0 ? st[u].row0[m] : 0.0; c1[m] = v1 ? st[u].row1[m] : 0.0; cg[m] = st[u].g[m]; }
            fetch(k + SPIKE_PD, st[u]);
            const double w = cv - acc0;                                 // lanes 0..DC-1: C_left(k, q) - pending sum of row k
            const double sh0 = lane_shift_down(acc0, DC), sh1 = lane_shift_down(acc1, DC);
            double sft0 = (lane + DC < 64) ? sh0 : sh1;
            if (!(lane + DC < T)) sft0 = 0.0;
            double sft1 = (lane + DC < 64) ? sh1 : 0.0;
            if (!(lane + 64 + DC < T)) sft1 = 0.0;
            double zz = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) zz += cg[m] * lane_bcast(w, m);       // z_k[lane] = sum_m G[lane][m] w[m]
            const bool pivot = k < r1;
            if (lane < DC) z[(size_t)k * DC + lane] = pivot ? zz : w;
            if (!pivot) zz = 0.0;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) { const double zm = lane_bcast(zz, m); s0 += c0[m] * zm; s1 += c1[m] * zm; }
            acc0 = sft0 + s0; acc1 = sft1 + s1;
        }
    }
}

// ---- 3. separator blocks: D = inner - Z^T Z (lower triangle, dense [Q][Q]), t = y_sep - Z^T y_seg ----------------------------
// grid (separators, lower 32x32 tiles + 1); the extra workgroup does the right-hand sides.
constexpr int SUB_TS = 32, SUB_KC = 64;
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_sub_sep_assemble(const double* __restrict__ band, const double* __restrict__ Z, const double* __restrict__ Y, const int* __restrict__ sep_lo,
                   const int* __restrict__ sep_rseg, const int* __restrict__ seg_lo, const int* __restrict__ seg_hi, int N, int b,
                   double* __restrict__ Dd, double* __restrict__ tt) {
    constexpr int BB = DC * DC;
    __shared__ double sA[SUB_TS][SUB_KC + 1], sB[SUB_TS][SUB_KC + 1];
    const int W = b + 1, n = N * DC, Q = b * DC, tid = threadIdx.x;
    const int s = blockIdx.x, p0 = sep_lo[s], rs = sep_rseg[s];
    const int k0 = seg_lo[rs] * DC, k1 = seg_hi[rs] * DC;
    const int ntl = (Q + SUB_TS - 1) / SUB_TS, ntiles = ntl * (ntl + 1) / 2;
    if ((int)blockIdx.y == ntiles) {                        // right-hand sides: one wave per output, lanes stride the segment rows
        const int wave = tid >> 6, lane = tid & 63;
        for (int o = wave; o < NR * Q; o += 4) {
            const int r = o / Q, q = o - r * Q;
            double acc = 0.0;
            for (int kk = k0 + lane; kk < k1; kk += 64) acc += Z[(size_t)q * n + kk] * Y[(size_t)r * n + kk];
            acc = wave_sum(acc);
            if (lane == 0) tt[((size_t)s * NR + r) * Q + q] = Y[(size_t)r * n + (size_t)p0 * DC + q] - acc;
        }
        return;
    }
    int ti = 0, tj = blockIdx.y;                            // lower tiles, row-major: (0,0) (1,0) (1,1) (2,0) ...
    while (tj > ti) { tj -= ti + 1; ti++; }
    const int tx = tid & 15, ty = tid >> 4;                 // outputs (ti*32 + 2*ty + {0,1}, tj*32 + 2*tx + {0,1})
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int kk0 = k0; kk0 < k1; kk0 += SUB_KC) {
        for (int e = tid; e < SUB_TS * SUB_KC; e += 256) {
            const int i = e / SUB_KC, kk = e - i * SUB_KC;
            const int qa = ti * SUB_TS + i, qb = tj * SUB_TS + i;
            const bool in = kk0 + kk < k1;
            sA[i][kk] = (in && qa < Q) ? Z[(size_t)qa * n + kk0 + kk] : 0.0;
            sB[i][kk] = (in && qb < Q) ? Z[(size_t)qb * n + kk0 + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < SUB_KC; kk++) {
            const double a0 = sA[2 * ty][kk], a1 = sA[2 * ty + 1][kk], b0 = sB[2 * tx][kk], b1 = sB[2 * tx + 1][kk];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const int q = ti * SUB_TS + 2 * ty + u, q2 = tj * SUB_TS + 2 * tx + v;
            if (q < Q && q2 <= q) {
                const int r = q / DC, a = q - r * DC, r2 = q2 / DC, a2 = q2 - r2 * DC;
                const double inner = band[((size_t)(p0 + r) * W + (r - r2)) * BB + a * DC + a2];
                Dd[((size_t)s * Q + q) * Q + q2] = inner - acc[u][v];
            }
        }
}

// ---- 4. block-tridiagonal chain over the separators of one component -----------------------------------------------------------
//   forward, j = 0..ns-1:  F_j = E_j Lc_{j-1}^-T,  D_j -= F_j F_j^T,  t_j -= F_j w_{j-1},  Lc_j = chol(D_j),  w_j = Lc_j^-1 t_j
//   backward:              x_j = Lc_j^-T (w_j - F_{j+1}^T x_{j+1})          -> Y rows of the separator
// E_j(i, c) = Z[c][(row of sep_j) i]: left by the spike kernel in the rows of sep_j.  One workgroup of 1024 per chain; LDS:
// packed lower triangle (Lc / D) + full F (column-major) + vectors = 8 (Q(Q+1)/2 + Q^2 + (2 NR + 1) Q) bytes <= 160 KB  <=>  Q <= 114.
// Right-looking eliminations with deferred scaling: column c is final after step c-1, every reader multiplies by 1/L_cc itself,
// one barrier per column.
template <int DC, int NR>
__global__ void __launch_bounds__(1024)
k_sub_sep_chain(const double* __restrict__ Z, const double* __restrict__ Dd, const double* __restrict__ tt, const int* __restrict__ chain_ptr,
                const int* __restrict__ sep_lo, int N, int b, double* __restrict__ Fbuf, double* __restrict__ Lbuf, double* __restrict__ wbuf,
                double* __restrict__ Y, int* __restrict__ fail_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Q = b * DC, n = N * DC, tid = threadIdx.x, nt = blockDim.x, NP = Q * (Q + 1) / 2;
    double* sL = lds;                  // [NP]   packed lower triangle, row-major: (i, c) at i(i+1)/2 + c
    double* sF = sL + NP;              // [Q][Q] column-major: F(i, c) at c*Q + i
    double* sT = sF + (size_t)Q * Q;   // [NR][Q] t_j -> w_j   (backward: v -> x_j)
    double* sW = sT + NR * Q;          // [NR][Q] w_{j-1}      (backward: x_{j+1})
    double* sInv = sW + NR * Q;        // [Q]    1 / L_cc
    const int s0 = chain_ptr[blockIdx.x], ns = chain_ptr[blockIdx.x + 1] - s0;
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    for (int j = 0; j < ns; j++) {
        const int s = s0 + j, p0 = sep_lo[s];
        for (int e = tid; e < NR * Q; e += nt) sT[e] = tt[(size_t)s * NR * Q + e];
        if (j > 0) {
            // ---- F = E Lc^-T with Lc = previous factor (still in sL, sInv)
            for (int e = tid; e < Q * Q; e += nt) { const int c = e / Q, i = e - c * Q; sF[e] = Z[(size_t)c * n + (size_t)p0 * DC + i]; }
            for (int c = 0; c < Q; c++) {
                __syncthreads();
                const double ic = sInv[c];
                const int rem = Q - 1 - c;
                for (int e = tid; e < rem * Q; e += nt) {
                    const int cp = c + 1 + e / Q, i = e % Q;
                    sF[cp * Q + i] -= sF[c * Q + i] * ic * sL[cp * (cp + 1) / 2 + c] ;
                }
            }
            __syncthreads();
            for (int e = tid; e < Q * Q; e += nt) sF[e] *= sInv[e / Q];
            __syncthreads();
            // ---- t_j -= F w_{j-1};  F to global for the backward pass
            for (int e = tid; e < NR * Q; e += nt) {
                const int r = e / Q, i = e - r * Q;
                double acc = 0.0;
                for (int c = 0; c < Q; c++) acc += sF[c * Q + i] * sW[r * Q + c];
                sT[e] -= acc;
            }
            for (int e = tid; e < Q * Q; e += nt) Fbuf[(size_t)s * Q * Q + e] = sF[e];
            __syncthreads();
        }
        // ---- D_j (- F F^T) into the packed triangle
        for (int e = tid; e < NP; e += nt) {
            int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while (i * (i + 1) / 2 > e) i--;
            while ((i + 1) * (i + 2) / 2 <= e) i++;
            const int c = e - i * (i + 1) / 2;
            double v = Dd[((size_t)s * Q + i) * Q + c];
            if (j > 0) for (int m = 0; m < Q; m++) v -= sF[m * Q + i] * sF[m * Q + c];
            sL[e] = v;
        }
        // ---- Cholesky, right-looking, forward substitution of t riding along
        for (int c = 0; c < Q; c++) {
            __syncthreads();
            const double d = sL[c * (c + 1) / 2 + c];
            if (!(d > 0.0)) { if (tid == 0) *fail_flag = 1; }
            const double ic = 1.0 / sqrt(d > 0.0 ? d : 1.0);
            if (tid == 0) sInv[c] = ic;
            const int rem = Q - 1 - c;
            for (int e = tid; e < rem * (rem + NR); e += nt) {
                const int i = c + 1 + e / (rem + NR), u = e % (rem + NR);
                const double lic = sL[i * (i + 1) / 2 + c] * ic;
                if (u < rem) {
                    const int cp = c + 1 + u;
                    if (cp <= i) sL[i * (i + 1) / 2 + cp] -= lic * sL[cp * (cp + 1) / 2 + c] * ic;
                } else {
                    const int r = u - rem;
                    sT[r * Q + i] -= lic * sT[r * Q + c] * ic;
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < NP; e += nt) {
            int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while (i * (i + 1) / 2 > e) i--;
            while ((i + 1) * (i + 2) / 2 <= e) i++;
            const int c = e - i * (i + 1) / 2;
            const double v = (i == c) ? sqrt(fmax(sL[e], 0.0)) : sL[e] * sInv[c];
            sL[e] = v; Lbuf[(size_t)s * NP + e] = v;
        }
        for (int e = tid; e < NR * Q; e += nt) { const double wv = sT[e] * sInv[e % Q]; sW[e] = wv; wbuf[(size_t)s * NR * Q + e] = wv; }
        __syncthreads();
    }
    // ---- backward
    for (int j = ns - 1; j >= 0; j--) {
        const int s = s0 + j, p0 = sep_lo[s];
        if (j < ns - 1) {                                   // reload this separator's factor and w; v = w - F_{j+1}^T x_{j+1}
            for (int e = tid; e < NP; e += nt) sL[e] = Lbuf[(size_t)s * NP + e];
            const double* Fn = Fbuf + (size_t)(s + 1) * Q * Q;
            for (int o = wave; o < NR * Q; o += nw) {
                const int r = o / Q, c = o - r * Q;
                double acc = 0.0;
                for (int i = lane; i < Q; i += 64) acc += Fn[(size_t)c * Q + i] * sW[r * Q + i];
                acc = wave_sum(acc);
                if (lane == 0) sT[o] = wbuf[(size_t)s * NR * Q + o] - acc;
            }
        } else {
            for (int e = tid; e < NR * Q; e += nt) sT[e] = sW[e];
        }
        // x = Lc^-T v, right-looking from the last row, deferred scaling
        for (int c = Q - 1; c >= 0; c--) {
            __syncthreads();
            const double ic = 1.0 / sL[c * (c + 1) / 2 + c];
            for (int e = tid; e < NR * c; e += nt) {
                const int r = e / c, cp = e - r * c;
                sT[r * Q + cp] -= sL[c * (c + 1) / 2 + cp] * sT[r * Q + c] * ic;
            }
        }
        __syncthreads();
        for (int e = tid; e < NR * Q; e += nt) {
            const int r = e / Q, c = e - r * Q;
            const double x = sT[e] / sL[c * (c + 1) / 2 + c];
            sW[e] = x; Y[(size_t)r * n + (size_t)p0 * DC + c] = x;
        }
        __syncthreads();
    }
}

// ---- 5. y(seg) -= Z x(separator in front) ------------------------------------------------------------------------------------------
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_sub_apply_left(const double* __restrict__ Z, double* __restrict__ Y, const int* __restrict__ seg_lo, const int* __restrict__ seg_hi,
                 const int* __restrict__ left_segs, int N, int b) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Q = b * DC, n = N * DC, tid = threadIdx.x;
    const int seg = left_segs[blockIdx.x], r0 = seg_lo[seg], r1 = seg_hi[seg];
    const int kk = r0 * DC + blockIdx.y * 256 + tid;
    if (r0 * DC + (int)blockIdx.y * 256 >= r1 * DC) return;
    for (int e = tid; e < NR * Q; e += 256) { const int r = e / Q, q = e - r * Q; lds[e] = Y[(size_t)r * n + (size_t)(r0 - b) * DC + q]; }
    __syncthreads();
    if (kk >= r1 * DC) return;
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = 0.0;
    for (int q = 0; q < Q; q++) {
        const double zv = Z[(size_t)q * n + kk];
#pragma unroll
        for (int r = 0; r < NR; r++) acc[r] += zv * lds[r * Q + q];
    }
#pragma unroll
    for (int r = 0; r < NR; r++) Y[(size_t)r * n + kk] -= acc[r];
}

}  // namespace ssfm
