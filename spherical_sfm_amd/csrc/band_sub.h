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
#include <cmath>
#include <cstdlib>
#include <vector>
#include "band_kernels2.h"
#include "ring_comp.h"
#include "ring_schedule.h"

namespace ssfm {

// ---- host: segment / separator tables ----------------------------------------------------------------------------------
struct BandSub {
    bool enabled = false;
    int nseg = 0, nsep = 0, nchain = 0, nleft = 0;
    std::vector<int> seg_lo, seg_hi, seg_wend;      // pivots [lo, hi), window end (hi, or hi + b with a separator behind)
    std::vector<int> left_segs;                     // segments with a separator in front
    std::vector<int> sep_lo, sep_rseg;              // first row of the separator; the segment behind it
    std::vector<int> chain_ptr;                     // separators of chain c: [chain_ptr[c], chain_ptr[c+1])  (one chain per cut component)
    // twisted components (ba_flatten.h: band_twist_plan): seg_0 | sep | seg_1 reversed | copy of sep.  Both segments have their separator
    // rows behind them; tw_lo / tw_hi = the separator as a small component of its own, tw_copy = first row of its second copy
    int ntwist = 0;
    std::vector<int> tw_lo, tw_hi, tw_copy;
    std::vector<int> seg_given;                     // per segment: rows where the solution of its given rows really is (reversed segments), else -1
    // fused launches (segments + the twisted separators that wait for them): tables over nseg + ntwist workgroups; flag 2t / 2t+1 = the halves of twisted component t
    std::vector<int> seg_twist;                     // per segment: 2t + half of the twisted component it belongs to, else -1
    std::vector<int> fz_lo, fz_hi, fz_wend, fz_merge, fz_await, fz_signal;
    // rings (round 5, band_ring.h): arcs are segments, their separators a cycle solved by cyclic reduction.  sep_copy: first band row of a separator's copy slot (the
    // last separator of a ring is also "in front of" its first arc) or -1; ring_seps: (first separator id, cuts) per ring; the schedule (ring_schedule.h): RING_REC ints
    // per elimination, eliminations of parallel step s = records [ring_step_ptr[s], ring_step_ptr[s + 1]), the tail of ring g = [ring_tail_ptr[g], ring_tail_ptr[g + 1])
    int nring = 0;
    std::vector<int> sep_copy; std::vector<std::pair<int, int>> ring_seps;
    std::vector<int> ring_rec, ring_step_ptr, ring_tail_ptr;
};

// Cost model in microseconds, fitted to MI355X measurements (profiles/r01_notes.md): per block row of a segment the factorisation
// (0.13 b - 0.17), the spike (1.1) and the back substitution (0.43); per separator 125 (b dc / 114)^2.5 for the chain (matrix-core kernel, r02); 170 fixed for the
// extra launches, the assembly and the memsets.  Segments must be at least b + 1 rows.  Components shorter than SUB_MIN_ROWS stay on
// one workgroup (config 2's rings of 75: 135 us uncut against ~180 us cut in two).
constexpr int SUB_MIN_ROWS = 512;                  // = BAND_CUT_MIN_ROWS (ba_flatten.h): shorter components are twisted instead
inline int sub_choose_segments(int rows, int b, int dc) {
    if (rows < SUB_MIN_ROWS) return 1;
    const double t_chol = std::max(0.13 * b - 0.17, 0.4), t_spike = 1.1, t_back = 0.43;
    const double t_sep = 125.0 * std::pow(b * dc / 114.0, 2.5), t_fixed = 170.0;
    int best = 1; double best_t = rows * (t_chol + t_back);
    for (int P = 2; P <= 64; P++) {
        const int m = (rows - (P - 1) * b) / P;
        if (m < b + 1) break;
        const double t = (m + b) * (t_chol + t_spike + t_back) + (P - 1) * t_sep + t_fixed;
        if (t < best_t) { best_t = t; best = P; }
    }
    return best;
}

inline void sub_build(const std::vector<int>& comp_ptr, const std::vector<char>& comp_twist, int b, int dc, BandSub& S, const std::vector<RingComp>* rings = nullptr) {
    S = BandSub();
    const char* env = std::getenv("SSFM_BAND_SEGMENTS");               // 1 = never cut; P >= 2 = cut every component that can take it into P
    const int forced = env ? std::atoi(env) : 0;
    const bool can_cut = b >= 1 && b * dc <= 114 && forced != 1;       // separator blocks must fit the chain kernel's LDS (see k_sub_sep_chain)
    S.chain_ptr.assign(1, 0);
    for (int pass = 0; pass < 2; pass++)                               // rings last: the separators of a chain are a contiguous range of ids (chain_ptr)
    for (size_t c = 0; c + 1 < comp_ptr.size(); c++) {
        const int c0 = comp_ptr[c], rows = comp_ptr[c + 1] - c0;
        const RingComp* ring = nullptr;
        if (rings) for (const RingComp& rc : *rings) if (rc.comp == (int)c) ring = &rc;
        if ((ring != nullptr) != (pass == 1)) continue;
        if (ring) {                                                    // rows = n + b: copy of S_{m-1} | A_0 | S_0 | ... | A_{m-1} | S_{m-1}   (ba_flatten.h: band_plan)
            const int m = (int)ring->arc_len.size(), seg0 = (int)S.seg_lo.size(), sep0 = (int)S.sep_lo.size();
            int pos = c0 + b;
            for (int k = 0; k < m; k++) {
                S.left_segs.push_back((int)S.seg_lo.size());
                S.seg_lo.push_back(pos); S.seg_hi.push_back(pos + ring->arc_len[k]); S.seg_wend.push_back(pos + ring->arc_len[k] + b); S.seg_given.push_back(-1); S.seg_twist.push_back(-1);
                pos += ring->arc_len[k];
                S.sep_lo.push_back(pos); S.sep_rseg.push_back(seg0 + (k + 1) % m); S.sep_copy.push_back(k == m - 1 ? c0 : -1);
                pos += b;
            }
            S.ring_seps.push_back({sep0, m}); S.nring++;
            S.enabled = true;
            continue;
        }
        if (c < comp_twist.size() && comp_twist[c]) {                  // rows = n + b: seg_0 | sep | seg_1 reversed | copy of sep
            const int n = rows - b, m0 = (n - b) / 2, m1 = n - b - m0, t2 = 2 * (int)S.tw_lo.size();
            S.seg_lo.push_back(c0); S.seg_hi.push_back(c0 + m0); S.seg_wend.push_back(c0 + m0 + b); S.seg_given.push_back(-1); S.seg_twist.push_back(t2);
            S.seg_lo.push_back(c0 + m0 + b); S.seg_hi.push_back(c0 + m0 + b + m1); S.seg_wend.push_back(c0 + rows); S.seg_given.push_back(c0 + m0); S.seg_twist.push_back(t2 + 1);
            S.tw_lo.push_back(c0 + m0); S.tw_hi.push_back(c0 + m0 + b); S.tw_copy.push_back(c0 + m0 + b + m1);
            S.enabled = true;
            continue;
        }
        int P = (forced >= 2 ? forced : sub_choose_segments(rows, b, dc));
        if (!can_cut) P = 1;
        while (P > 1 && (rows - (P - 1) * b) / P < b + 1) P--;
        const int m_total = rows - (P - 1) * b;
        int pos = c0;
        for (int i = 0; i < P; i++) {
            const int m = m_total / P + (i < m_total % P ? 1 : 0);
            if (i > 0) S.left_segs.push_back((int)S.seg_lo.size());
            S.seg_lo.push_back(pos); S.seg_hi.push_back(pos + m); S.seg_wend.push_back(i + 1 < P ? pos + m + b : pos + m); S.seg_given.push_back(-1); S.seg_twist.push_back(-1);
            pos += m;
            if (i + 1 < P) { S.sep_lo.push_back(pos); S.sep_rseg.push_back((int)S.seg_lo.size()); S.sep_copy.push_back(-1); pos += b; }
        }
        if (P > 1) { S.enabled = true; S.chain_ptr.push_back((int)S.sep_lo.size()); }
    }
    S.nseg = (int)S.seg_lo.size(); S.nsep = (int)S.sep_lo.size(); S.nchain = (int)S.chain_ptr.size() - 1; S.nleft = (int)S.left_segs.size();
    S.ntwist = (int)S.tw_lo.size();
    if (!S.enabled) { S = BandSub(); return; }
    if (S.nring > 0) ring_schedule(S.ring_seps, S.sep_lo, S.sep_copy, S.ring_rec, S.ring_step_ptr, S.ring_tail_ptr);
    S.fz_lo = S.seg_lo; S.fz_hi = S.seg_hi; S.fz_wend = S.seg_wend; S.fz_merge.assign(S.nseg, -1); S.fz_await.assign(S.nseg, -1); S.fz_signal = S.seg_twist;
    for (int t = 0; t < S.ntwist; t++) { S.fz_lo.push_back(S.tw_lo[t]); S.fz_hi.push_back(S.tw_hi[t]); S.fz_wend.push_back(S.tw_hi[t]); S.fz_merge.push_back(S.tw_copy[t]); S.fz_await.push_back(2 * t); S.fz_signal.push_back(-1); }
}

// ---- 2. spike: Z(:, q) = L_seg^-1 C_left(:, q), one wave per (segment, SPIKE_NC columns) -----------------------------------
// Column q = (separator row r0 - b + q / DC, component q % DC).  C_left lives in the band rows of the segment's first b rows
// (blocks whose column lies in front of r0).  Right-looking: task t = (d-1)*DC + a (d = 1..b) owns the pending sum of row
// k + (d-1), component a; lanes carry tasks t = lane and t = lane + 64.  Rows [r1, re) take no pivot: they receive -sum = E.
// Z: [b*DC][N*DC], row index = global scalar row.  A step is latency bound and every column streams the same factor rows, so a
// wave carries SPIKE_NC columns through one stream of L.  Round 1 (half-width 19, flat 64-bit indices): one column per wave was L2-bandwidth bound and four
// columns per wave won.  Round 2 (half-width 14 after the ordering change, scalar row bases): a step is bound by what one wave can issue, the columns of a
// wave run one after the other inside a step, and there are CUs to spare -- 4 columns 210 us, 2 columns 169 us, 1 column 130 us per launch at the configs[4] size.
// Round 5: narrow bands (the ring-native layout halves the half-width) turn the balance again -- the arithmetic of a step shrinks with the band, the factor rows every
// column streams do not: 42 columns x 102 arcs at the configs[4] size read the whole band 42 times (311 MB from L2, ~72 us whatever the number of arcs).  NC is a template
// parameter now; the launch picks it by the number of one-wave workgroups it would have (ba_handle.h, r05ac: one column per wave until the chip is full).
constexpr int SPIKE_PD = 4, SPIKE_NC = 1;
template <int DC, int NC = SPIKE_NC>
__global__ void __launch_bounds__(64)
k_sub_spike_fwd(const double* __restrict__ band, const double* __restrict__ Ginv, double* __restrict__ Z, const int* __restrict__ seg_lo,
                const int* __restrict__ seg_hi, const int* __restrict__ seg_wend, const int* __restrict__ left_segs, int N, int b) {
    constexpr int BB = DC * DC;
    const int W = b + 1, n = N * DC, lane = threadIdx.x, Q = b * DC;
    const int seg = left_segs[blockIdx.x], q0 = blockIdx.y * NC;
    const int r0 = seg_lo[seg], r1 = seg_hi[seg], re = seg_wend[seg];
    int cs[NC], cc[NC]; bool qv[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { const int q = min(q0 + c, Q - 1); qv[c] = q0 + c < Q; cs[c] = q / DC; cc[c] = q - cs[c] * DC; }
    const int T = b * DC;
    const int t0 = min(lane, T - 1), t1 = min(lane + 64, T - 1);
    const int d0 = t0 / DC + 1, a0 = t0 - (d0 - 1) * DC, d1 = t1 / DC + 1, a1 = t1 - (d1 - 1) * DC;
    const bool has0 = lane < T, has1 = lane + 64 < T;
    const int lc = min(lane, DC - 1);
    struct Stage { double row0[DC], row1[DC], g[DC], cv[NC]; };
    Stage st[SPIKE_PD];
    // (wave-uniform row bases + 32-bit lane offsets instead of a flat 64-bit index per load, as in k_band_back_v2)
    const size_t row_stride = (size_t)W * BB;
    const int o0 = d0 * BB + a0 * DC, o1 = d1 * BB + a1 * DC;
    auto fetch = [&](int k, Stage& s) {    // row a of L(k+d, k) for this lane's tasks; row `lane` of G_k; C_left(k, q)[lane]
        const int kc = min(k, re - 1);
        const double* __restrict__ base = band + (size_t)kc * row_stride;
        const double* __restrict__ gpt = Ginv + (size_t)kc * BB;
        // rows kc + d of this lane's tasks: the lane offset carries d rows; rows past the window are clamped to its last row
        const int k0 = min(kc + d0, re - 1) - kc, k1 = min(kc + d1, re - 1) - kc;
        const int f0 = k0 * (int)row_stride + o0, f1 = k1 * (int)row_stride + o1;
#pragma unroll
        for (int m = 0; m < DC; m++) {
            s.row0[m] = base[f0 + m];
            s.row1[m] = base[f1 + m];
            s.g[m] = gpt[lc * DC + m];                                    // G[lane][m], zero for m > lane
        }
#pragma unroll
        for (int c = 0; c < NC; c++) { const int dl = min(kc - r0 + b - cs[c], b); s.cv[c] = base[dl * BB + lc * DC + cc[c]]; }
    };
#pragma unroll
    for (int u = 0; u < SPIKE_PD; u++) fetch(r0 + u, st[u]);
    double acc0[NC], acc1[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { acc0[c] = 0.0; acc1[c] = 0.0; }
    for (int kb = r0; kb < re; kb += SPIKE_PD) {
#pragma unroll
        for (int u = 0; u < SPIKE_PD; u++) {
            const int k = kb + u;
            if (k >= re) break;
            double c0[DC], c1[DC], cg[DC], cvv[NC];
            const bool v0 = has0 && k + d0 < re, v1 = has1 && k + d1 < re;
#pragma unroll
            for (int c = 0; c < NC; c++) cvv[c] = (k - r0 <= cs[c]) ? st[u].cv[c] : 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) { c0[m] = st[u].row0[m]; c1[m] = st[u].row1[m]; cg[m] = st[u].g[m]; }      // rows past the window: their sums are dropped below
            fetch(k + SPIKE_PD, st[u]);
            const bool pivot = k < r1;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const double w = cvv[c] - acc0[c];                      // lanes 0..DC-1: C_left(k, q) - pending sum of row k
                const double sh0 = lane_shift_down(acc0[c], DC), sh1 = lane_shift_down(acc1[c], DC);
                double sft0 = (lane + DC < 64) ? sh0 : sh1;
                if (!(lane + DC < T)) sft0 = 0.0;
                double sft1 = (lane + DC < 64) ? sh1 : 0.0;
                if (!(lane + 64 + DC < T)) sft1 = 0.0;
                double zz = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) zz += cg[m] * lane_bcast(w, m);   // z_k[lane] = sum_m G[lane][m] w[m]
                if (lane < DC && qv[c]) Z[(size_t)(q0 + c) * n + (size_t)k * DC + lane] = pivot ? zz : w;
                if (!pivot) zz = 0.0;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) { const double zm = lane_bcast(zz, m); s0 += c0[m] * zm; s1 += c1[m] * zm; }
                acc0[c] = sft0 + (v0 ? s0 : 0.0); acc1[c] = sft1 + (v1 ? s1 : 0.0);
            }
        }
    }
}

// ---- 3. separator blocks: D = inner - Z^T Z (lower triangle, dense [Q][Q]), t = y_sep - Z^T y_seg ----------------------------
// grid (separators, lower 32x32 tiles, K chunks).  Dd and tt are zeroed beforehand; every chunk adds its part, chunk 0 also the
// inner blocks / y_sep.  The tiles of the first tile column carry the right-hand sides along (NR more columns of the product).
constexpr int SUB_TS = 32, SUB_KC = 64;
template <int DC, int NR>
__global__ void __launch_bounds__(256)
k_sub_sep_assemble(const double* __restrict__ band, const double* __restrict__ Z, const double* __restrict__ Y, const int* __restrict__ sep_lo,
                   const int* __restrict__ sep_rseg, const int* __restrict__ seg_lo, const int* __restrict__ seg_hi, int N, int b,
                   double* __restrict__ Dd, double* __restrict__ tt) {
    constexpr int BB = DC * DC;
    __shared__ double sA[SUB_TS][SUB_KC + 1], sB[SUB_TS][SUB_KC + 1], sY[NR][SUB_KC + 1];
    const int W = b + 1, n = N * DC, Q = b * DC, tid = threadIdx.x;
    const int s = blockIdx.x, p0 = sep_lo[s], rs = sep_rseg[s];
    const int ka = seg_lo[rs] * DC, kb = seg_hi[rs] * DC;
    const int kchunk = (((kb - ka + (int)gridDim.z - 1) / (int)gridDim.z + SUB_KC - 1) / SUB_KC) * SUB_KC;
    const int k0 = ka + (int)blockIdx.z * kchunk, k1 = min(kb, k0 + kchunk);
    const bool first = blockIdx.z == 0;
    if (k0 >= k1 && !first) return;
    int ti = 0, tj = blockIdx.y;                            // lower tiles, row-major: (0,0) (1,0) (1,1) (2,0) ...
    while (tj > ti) { tj -= ti + 1; ti++; }
    const int tx = tid & 15, ty = tid >> 4;                 // outputs (ti*32 + 2*ty + {0,1}, tj*32 + 2*tx + {0,1})
    const bool with_rhs = tj == 0;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, accy[2][NR];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < NR; r++) accy[u][r] = 0.0;
    for (int kk0 = k0; kk0 < k1; kk0 += SUB_KC) {
        for (int e = tid; e < SUB_TS * SUB_KC; e += 256) {
            const int i = e / SUB_KC, kk = e - i * SUB_KC;
            const int qa = ti * SUB_TS + i, qb = tj * SUB_TS + i;
            const bool in = kk0 + kk < k1;
            sA[i][kk] = (in && qa < Q) ? Z[(size_t)qa * n + kk0 + kk] : 0.0;
            sB[i][kk] = (in && qb < Q) ? Z[(size_t)qb * n + kk0 + kk] : 0.0;
        }
        if (with_rhs) for (int e = tid; e < NR * SUB_KC; e += 256) {
            const int r = e / SUB_KC, kk = e - r * SUB_KC;
            sY[r][kk] = (kk0 + kk < k1) ? Y[(size_t)r * n + kk0 + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < SUB_KC; kk++) {
            const double a0 = sA[2 * ty][kk], a1 = sA[2 * ty + 1][kk], b0 = sB[2 * tx][kk], b1 = sB[2 * tx + 1][kk];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
            if (with_rhs && tx == 0) {
#pragma unroll
                for (int r = 0; r < NR; r++) { const double yv = sY[r][kk]; accy[0][r] += a0 * yv; accy[1][r] += a1 * yv; }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int q = ti * SUB_TS + 2 * ty + u;
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const int q2 = tj * SUB_TS + 2 * tx + v;
            if (q < Q && q2 <= q) {
                const int r = q / DC, a = q - r * DC, r2 = q2 / DC, a2 = q2 - r2 * DC;
                const double inner = first ? band[((size_t)(p0 + r) * W + (r - r2)) * BB + a * DC + a2] : 0.0;
                unsafeAtomicAdd(&Dd[((size_t)s * Q + q) * Q + q2], inner - acc[u][v]);
            }
        }
        if (with_rhs && tx == 0 && q < Q) {
#pragma unroll
            for (int r = 0; r < NR; r++)
                unsafeAtomicAdd(&tt[((size_t)s * NR + r) * Q + q], (first ? Y[(size_t)r * n + (size_t)p0 * DC + q] : 0.0) - accy[u][r]);
        }
    }
}

typedef double v4d_t __attribute__((ext_vector_type(4)));

// ---- 3b. the same on the matrix cores (round 4) ---------------------------------------------------------------------------------------
// D_s = inner - Z^T Z is a rank-K update with K = the rows of the segment behind the separator (~1200 at the configs[4] size): 16x16 output tiles of
// v_mfma_f64_16x16x4 with the operands STRAIGHT FROM GLOBAL MEMORY -- Z is [Q][n] with the segment rows contiguous, lane (li, lk) loads Z[r0 + li][k0 + 4 lk .. + 3]
// (32 bytes; the sixteen lanes of a k-slice group cover 16 rows x 128 contiguous bytes per trip) and feeds one value to each of four products (any partition of the
// k index over the products is a valid one as long as both operands use the same).  One workgroup per (separator, lower tile or row group of the right-hand sides), its
// sixteen waves split K and fold their accumulators through LDS in wave order: no atomics, no zeroing of Dd / tt beforehand, and the result does not depend on
// the order in which workgroups finish (the VALU kernel above added K chunks with atomics).  80 -> see profiles/r04_notes.md us at the configs[4] size.
constexpr int ASM_NW = 8;                          // waves per workgroup = K chunks of a tile
template <int DC, int NR>
__global__ void __launch_bounds__(64 * ASM_NW)
k_sub_sep_assemble_mfma(const double* __restrict__ band, const double* __restrict__ Z, const double* __restrict__ Y, const int* __restrict__ sep_lo,
                        const int* __restrict__ sep_rseg, const int* __restrict__ seg_lo, const int* __restrict__ seg_hi, int N, int b,
                        double* __restrict__ Dd, double* __restrict__ tt) {
    constexpr int BB = DC * DC, TB = 16;
    __shared__ double red[ASM_NW - 1][64][4];
    const int W = b + 1, n = N * DC, Q = b * DC, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int TQ = (Q + TB - 1) / TB, ntile = TQ * (TQ + 1) / 2;
    // 1-D grid, separator-major task list, each XCD a contiguous eighth of it: the 21 tiles of a separator re-read its Z rows (0.8 MB) from that XCD's L2
    // instead of from the Infinity Cache once per XCD (116 MB of operand reads at the configs[4] size: 28 us at ~4 TB/s before)
    const int lin = xcd_contiguous_block(blockIdx.x, gridDim.x), ntask = ntile + TQ;
    const int s = lin / ntask, ty = lin - s * ntask, p0 = sep_lo[s], rs = sep_rseg[s];
    const int ka = seg_lo[rs] * DC, kb = seg_hi[rs] * DC;
    const int kq = (((kb - ka + ASM_NW - 1) / ASM_NW + 15) / 16) * 16;       // K per wave, a multiple of 16
    const int k0w = ka + wave * kq, k1w = min(kb, k0w + kq);
    const bool is_rhs = ty >= ntile;
    int I = 0, J = 0;
    if (!is_rhs) { const int t = ty; while ((I + 1) * (I + 2) / 2 <= t) I++; J = t - I * (I + 1) / 2; } else I = ty - ntile;
    const int r0 = TB * I, c0 = TB * J;
    const double* za = Z + (size_t)min(r0 + li, Q - 1) * n;                  // (rows beyond Q only feed outputs that are not stored)
    const double* zb = Z + (size_t)min(c0 + li, Q - 1) * n;
    double acc4[4] = {0.0, 0.0, 0.0, 0.0};
    if (!is_rhs) {
        v4d_t acc = {0.0, 0.0, 0.0, 0.0};
        const bool diag = I == J;
        auto load4 = [&](const double* zp, int k, double (&v)[4]) {
            // 16-byte loads need an even element offset: rows start at multiples of n = N DC and k at multiples of DC (+ 4 lk), so DC = 6 is always
            // aligned; the unmerged 3-dof band (ssfm_band_solve_probe with dc = 3, odd N or an odd segment start) is not and takes the scalar loads
            if (k + 4 <= k1w && (DC % 2 == 0 || ((reinterpret_cast<uintptr_t>(zp + k) & 15) == 0))) { const double2 v01 = *reinterpret_cast<const double2*>(zp + k), v23 = *reinterpret_cast<const double2*>(zp + k + 2); v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y; }
            else {
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (k + u < k1w) ? zp[k + u] : 0.0;
            }
        };
        constexpr int DEPTH = 5;                                             // trips of 16 k whose loads are in flight together (a trip is one global round trip otherwise)
        for (int k0 = k0w; k0 < k1w; k0 += 16 * DEPTH) {
            double a[DEPTH][4], bb[DEPTH][4];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const int k = k0 + 16 * d + 4 * lk;
                if (k0 + 16 * d < k1w) { load4(za, k, a[d]); if (!diag) load4(zb, k, bb[d]); }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                if (k0 + 16 * d < k1w) {
#pragma unroll
                    for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][u], diag ? a[d][u] : bb[d][u], acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc4[q] = acc[q];
    } else {
        // right-hand sides: t(q) = y_sep(q) - sum_k Z[q][k] y(k) for the 16 rows of group I; lane (li, lk) sums its k-slices, folded over lk below
        for (int k0 = k0w; k0 < k1w; k0 += 16) {
            const int k = k0 + 4 * lk;
#pragma unroll
            for (int u = 0; u < 4; u++) if (k + u < k1w) {
                const double z = za[k + u];
#pragma unroll
                for (int r = 0; r < NR; r++) acc4[r] += z * Y[(size_t)r * n + k + u];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; r++) { double v = acc4[r]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); acc4[r] = v; }
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) red[wave - 1][lane][q] = acc4[q];
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 0; w < ASM_NW - 1; w++)                                     // in wave order: the same bits whatever the schedule
#pragma unroll
        for (int q = 0; q < 4; q++) acc4[q] += red[w][lane][q];
    if (!is_rhs) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = r0 + lk + 4 * q, col = c0 + li;
            if (row < Q && col <= row) {
                const int r = row / DC, a = row - r * DC, r2 = col / DC, a2 = col - r2 * DC;
                Dd[((size_t)s * Q + row) * Q + col] = band[((size_t)(p0 + r) * W + (r - r2)) * BB + a * DC + a2] - acc4[q];
            }
        }
    } else if (lk == 0 && r0 + li < Q) {
#pragma unroll
        for (int r = 0; r < NR; r++) tt[((size_t)s * NR + r) * Q + r0 + li] = Y[(size_t)r * n + (size_t)p0 * DC + r0 + li] - acc4[r];
    }
}

// ---- 4. block-tridiagonal chain over the separators of one component -----------------------------------------------------------
//   forward, j = 0..ns-1:  F_j = E_j Lc_{j-1}^-T,  D_j -= F_j F_j^T,  t_j -= F_j w_{j-1},  Lc_j = chol(D_j),  w_j = Lc_j^-1 t_j
//   backward:              x_j = Lc_j^-T (w_j - F_{j+1}^T x_{j+1})          -> Y rows of the separator
// E_j(i, c) = Z[c][(row of sep_j) i]: left by the spike kernel in the rows of sep_j.  One workgroup of 1024 per chain; LDS:
// packed lower triangle (Lc / D) + full F (column-major) + vectors = 8 (Q(Q+1)/2 + Q^2 + 2 NR Q) bytes <= 160 KB  <=>  Q <= 114.
// All three eliminations are blocked by DC columns: the DCxDC diagonal block is factored AND inverted by one wave
// (wave_chol_inverse), its inverse G replaces it in the packed triangle, panels are products with G^T (no substitution chain),
// trailing updates take DC columns per pass.  Thread (tx, ty) = (tid & 127, tid >> 7): tx = row, ty strides the columns.
template <int DC, int NR>
__global__ void __launch_bounds__(1024)
k_sub_sep_chain(const double* __restrict__ Z, const double* __restrict__ Dd, const double* __restrict__ tt, const int* __restrict__ chain_ptr,
                const int* __restrict__ sep_lo, int N, int b, double* __restrict__ Fbuf, double* __restrict__ Lbuf, double* __restrict__ wbuf,
                double* __restrict__ Y, int* __restrict__ fail_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NB = DC;
    const int Q = b * DC, n = N * DC, tid = threadIdx.x, nt = blockDim.x, NP = Q * (Q + 1) / 2;
    double* sL = lds;                  // [NP]   packed lower triangle, row-major: (i, c) at i(i+1)/2 + c; diagonal blocks hold G = L_blk^-1
    double* sF = sL + NP;              // [Q][Q] column-major: F(i, c) at c*Q + i
    double* sT = sF + (size_t)Q * Q;   // [NR][Q] t_j -> w_j   (backward: v -> x_j)
    double* sW = sT + NR * Q;          // [NR][Q] w_{j-1}      (backward: x_{j+1})
    const int s0 = chain_ptr[blockIdx.x], ns = chain_ptr[blockIdx.x + 1] - s0;
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int tx = tid & 127, ty = tid >> 7;
    const bool rowt = tx < Q;
#define PK(i_, c_) ((i_) * ((i_) + 1) / 2 + (c_))
    for (int j = 0; j < ns; j++) {
        const int s = s0 + j, p0 = sep_lo[s];
        for (int e = tid; e < NR * Q; e += nt) sT[e] = tt[(size_t)s * NR * Q + e];
        if (j > 0) {
            // ---- F = E Lc^-T with Lc = previous factor (still in sL)
            if (rowt) for (int c = ty; c < Q; c += 8) sF[c * Q + tx] = Z[(size_t)c * n + (size_t)p0 * DC + tx];
            for (int c0 = 0; c0 < Q; c0 += NB) {
                __syncthreads();
                if (ty == 0 && rowt) {                                  // panel: F(:, c0..) = E'(:, c0..) G^T
                    double ev[NB], pv[NB];
#pragma unroll
                    for (int k = 0; k < NB; k++) ev[k] = sF[(c0 + k) * Q + tx];
#pragma unroll
                    for (int k = 0; k < NB; k++) { double a = 0.0;
#pragma unroll
                        for (int m = 0; m <= k; m++) a += ev[m] * sL[PK(c0 + k, c0 + m)];
                        pv[k] = a; }
#pragma unroll
                    for (int k = 0; k < NB; k++) sF[(c0 + k) * Q + tx] = pv[k];
                }
                __syncthreads();
                if (rowt) {
                    double pv[NB];
#pragma unroll
                    for (int k = 0; k < NB; k++) pv[k] = sF[(c0 + k) * Q + tx];
                    for (int cp = c0 + NB + ty; cp < Q; cp += 8) {
                        const double* Lr = sL + PK(cp, c0);
                        double v = sF[cp * Q + tx];
#pragma unroll
                        for (int k = 0; k < NB; k++) v -= pv[k] * Lr[k];
                        sF[cp * Q + tx] = v;
                    }
                }
            }
            __syncthreads();
            // ---- t_j -= F w_{j-1};  F to global for the backward pass
            if (rowt && ty < NR) {
                double acc = 0.0;
                for (int c = 0; c < Q; c++) acc += sF[c * Q + tx] * sW[ty * Q + c];
                sT[ty * Q + tx] -= acc;
            }
            for (int e = tid; e < Q * Q; e += nt) Fbuf[(size_t)s * Q * Q + e] = sF[e];
        }
        // ---- D_j into the packed triangle (previous factor is dead: it went to Lbuf)
        __syncthreads();
        if (rowt) for (int cp = ty; cp <= tx; cp += 8) sL[PK(tx, cp)] = Dd[((size_t)s * Q + tx) * Q + cp];
        __syncthreads();
        if (j > 0) {
            // ---- D_j -= F F^T, 4x4 register tiles over the lower triangle
            const int R4 = (Q + 3) / 4, ntile = R4 * (R4 + 1) / 2;
            if (tid < ntile) {
                int ti = (int)((sqrt(8.0 * tid + 1.0) - 1.0) * 0.5);
                while (ti * (ti + 1) / 2 > tid) ti--;
                while ((ti + 1) * (ti + 2) / 2 <= tid) ti++;
                const int tj = tid - ti * (ti + 1) / 2;
                int ri[4], rj[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { ri[u] = min(4 * ti + u, Q - 1); rj[u] = min(4 * tj + u, Q - 1); }
                double acc[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[u][v] = 0.0;
                for (int m = 0; m < Q; m++) {
                    const double* col = sF + m * Q;
                    double av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { av[u] = col[ri[u]]; bv[u] = col[rj[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int v = 0; v < 4; v++) acc[u][v] += av[u] * bv[v];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int i = 4 * ti + u, c = 4 * tj + v;
                        if (i < Q && c <= i) sL[PK(i, c)] -= acc[u][v];
                    }
            }
        }
        // ---- blocked Cholesky, forward substitution of t riding along
        for (int c0 = 0; c0 < Q; c0 += NB) {
            __syncthreads();
            if (wave == 0) {
                double row[NB], g[NB];
#pragma unroll
                for (int c = 0; c < NB; c++) row[c] = (lane < NB) ? sL[PK(c0 + max(lane, c), c0 + min(lane, c))] : ((lane == c) ? 1.0 : 0.0);
                if (!wave_chol_inverse<NB>(row, g) && lane == 0) *fail_flag = 1;
                if (lane < NB) {
#pragma unroll
                    for (int r = 0; r < NB; r++) if (r >= lane) sL[PK(c0 + r, c0 + lane)] = g[r];
                }
            }
            __syncthreads();
            if (ty == 0 && rowt && tx >= c0 + NB) {                     // panel rows: L(i, c0..) = A'(i, c0..) G^T
                double* Pr = sL + PK(tx, c0);
                double ev[NB], pv[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) ev[k] = Pr[k];
#pragma unroll
                for (int k = 0; k < NB; k++) { double a = 0.0;
#pragma unroll
                    for (int m = 0; m <= k; m++) a += ev[m] * sL[PK(c0 + k, c0 + m)];
                    pv[k] = a; }
#pragma unroll
                for (int k = 0; k < NB; k++) Pr[k] = pv[k];
            }
            if (ty == 1 && tx < NR) {                                   // w block = G t block
                double ev[NB], pv[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) ev[k] = sT[tx * Q + c0 + k];
#pragma unroll
                for (int k = 0; k < NB; k++) { double a = 0.0;
#pragma unroll
                    for (int m = 0; m <= k; m++) a += ev[m] * sL[PK(c0 + k, c0 + m)];
                    pv[k] = a; }
#pragma unroll
                for (int k = 0; k < NB; k++) sT[tx * Q + c0 + k] = pv[k];
            }
            __syncthreads();
            if (rowt && tx >= c0 + NB) {
                const double* Pr = sL + PK(tx, c0);
                double pv[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) pv[k] = Pr[k];
                for (int cp = c0 + NB + ty; cp <= tx; cp += 8) {
                    const double* Lr = sL + PK(cp, c0);
                    double v = sL[PK(tx, cp)];
#pragma unroll
                    for (int k = 0; k < NB; k++) v -= pv[k] * Lr[k];
                    sL[PK(tx, cp)] = v;
                }
                if (ty < NR) {
                    double v = sT[ty * Q + tx];
#pragma unroll
                    for (int k = 0; k < NB; k++) v -= pv[k] * sT[ty * Q + c0 + k];
                    sT[ty * Q + tx] = v;
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < NP; e += nt) Lbuf[(size_t)s * NP + e] = sL[e];
        for (int e = tid; e < NR * Q; e += nt) { const double wv = sT[e]; sW[e] = wv; wbuf[(size_t)s * NR * Q + e] = wv; }
        __syncthreads();
    }
    // ---- backward
    for (int j = ns - 1; j >= 0; j--) {
        const int s = s0 + j, p0 = sep_lo[s];
        if (j < ns - 1) {                                   // reload this separator's factor and w; v = w - F_{j+1}^T x_{j+1}
            for (int e = tid; e < NP; e += nt) sL[e] = Lbuf[(size_t)s * NP + e];
            const double* Fn = Fbuf + (size_t)(s + 1) * Q * Q;
            for (int o = wave; o < NR * Q; o += nw) {
                const int r = (o >= Q) ? o / Q : 0, c = o - r * Q;
                double acc = 0.0;
                for (int i = lane; i < Q; i += 64) acc += Fn[(size_t)c * Q + i] * sW[r * Q + i];
                acc = wave_sum(acc);
                if (lane == 0) sT[o] = wbuf[(size_t)s * NR * Q + o] - acc;
            }
        } else {
            for (int e = tid; e < NR * Q; e += nt) sT[e] = sW[e];
        }
        // x = Lc^-T v, blocked from the last block row
        for (int c0 = Q - NB; c0 >= 0; c0 -= NB) {
            __syncthreads();
            double xv = 0.0;
            const int br = tid / NB, bk = tid - br * NB;                 // (right-hand side, row in block) for tid < NR*NB
            if (tid < NR * NB) {
#pragma unroll
                for (int m = 0; m < NB; m++) if (m >= bk) xv += sL[PK(c0 + m, c0 + bk)] * sT[br * Q + c0 + m];       // x = G^T v
            }
            __syncthreads();
            if (tid < NR * NB) sT[br * Q + c0 + bk] = xv;
            __syncthreads();
            if (ty < NR && tx < c0) {
                double v = sT[ty * Q + tx];
#pragma unroll
                for (int k = 0; k < NB; k++) v -= sL[PK(c0 + k, tx)] * sT[ty * Q + c0 + k];
                sT[ty * Q + tx] = v;
            }
        }
        __syncthreads();
        for (int e = tid; e < NR * Q; e += nt) {
            const int r = (e >= Q) ? e / Q : 0, c = e - r * Q;
            const double x = sT[e];
            sW[e] = x; Y[(size_t)r * n + (size_t)p0 * DC + c] = x;
        }
        __syncthreads();
    }
#undef PK
}

// ---- 4b. the same chain on the matrix cores ------------------------------------------------------------------------------------------
// The three dense phases of a separator -- F = E Lc^-T (Q^3 flop), D -= F F^T (Q^3) and chol(D) (Q^3 / 3), Q = b DC <= 114 -- as
// 16x16 tiles of v_mfma_f64_16x16x4_f64 (one wave per tile, operands straight from LDS: A(i, k) on lane (i = l & 15, k = l >> 4), B(k, j)
// likewise, four accumulators per lane at rows (l >> 4) + 4 q, column l & 15).  A tile product of 16 columns is four instructions of
// 1024 multiply-adds each instead of ~4000 LDS reads feeding scalar FMAs, which is what bounded the VALU kernel above (profiles/r01_notes.md).
// The blocked algorithms take 16 columns per step instead of DC:
//   * F = E Lc^-T: rows of E are independent, so wave I owns row tile I and walks the block columns J = 0.. on its own -- F(I, J) =
//     (E(I, J) - sum_{K<J} F(I, K) Lc(J, K)^T) G_J^T with G_J = Lc(J, J)^-1 -- without a single workgroup barrier;
//   * D -= F F^T: the 36 lower tiles over the 16 waves, K = Q in steps of 4;
//   * chol(D): per block column J one wave factors AND inverts the 16x16 diagonal block (wave_chol_inverse16: lane per row, v_readlane
//     broadcasts), panels L(I, J) = A(I, J) G_J^T and trailing tiles A(I, K) -= L(I, J) L(K, J)^T on the matrix cores; 3 barriers per
//     block column (24 per separator instead of 57); the forward substitution of the right-hand sides rides along.
// Q is not a multiple of 16: the last tile is partial.  Rows / columns beyond Q are never stored; operand rows beyond Q only feed such
// outputs and are left as they are (they stay inside the LDS allocation), the K index is masked where it can run past Q.  The diagonal
// blocks of the packed triangle hold G_J (16x16) here, so the substitutions work on 16-blocks too.  Same arguments and buffers as
// k_sub_sep_chain (SSFM_CHAIN_MFMA=0 selects that one).

// lane r < 16 enters with row r of an SPD 16x16 block; on exit lane c holds column c of G = L^-1 (g[r] = G[r][c], zero above the diagonal)
__device__ __forceinline__ bool wave_chol_inverse16(double (&row)[16], double (&g)[16]) {
    const int lane = threadIdx.x & 63;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const double d = lane_bcast(row[c], c);
        ok = ok && (d > 0.0);                                // (off the dependent chain, as in wave_chol_inverse: band_kernels2.h)
        const double l = row[c] * fast_rsqrt(d);             // L[r][c] on lane r (meaningful for r >= c); replaces row[c]
        row[c] = l;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; c2++) row[c2] -= l * lane_bcast(l, c2);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {                           // G[r][c] = ( [r == c] - sum_{k<r} L[r][k] G[k][c] ) / L[r][r]  on lane c
        double acc = (lane == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; k++) acc -= lane_bcast(row[k], r) * g[k];
        g[r] = acc * fast_rcp(lane_bcast(row[r], r));
    }
    return ok;
}

// The same factor-and-invert with the eliminations on the matrix cores (round 4).  The lane-per-row routine above is ~1150 dependent instructions of ONE wave
// (two v_readlane + one FMA per updated entry) = 11k cycles, 41-43 % of a separator step.  Here the block lives in the ACCUMULATOR layout of
// v_mfma_f64_16x16x4 (lane (li, lk) = (l & 15, l >> 4) holds S[lk + 4 q][li], q = 0..3) and column c is eliminated by ONE instruction: the operand
// layouts are A(i, k) on lane (i, k) and B(k, j) on lane (j, k), so the lanes with lk == c % 4 already hold S[c][li] = S[li][c] in register c / 4 -- the pivot
// column is an operand WITHOUT any cross-lane move (k-slice c % 4 carries it, the other three slices are zero).  Square-root free: S -= (a / d) a^T.  The same
// multipliers applied to W (starts as I; row c of W sits on the same lanes, same register) leave W = L~^-1 of S = L~ D L~^T, so the inverse needs no
// substitution pass: G = (L~ D^1/2)^-1 = D^-1/2 W.  Per column: two v_readlane (the pivot), one reciprocal, two selects, one multiply, two matrix
// instructions.  Returns G in the accumulator layout (g[q] = G[lk + 4 q][li], zero above the diagonal); false if a pivot is not positive.
__device__ __forceinline__ bool wave_ldl_inverse16_mfma(v4d_t& S, v4d_t& G) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    v4d_t W;
#pragma unroll
    for (int q = 0; q < 4; q++) W[q] = (lk + 4 * q == li) ? 1.0 : 0.0;
    double dmine[4] = {1.0, 1.0, 1.0, 1.0};                 // pivots of the rows this lane holds (lk + 4 q)
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const double d = lane_bcast(S[c >> 2], c + 16 * (c & 3));          // S[c][c]: lane (li = c, lk = c % 4), register c / 4
        ok = ok && (d > 0.0);
        if (lk == (c & 3)) dmine[c >> 2] = d;
        if (c == 15) break;
        const bool src = lk == (c & 3);
        const double a = (src && li > c) ? S[c >> 2] : 0.0;               // a_i = S[i][c], rows below the pivot
        const double w = src ? W[c >> 2] : 0.0;                           // row c of W (columns <= c are the only non-zeros)
        const double m = -a * fast_rcp(d);
        S = __builtin_amdgcn_mfma_f64_16x16x4f64(m, a, S, 0, 0, 0);
        W = __builtin_amdgcn_mfma_f64_16x16x4f64(m, w, W, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) G[q] = W[q] * fast_rsqrt(dmine[q]);
    return ok;
}

// NTHR = 1024 or 512 threads: at 1024 the kernel sits on the 128-register limit of four waves per SIMD (35 scratch accesses); 512 threads have 256 registers each
template <int DC, int NR, bool CHAIN_DIAG_MFMA = true, bool CHAIN_PREFETCH = true, int NTHR = 1024, bool CHAIN_STAMPED = false>
__global__ void __launch_bounds__(NTHR)
k_sub_sep_chain_mfma(const double* __restrict__ Z, const double* __restrict__ Dd, const double* __restrict__ tt, const int* __restrict__ chain_ptr,
                     const int* __restrict__ sep_lo, int N, int b, double* __restrict__ Fbuf, double* __restrict__ Lbuf, double* __restrict__ wbuf,
                     double* __restrict__ Y, int* __restrict__ fail_flag, long long* __restrict__ stamps /* null, or [16] phase stamps of chain 0, separator 1 */,
                     int twist, int nsep_total, double* __restrict__ Cbuf, double* __restrict__ Tcbuf, int* __restrict__ flags, int seq, int pingpong = 0) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int TB = 16, NTY = NTHR / 128, PFG = (NTHR - 64) / 128, PFN = (16 + PFG - 1) / PFG;     // row groups of 128 threads; groups / values per thread of the prefetching waves
#define STAMP(k_) do { if (CHAIN_STAMPED && stamps && blockIdx.x == 0 && j == 1 && tid == 0) stamps[k_] = (long long)__builtin_amdgcn_s_memtime(); } while (0)   /* compiled out of the product build: the kernel is at its register limit */
    const int Q = b * DC, n = N * DC, tid = threadIdx.x, nt = NTHR, NP = Q * (Q + 1) / 2, TQ = (Q + TB - 1) / TB;
    // pingpong (the host sets it when a second triangle fits the LDS, Q <= 96): two triangles take turns -- the factor of separator j - 1 (read by the F solve of
    // separator j) in one, D_j in the other, and D_{j+1} is brought into the first while separator j is factored: the load of D leaves the dependent chain
    const bool pp = CHAIN_PREFETCH && pingpong != 0;
    double* sL = lds;                  // [NP]   packed lower triangle, row-major; the 16x16 diagonal blocks hold G_J = L_JJ^-1
    double* sLp = pp ? lds + NP : lds; // the triangle of the previous separator's factor (the same one without pingpong)
    double* sF = lds + (pp ? 2 : 1) * NP;   // [Q][Q] column-major: F(i, c) at c*Q + i
    double* sT = sF + (size_t)Q * Q;   // [NR][Q]
    double* sW = sT + NR * Q;          // [NR][Q]
    // Two-sided elimination of a chain (twist): workgroup 2c takes the separators from the front up to and including the middle one,
    // workgroup 2c + 1 the ones behind it from the BACK (coupling blocks read transposed) and ends with a virtual step on the middle
    // separator that only produces its Schur contribution (-F'F'^T, -F'w); the front workgroup adds it before it factors the middle,
    // publishes x_middle, and both substitute backwards on their own halves.  Depth ceil(ns / 2) + 1 instead of ns.  The two workgroups of
    // a chain meet through two flags in global memory (release: __threadfence + barrier + atomic store; acquire: spin + fence).
    const int chain = twist ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, side = twist ? (int)(blockIdx.x & 1) : 0;
    const int s0 = chain_ptr[chain], ns = chain_ptr[chain + 1] - s0;
    const bool tw = twist && ns >= 3;
    if (side == 1 && !tw) return;
    const int mid = tw ? ns / 2 : ns - 1;                          // position of the front side's last separator
    const int npos = (side == 0) ? mid + 1 : ns - mid;              // back side: ns - 1 - mid real separators + the virtual step
    int* flag_contrib = flags + 2 * chain; int* flag_x = flags + 2 * chain + 1;
    auto publish_flag = [&](int* f) { __threadfence(); __syncthreads(); if (tid == 0) __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); };
    auto await_flag = [&](int* f) {
        if (tid == 0) while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(16);
        __syncthreads(); __threadfence();
    };
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int tx = tid & 127, ty = tid >> 7;
    const bool rowt = tx < Q;
#define PK(i_, c_) ((i_) * ((i_) + 1) / 2 + (c_))
#define MFMA64(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)
    for (int j = 0; j < npos; j++) {
        const int s = (side == 0) ? s0 + j : s0 + ns - 1 - j, p0 = sep_lo[s];
        const bool virt = side == 1 && j == npos - 1;               // the back side's step on the middle separator: contribution only
        const size_t fslot = virt ? (size_t)(nsep_total + chain) : (size_t)s;
        STAMP(0);
        for (int e = tid; e < NR * Q; e += nt) sT[e] = virt ? 0.0 : tt[(size_t)s * NR * Q + e];
        if (j > 0) {
            if (!CHAIN_PREFETCH) {                                  // (else: the waves that idle during the previous separator's diagonal blocks brought E in, below)
                if (side == 0) { if (rowt) for (int c = ty; c < Q; c += NTY) sF[c * Q + tx] = Z[(size_t)c * n + (size_t)p0 * DC + tx]; }
                else {                                              // E' = E_{s+1}^T: rows = this separator, columns = the one behind it (s + 1)
                    const int pn = sep_lo[s + 1];
                    if (rowt) for (int r = ty; r < Q; r += NTY) sF[tx * Q + r] = Z[(size_t)r * n + (size_t)pn * DC + tx];
                }
            }
            __syncthreads();
            STAMP(1);
            // ---- F = E Lc^-T: wave I = row tile I, block columns in order, no barrier.
            // Round 4: the tile is accumulated TRANSPOSED (T^T = E^T - Lc(J, K) F(I, K)^T: the two operands of the round-2 product swapped), because the accumulator
            // layout of T^T (lane (li, lk), register q = T[r0 + li][c0 + lk + 4 q]) IS the A-operand layout of the product T G_J^T that follows (k-slice q): T never goes
            // through LDS (it did: four read-modify-writes, a fence and four reads per block column on the dependent chain), and E(I, J), G_J and the operands of
            // K = 0 are all in flight before the first product.  24.4k -> see profiles/r04_notes.md cycles at Q = 84.
            if (wave < TQ) {
                const int I = wave, r0 = TB * I;
                for (int J = 0; J < TQ; J++) {
                    const int c0 = TB * J;
                    v4d_t Tt; double gop[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int k = lk + 4 * q, col = c0 + k;
                        Tt[q] = (col < Q) ? sF[col * Q + r0 + li] : 0.0;                                            // E(I, J)^T
                        gop[q] = (k <= li && c0 + li < Q) ? sLp[PK(c0 + li, col)] : 0.0;                           // G_J[li][k]
                    }
                    {
                        double fa[4], fl[4];
                        const double* Lrow = sLp + PK(c0 + li, 0);
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) { const int kc = 4 * kk + lk; fa[kk] = sF[kc * Q + r0 + li]; fl[kk] = -Lrow[kc]; }   // (J == 0: read, not used)
                        for (int K = 0; K < J; K++) {
                            double ca[4], cl[4];
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) { ca[kk] = fa[kk]; cl[kk] = fl[kk]; }
                            if (K + 1 < J) {
#pragma unroll
                                for (int kk = 0; kk < 4; kk++) { const int kc = TB * (K + 1) + 4 * kk + lk; fa[kk] = sF[kc * Q + r0 + li]; fl[kk] = -Lrow[kc]; }     // kc < c0 <= Q - 1: always a real column
                            }
                            // (Lc(J, K) F(I, K)^T)[i][j] lands at lane (j, .), register i: transposed
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) Tt = MFMA64(cl[kk], ca[kk], Tt);
                        }
                    }
                    v4d_t f = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) f = MFMA64(Tt[kk], gop[kk], f);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int row = r0 + lk + 4 * q, col = c0 + li;
                        if (row < Q && col < Q) sF[col * Q + row] = f[q];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            __syncthreads();
            STAMP(2);
            // ---- t_j -= F w_{j-1};  F to global for the backward pass
            if (CHAIN_PREFETCH) {                                   // four lanes per (row, right-hand side): Q / 4 terms each and two xor-shuffles (one thread per row: Q dependent terms)
                const int quad = tid >> 2, part = tid & 3;
                for (int task = quad; task < NR * Q; task += NTHR / 4) {
                    const int r = (task >= Q) ? task / Q : 0, row = task - r * Q;
                    double acc = 0.0;
                    for (int c = part; c < Q; c += 4) acc += sF[c * Q + row] * sW[r * Q + c];
                    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
                    if (part == 0) sT[r * Q + row] -= acc;
                }
            } else if (rowt && ty < NR) {
                double acc = 0.0;
                for (int c = 0; c < Q; c++) acc += sF[c * Q + tx] * sW[ty * Q + c];
                sT[ty * Q + tx] -= acc;
            }
            for (int e = tid; e < Q * Q; e += nt) Fbuf[fslot * Q * Q + e] = sF[e];
        }
        // ---- D_j into the packed triangle (the previous factor is dead: it went to Lbuf, and its last reader -- the F solve -- is behind a barrier): in the same
        // phase as the t update and the store of F, no barrier between them
        STAMP(3);
        if (!(pp && j > 0)) { if (rowt) for (int cp = ty; cp <= tx; cp += NTY) sL[PK(tx, cp)] = virt ? 0.0 : Dd[((size_t)s * Q + tx) * Q + cp]; }    // (else: prefetched during the previous separator)
        __syncthreads();
        STAMP(4);
        if (j > 0) {
            // ---- D_j -= F F^T: lower tiles over the waves
            const int ntile = TQ * (TQ + 1) / 2;
            for (int t = wave; t < ntile; t += nw) {
                int I = 0; while ((I + 1) * (I + 2) / 2 <= t) I++;
                const int J = t - I * (I + 1) / 2, r0 = TB * I, c0 = TB * J;
                v4d_t acc = {0.0, 0.0, 0.0, 0.0};      // (a second accumulator for alternate products changes nothing: the phase is bound by the matrix pipe of the busiest SIMD, 6 of 21 tiles, and by 6.2 cycles per ds_read_b64 -- scripts/lab/syrk_lab.hip: 8.7k MFMA only, 5.4k reads only, 10.9k together at Q = 84)
                int k0 = 0;
                for (; k0 + 16 <= Q; k0 += 16) {                    // four k-steps per trip: the eight operand reads first, then the four products
                    double a4[4], b4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int kc = k0 + 4 * u + lk; a4[u] = sF[kc * Q + r0 + li]; b4[u] = sF[kc * Q + c0 + li]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) acc = MFMA64(a4[u], b4[u], acc);
                }
                for (; k0 < Q; k0 += 4) {
                    const int kc = k0 + lk; const bool in = kc < Q; const int kcc = in ? kc : Q - 1;
                    const double a = in ? sF[kcc * Q + r0 + li] : 0.0, bb = in ? sF[kcc * Q + c0 + li] : 0.0;
                    acc = MFMA64(a, bb, acc);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = r0 + lk + 4 * q, col = c0 + li;
                    if (row < Q && col <= row) sL[PK(row, col)] -= acc[q];
                }
            }
        }
        if (virt) {                                                 // -F'F'^T and -F'w of the back side go to the front side; no factorisation
            __syncthreads();
            for (int e = tid; e < NP; e += nt) Cbuf[(size_t)chain * NP + e] = sL[e];
            for (int e = tid; e < NR * Q; e += nt) Tcbuf[(size_t)chain * NR * Q + e] = sT[e];
            publish_flag(flag_contrib);
            break;
        }
        if (tw && side == 0 && j == mid) {                          // the middle separator: add the back side's contribution
            await_flag(flag_contrib);
            for (int e = tid; e < NP; e += nt) sL[e] += Cbuf[(size_t)chain * NP + e];
            for (int e = tid; e < NR * Q; e += nt) sT[e] += Tcbuf[(size_t)chain * NR * Q + e];
        }
        // ---- blocked Cholesky (16 columns per step), forward substitution of t riding along
        for (int J = 0; J < TQ; J++) {
            const int c0 = TB * J, nJ = min(TB, Q - c0);
            __syncthreads();
            if (J == 0) STAMP(5);
            if (J == 1) STAMP(9);
            if (wave == 0) {
                if (CHAIN_DIAG_MFMA) {                                      // eliminations on the matrix cores: 3.7k cycles against 10.9k (scripts/lab/ldl16_lab.hip)
                    v4d_t Sd, Gd;
#pragma unroll
                    for (int q = 0; q < 4; q++) { const int r = lk + 4 * q; Sd[q] = (r < nJ && li < nJ) ? sL[PK(c0 + max(r, li), c0 + min(r, li))] : ((r == li) ? 1.0 : 0.0); }
                    if (!wave_ldl_inverse16_mfma(Sd, Gd) && lane == 0) *fail_flag = 1;
#pragma unroll
                    for (int q = 0; q < 4; q++) { const int r = lk + 4 * q; if (r >= li && r < nJ) sL[PK(c0 + r, c0 + li)] = Gd[q]; }
                } else {
                    double row[16], g[16];
#pragma unroll
                    for (int c = 0; c < 16; c++) row[c] = (li < nJ && c < nJ && lane < 16) ? sL[PK(c0 + max(li, c), c0 + min(li, c))] : ((lane == c) ? 1.0 : 0.0);
                    if (!wave_chol_inverse16(row, g) && lane == 0) *fail_flag = 1;
                    if (lane < nJ) {
#pragma unroll
                        for (int r = 0; r < 16; r++) if (r >= lane && r < nJ) sL[PK(c0 + r, c0 + lane)] = g[r];
                    }
                }
            } else if (CHAIN_PREFETCH) {
                // the other fifteen waves have nothing to do until the diagonal block is there.
                // (2) slice J (columns == J mod TQ) of the NEXT separator's coupling block E into sF, which is dead from the rank-Q update of this separator to the
                // F solve of the next (10 of 116 k cycles per step), and (3) with two triangles the same slice of its D: the global loads are issued FIRST, the
                // right-hand-side update (1) runs under their latency, the LDS stores come last
                const int t2 = tid - 64, tx2 = t2 & 127, ty2 = t2 >> 7;                       // PFG full groups of 128 rows
                const bool pre = j + 1 < npos && t2 < 128 * PFG && tx2 < Q;
                const bool vnext = side == 1 && j + 1 == npos - 1;
                double ev[PFN], dv[PFN];
#pragma unroll
                for (int u = 0; u < PFN; u++) { ev[u] = 0.0; dv[u] = 0.0; }                     // <= ceil(16 / 7) columns of a slice per thread (Q <= 114: a slice has <= 16 columns)
                if (pre) {
                    const size_t pnx = (size_t)sep_lo[side == 0 ? s + 1 : s] * DC;             // back side: the next step's s is s - 1 and the separator behind it is this one
#pragma unroll
                    for (int u = 0; u < PFN; u++) { const int c = J + TQ * (ty2 + PFG * u); if (c < Q) ev[u] = Z[(size_t)c * n + pnx + tx2]; }
                    if (pp && !vnext) {
                        const double* Dn = Dd + (size_t)((side == 0) ? s + 1 : s - 1) * Q * Q + (size_t)tx2 * Q;
#pragma unroll
                        for (int u = 0; u < PFN; u++) { const int cp = J + TQ * (ty2 + PFG * u); if (cp <= tx2) dv[u] = Dn[cp]; }
                    }
                }
                // (1) t(rows below block J - 1) -= L(rows, block J - 1) w_{J-1}: it was the long pole of the trailing phase (one thread per row, sixteen dependent terms, on
                // waves that also had a tile); here four lanes share a row (four terms each, two xor-shuffles) and the block column J's w is only needed behind the next barrier
                if (J > 0) {
                    const int quad = t2 >> 2, part = t2 & 3, nrow = Q - c0, cp = c0 - TB;
                    for (int task = quad; task < NR * nrow; task += (NTHR - 64) / 4) {
                        const int r = (task >= nrow) ? task / nrow : 0, row = c0 + task - r * nrow;
                        const double* Pr = sL + PK(row, cp + 4 * part); const double* wv = sT + r * Q + cp + 4 * part;
                        double v = Pr[0] * wv[0] + Pr[1] * wv[1] + Pr[2] * wv[2] + Pr[3] * wv[3];
                        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
                        if (part == 0) sT[r * Q + row] -= v;
                    }
                }
                if (pre) {
#pragma unroll
                    for (int u = 0; u < PFN; u++) {
                        const int c = J + TQ * (ty2 + PFG * u);
                        if (c < Q) { if (side == 0) sF[c * Q + tx2] = ev[u]; else sF[tx2 * Q + c] = ev[u]; }
                        if (pp && c <= tx2) sLp[PK(tx2, c)] = dv[u];
                    }
                }
            }
            __syncthreads();
            if (J == 0) STAMP(6);
            const int below = TQ - 1 - J;                                   // row tiles under the diagonal block
            if (wave < below) {                                             // panel: L(I, J) = A(I, J) G_J^T
                const int r0 = TB * (J + 1 + wave);
                v4d_t pacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int k = 4 * kk + lk, kc = c0 + k;
                    const double a = (k < nJ) ? sL[PK(r0 + li, kc)] : 0.0;
                    const double g = (k <= li && li < nJ) ? sL[PK(c0 + li, kc)] : 0.0;
                    pacc = MFMA64(a, g, pacc);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = r0 + lk + 4 * q;
                    if (row < Q && li < nJ) sL[PK(row, c0 + li)] = pacc[q];
                }
            } else if (wave == nw - 1 && lane < NR * TB) {                  // w block = G_J t block
                const int r = lane / TB, k = lane - r * TB;
                double acc = 0.0;
                if (k < nJ) for (int m = 0; m <= k; m++) acc += sL[PK(c0 + k, c0 + m)] * sT[r * Q + c0 + m];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (k < nJ) sT[r * Q + c0 + k] = acc;
            }
            __syncthreads();
            if (J == 0) STAMP(7);
            if (below > 0) {                                                // trailing tiles (I, K), J < K <= I
                const int ntile = below * (below + 1) / 2;
                for (int t = wave; t < ntile; t += nw) {
                    int a = 0; while ((a + 1) * (a + 2) / 2 <= t) a++;
                    const int r0 = TB * (J + 1 + a), k0r = TB * (J + 1 + (t - a * (a + 1) / 2));
                    v4d_t u = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        const int kc = c0 + 4 * kk + lk;                    // nJ = 16 whenever a tile lies below this block
                        u = MFMA64(sL[PK(r0 + li, kc)], sL[PK(k0r + li, kc)], u);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int row = r0 + lk + 4 * q, col = k0r + li;
                        if (row < Q && col <= row) sL[PK(row, col)] -= u[q];
                    }
                }
                if (!CHAIN_PREFETCH && rowt && tx >= c0 + TB && ty >= NTY - NR) {    // t(rows below) -= L(rows, block) w block   (threads of the last waves)
                    const int r = ty - (NTY - NR);
                    double v = sT[r * Q + tx];
                    const double* Pr = sL + PK(tx, c0);
#pragma unroll
                    for (int k = 0; k < TB; k++) v -= Pr[k] * sT[r * Q + c0 + k];
                    sT[r * Q + tx] = v;
                }
            }
        }
        __syncthreads();
        STAMP(8);
        for (int e = tid; e < NP; e += nt) Lbuf[(size_t)s * NP + e] = sL[e];
        for (int e = tid; e < NR * Q; e += nt) { const double wv = sT[e]; sW[e] = wv; wbuf[(size_t)s * NR * Q + e] = wv; }
        __syncthreads();
        STAMP(10);
        if (pp && j + 1 < npos) { double* t_ = sL; sL = sLp; sLp = t_; }    // the next separator's D is in the other triangle; this factor becomes "the previous one"
    }
#undef STAMP
    // ---- backward
    if (side == 1) {                                        // x of the middle separator, from the front side
        await_flag(flag_x);
        const int pm = sep_lo[s0 + mid];
        for (int e = tid; e < NR * Q; e += nt) { const int r = (e >= Q) ? e / Q : 0, c = e - r * Q; sW[e] = Y[(size_t)r * n + (size_t)pm * DC + c]; }
        __syncthreads();
    }
    const int jlast = (side == 0) ? npos - 1 : npos - 2;    // last REAL separator of this side (the back side's virtual step has no unknowns)
    for (int j = jlast; j >= 0; j--) {
        const int s = (side == 0) ? s0 + j : s0 + ns - 1 - j, p0 = sep_lo[s];
        if (side == 1 || j < jlast) {                       // reload this separator's factor and w; v = w - F_next^T x_next (next = the step after this one)
            for (int e = tid; e < NP; e += nt) sL[e] = Lbuf[(size_t)s * NP + e];
            const size_t nslot = (side == 0) ? (size_t)(s + 1) : ((j + 1 == npos - 1) ? (size_t)(nsep_total + chain) : (size_t)(s - 1));
            const double* Fn = Fbuf + nslot * Q * Q;
            for (int o = wave; o < NR * Q; o += nw) {
                const int r = (o >= Q) ? o / Q : 0, c = o - r * Q;
                double acc = 0.0;
                for (int i = lane; i < Q; i += 64) acc += Fn[(size_t)c * Q + i] * sW[r * Q + i];
                acc = wave_sum(acc);
                if (lane == 0) sT[o] = wbuf[(size_t)s * NR * Q + o] - acc;
            }
        } else {
            for (int e = tid; e < NR * Q; e += nt) sT[e] = sW[e];
        }
        // x = Lc^-T v, 16 rows per step from the last block
        for (int J = TQ - 1; J >= 0; J--) {
            const int c0 = TB * J, nJ = min(TB, Q - c0);
            __syncthreads();
            double xv = 0.0;
            const int br = tid / TB, bk = tid - br * TB;
            if (tid < NR * TB && bk < nJ) for (int m = bk; m < nJ; m++) xv += sL[PK(c0 + m, c0 + bk)] * sT[br * Q + c0 + m];       // x = G^T v
            __syncthreads();
            if (tid < NR * TB && bk < nJ) sT[br * Q + c0 + bk] = xv;
            __syncthreads();
            if (ty < NR && tx < c0) {
                double v = sT[ty * Q + tx];
                for (int k = 0; k < nJ; k++) v -= sL[PK(c0 + k, tx)] * sT[ty * Q + c0 + k];
                sT[ty * Q + tx] = v;
            }
        }
        __syncthreads();
        for (int e = tid; e < NR * Q; e += nt) {
            const int r = (e >= Q) ? e / Q : 0, c = e - r * Q;
            const double x = sT[e];
            sW[e] = x; Y[(size_t)r * n + (size_t)p0 * DC + c] = x;
        }
        __syncthreads();
        if (tw && side == 0 && j == mid) publish_flag(flag_x);     // the back side may start its substitution
    }
#undef PK
#undef MFMA64
}

// ---- 5. y(seg) -= Z x(separator in front) ------------------------------------------------------------------------------------------
// r05ah: 64 rows per workgroup, four threads per row (a quarter of the Q columns each, partial sums through LDS in slice order): one thread per row walked Q loads
// one after the other -- 11.4 us at Q = 78 for 150 rows per arc.  LDS: NR * Q doubles (x of the separator) + 4 * 64 * NR (partial sums).
constexpr int APPLY_ROWS = 64, APPLY_SLICES = 4;
template <int DC, int NR>
__global__ void __launch_bounds__(APPLY_ROWS * APPLY_SLICES)
k_sub_apply_left(const double* __restrict__ Z, double* __restrict__ Y, const int* __restrict__ seg_lo, const int* __restrict__ seg_hi,
                 const int* __restrict__ left_segs, int N, int b) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Q = b * DC, n = N * DC, tid = threadIdx.x;
    const int seg = left_segs[blockIdx.x], r0 = seg_lo[seg], r1 = seg_hi[seg];
    if (r0 * DC + (int)blockIdx.y * APPLY_ROWS >= r1 * DC) return;
    const int row = tid & (APPLY_ROWS - 1), slice = tid / APPLY_ROWS;
    const int kk = r0 * DC + blockIdx.y * APPLY_ROWS + row;
    double* part = lds + NR * Q;                       // [APPLY_SLICES][APPLY_ROWS][NR]
    for (int e = tid; e < NR * Q; e += APPLY_ROWS * APPLY_SLICES) { const int r = e / Q, q = e - r * Q; lds[e] = Y[(size_t)r * n + (size_t)(r0 - b) * DC + q]; }
    __syncthreads();
    const int qper = (Q + APPLY_SLICES - 1) / APPLY_SLICES, q0 = slice * qper, q1 = min(Q, q0 + qper);
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = 0.0;
    if (kk < r1 * DC) {
#pragma unroll 4
        for (int q = q0; q < q1; q++) {
            const double zv = Z[(size_t)q * n + kk];
#pragma unroll
            for (int r = 0; r < NR; r++) acc[r] += zv * lds[r * Q + q];
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) part[(slice * APPLY_ROWS + row) * NR + r] = acc[r];
    __syncthreads();
    if (slice == 0 && kk < r1 * DC) {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            double a = 0.0;
#pragma unroll
            for (int sl = 0; sl < APPLY_SLICES; sl++) a += part[(sl * APPLY_ROWS + row) * NR + r];
            Y[(size_t)r * n + kk] -= a;
        }
    }
}

}  // namespace ssfm
