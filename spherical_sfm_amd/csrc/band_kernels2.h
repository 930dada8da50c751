// spherical_sfm_amd -- second-generation block-banded Cholesky kernels (the reduced camera system).
//
// The factorisation of a camera ring is a chain of N dependent steps with a few kflop each, so what matters is the number
// of instructions (and barriers) on the dependent path of one step, not flops:
//   * the diagonal block is factored by wave 0 with one lane per ROW and v_readlane broadcasts (no LDS round trips, ~125
//     instructions instead of ~250 on a single lane), and its inverse G = L^-1 follows with one lane per column;
//   * the panel is a multiplication by G^T (independent 6-term dot products) instead of a forward substitution (a chain);
//   * the panel lands in its own LDS buffer, which frees the slot of the leaving row at once: the prefetched row is stored
//     during the same step and a step has TWO barriers (panel | trailing update + look-ahead factorisation);
//   * only G and the off-diagonal blocks of L are ever stored; nothing downstream needs L_jj itself.
// The back substitution runs on ONE wave per (component, right-hand side), right-looking: once x_j is known every pending
// row sum takes its L(j,i)^T x_j term at once, the sums shift by one block row per step through ds_bpermute, and no barrier
// is needed at all.
#pragma once
#include <type_traits>
#include "band_kernels.h"

namespace ssfm {

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {            // src_lane: compile-time constant after unrolling
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_shift_down(double v, int delta) {          // value of lane (l + delta); garbage when out of range
    const int idx = (int)(((threadIdx.x & 63) + delta) & 63) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Wave-level factorisation of one DCxDC block.  Lane r (< DC) enters with row r of the SPD block in row[]; on exit lane c
// holds column c of G = L^-1 in g[] (g[r] = G[r][c], zero above the diagonal).  Returns false if a pivot is not positive.
template <int DC>
__device__ __forceinline__ bool wave_chol_inverse(double (&row)[DC], double (&g)[DC]) {
    const int lane = threadIdx.x & 63;
    double L[DC], rs[DC]; bool ok = true;
#pragma unroll
    for (int c = 0; c < DC; c++) {
        double d = lane_bcast(row[c], c);
        // round 4: the pivot test no longer feeds the chain (it used to replace a bad pivot by 1.0: two selects in front of every reciprocal square root); a
        // non-positive or NaN pivot makes the factor NaN / inf, the caller raises the failure flag and the LM loop discards the step, as before
        ok = ok && (d > 0.0);
        rs[c] = fast_rsqrt(d);                           // (a third-order single step, fast_rsqrt3, saves two instructions per column and measured 0.4 %: not worth a change of rounding --
                                                         //  the 2000-node pose graph of tests/test_rotavg_gpu.py sits on a chaotic path where that moves an accept / reject decision)
        const double l = row[c] * rs[c];                 // L[r][c] on lane r (meaningful for r >= c)
        L[c] = l;
#pragma unroll
        for (int c2 = c + 1; c2 < DC; c2++) row[c2] -= l * lane_bcast(l, c2);      // a[r][c2] -= L[r][c] L[c2][c]
    }
#pragma unroll
    for (int r = 0; r < DC; r++) {                       // G[r][c] = rs_r ( [r == c] - sum_{k<r} L[r][k] G[k][c] ) on lane c
        double acc = (lane == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; k++) acc -= lane_bcast(L[k], r) * g[k];
        g[r] = acc * rs[r];
    }
    return ok;
}

// Right-looking block-band Cholesky with look-ahead, LDS-resident window, one workgroup per connected component.
//   band  [N][b+1][DC*DC]  in/out: block d of row i = (i, i-d); off-diagonal blocks leave as L, diagonal blocks are left alone
//   Ginv  [N][DC*DC]       out: L_jj^-1 (row-major, lower)
//   Y     [NR][N*DC]       in/out: right-hand sides -> L^-1 Y (forward substitution rides along)
// Wave roles (blockDim.x = 64 * (1 + ntw + CHOL2_LOADERS + 1)):
//   wave 0                      look-ahead: next diagonal block, its factor and inverse.  No global memory traffic at all.
//   waves 1..ntw                trailing update of the window + right-hand sides (LDS only)
//   next CHOL2_LOADERS waves    loaders: together they bring in the row that enters the window, one step ahead; they are the
//                               only waves that ever wait on vmcnt
//   last wave                   writer: panel, y_j and G to global memory (stores only, never waited on)
// Every wave takes its share of the panel product in phase B.
constexpr int CHOL2_LOADERS = 2;
// Role of every PHYSICAL wave of the workgroup (wave p runs on SIMD p % 4): v[p] = the role index the kernel's logic uses (0 look-ahead,
// 1..ntw trailing update, then the loaders, the writer last).  With the identity a nine-wave workgroup puts the look-ahead (the dependent chain of
// the step), a trailing wave and the writer on SIMD 0: their phases took 1.6k / 2.2k / 2.1k cycles against 1.3-1.5k for the trailing waves
// that share a SIMD with one loader only, and the step waits for the slowest (s_memtime stamps, scripts/lab/chol_stamps.hip).
struct CholWaveMap { unsigned char v[16]; };
inline CholWaveMap chol_wave_map(int nw, int ntw, int heavy_tw) {      // heavy_tw: trailing waves that carry block tasks (the rest only right-hand sides)
    CholWaveMap m; for (int i = 0; i < 16; i++) m.v[i] = (unsigned char)i;
    // Measured, not derived: the nine-wave workgroup of config 2 (6x6 blocks, half-width 10) gains 18 % per step; with eleven waves (half-width 14,
    // where both loaders then sit with the look-ahead: 286 -> 340 us per launch at the configs[4] size) and with six (merged 3-dof pairs: 2.18 -> 2.20 ms per
    // solve) the identity is better, so only the shape that was stamped is remapped.
    if (nw != 9) return m;
    int weight[16], order[16];
    for (int r = 0; r < nw; r++) {
        weight[r] = (r == 0) ? 12 : (r <= ntw) ? ((r <= heavy_tw) ? 6 : 2) : (r == nw - 1) ? 5 : 4;
        order[r] = r;
    }
    for (int a = 0; a < nw; a++) for (int c = a + 1; c < nw; c++) if (weight[order[c]] > weight[order[a]]) { const int t = order[a]; order[a] = order[c]; order[c] = t; }
    int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0}, cap[4];
    for (int q = 0; q < 4; q++) cap[q] = (nw - q + 3) / 4;                // physical waves q, q + 4, ...
    // SIMD 0: the look-ahead (first in the order) and the LIGHTEST roles for its other waves; everything else heaviest first onto the least
    // loaded of the other three SIMDs
    bool placed[16] = {false};
    m.v[0] = (unsigned char)order[0]; placed[0] = true; used[0] = 1;
    for (int a = nw - 1; a >= 1 && used[0] < cap[0]; a--) { m.v[4 * used[0]] = (unsigned char)order[a]; placed[a] = true; used[0]++; }
    for (int a = 1; a < nw; a++) {
        if (placed[a]) continue;
        int best = -1;
        for (int q = 1; q < 4; q++) if (used[q] < cap[q] && (best < 0 || load[q] < load[best])) best = q;
        m.v[best + 4 * used[best]] = (unsigned char)order[a];
        load[best] += weight[order[a]]; used[best]++;
    }
    return m;
}

// MF = true (DC = 6 only): the panel product and the trailing update run on the matrix cores -- v_mfma_f64_16x16x4_f64 tiles in
// window-relative coordinates (row i of the window = block i / 6 + 1 behind the pivot, scalar row i % 6), K = the pivot's six columns in two
// instructions (the second one half masked).  Operands come straight from the panel in LDS (row-major 6 doubles per scalar row: the 16
// rows of a tile at one k sit in 16 different bank pairs); the four accumulators of a lane go back into the ring of window rows by
// read-modify-write.  Against the 3x3 register tiles of the VALU version (54 LDS operations per 54 multiply-adds and lane, scattered over
// the panel: the phase was LDS-conflict bound at ~80 B/clk) a tile is 8 conflict-free operand reads + 4 read-modify-writes per 1536
// multiply-adds.  SSFM_BAND_MFMA=0 selects the VALU version.
constexpr int BACK_PD = 4;
// One sweep of the back substitution from row re-1 down to r0 by ONE wave: rows >= r1 are given (from y, from the reversed copy at gf, or -- FROM_LDS -- from xs), the others are solved
// and go to y (TO_LDS: to xs).  Shared by k_band_back_v2 and by the epilogue of k_band_chol_v2 that solves a small component in the launch that factored it.
template <int DC, int NS, bool to_lds, bool from_lds>
__device__ __forceinline__ void band_back_sweep(const double* __restrict__ band, const double* __restrict__ Ginv, double* __restrict__ y, double* xs,
                                                const int r0, const int r1, const int re, const int gf, const int b, const int lane) {
    constexpr int BB = DC * DC;
    const int W = b + 1, T = b * DC;
    int dd[NS], off[NS]; bool has[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int t = min(lane + 64 * s, T - 1);                    // clamped: lanes without a task recompute a valid one and are masked
        dd[s] = t / DC + 1; off[s] = dd[s] * BB + (t - (dd[s] - 1) * DC); has[s] = lane + 64 * s < T;
    }
    const int lc = min(lane, DC - 1);
    struct Stage { double col[NS][DC], li[DC], yv; };
    // the row is wave-uniform: a scalar base per row and a 32-bit lane offset per stream (the per-lane 64-bit products of a flat index cost
    // ~16 vector instructions per step of a kernel whose step is bound by the instructions ONE wave can issue)
    const size_t row_stride = (size_t)W * BB;
        Stage st[BACK_PD];
        auto fetch = [&](int j, Stage& s) {    // column a of block (j, j-d) for this lane's tasks; column `lane` of G_j; y_j
            const int jc = max(j, r0);
            const double* __restrict__ rowp = band + (size_t)jc * row_stride;
            const double* __restrict__ gpt = Ginv + (size_t)jc * BB;
#pragma unroll
            for (int m = 0; m < DC; m++) {
#pragma unroll
                for (int q = 0; q < NS; q++) s.col[q][m] = rowp[off[q] + m * DC];
                s.li[m] = gpt[m * DC + lc];                                   // G[m][lane], zero for m < lane
            }
            const int jy = (gf >= 0 && jc >= r1) ? gf + (re - 1 - jc) : jc;
            s.yv = (from_lds && jc >= r1) ? xs[(jy - gf) * DC + lc] : y[(size_t)jy * DC + lc];
        };
#pragma unroll
        for (int u = 0; u < BACK_PD; u++) fetch(re - 1 - u, st[u]);
        double acc[NS];                        // pending sums: task (d, a) = sum over processed k of (L(k, i)^T x_k)[a], i = j-(d-1)
#pragma unroll
        for (int q = 0; q < NS; q++) acc[q] = 0.0;
        for (int jb = re - 1; jb >= r0; jb -= BACK_PD) {
#pragma unroll
            for (int u = 0; u < BACK_PD; u++) {
                const int j = jb - u;
                if (j < r0) break;
                double c[NS][DC], cl[DC]; const double cy = st[u].yv;
                bool v[NS];
#pragma unroll
                for (int q = 0; q < NS; q++) v[q] = has[q] && j - dd[q] >= r0;       // rows above the component do not exist: their terms are dropped below
#pragma unroll
                for (int m = 0; m < DC; m++) {
#pragma unroll
                    for (int q = 0; q < NS; q++) c[q][m] = st[u].col[q][m];
                    cl[m] = st[u].li[m];
                }
                fetch(j - BACK_PD, st[u]);                              // in flight for the next BACK_PD steps
                // task d owns the pending sum of row j-(d-1): the sum of row j sits in lanes 0..DC-1 of acc[0]
                const double z = cy - acc[0];                           // lanes 0..DC-1
                // the shift does not depend on x_j: issue it before the dependent chain
                double sh[NS], sft[NS];
#pragma unroll
                for (int q = 0; q < NS; q++) sh[q] = lane_shift_down(acc[q], DC);
#pragma unroll
                for (int q = 0; q < NS; q++) {
                    const double next = (q + 1 < NS) ? sh[q + 1 < NS ? q + 1 : q] : 0.0;       // lanes near the top of a set take from the bottom of the next one
                    sft[q] = (lane + DC < 64) ? sh[q] : next;
                    if (!(lane + 64 * q + DC < T)) sft[q] = 0.0;
                }
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < DC; k++) x += cl[k] * lane_bcast(z, k);        // x_j[lane] = sum_k G[k][lane] z[k]
                if (j >= r1) x = cy;                                               // given
                else if (lane < DC) { if (to_lds) xs[(j - r0) * DC + lane] = x; else y[(size_t)j * DC + lane] = x; }
                double sm[NS];
#pragma unroll
                for (int q = 0; q < NS; q++) sm[q] = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) { const double xm = lane_bcast(x, m);
#pragma unroll
                    for (int q = 0; q < NS; q++) sm[q] += c[q][m] * xm; }
                // next step: task d owns row (j-1)-(d-1) = j-d, i.e. what task d+1 owned, plus this step's term for row j-d
#pragma unroll
                for (int q = 0; q < NS; q++) acc[q] = sft[q] + (v[q] ? sm[q] : 0.0);      // (one select per sum instead of one per loaded entry)
            }
        }
}

template <int DC, int NR, int MF = 0, int TCW = 3>      // MF bit 0: matrix-core panel, bit 1: matrix-core trailing update, bit 2 (alone): early look-ahead; TCW: columns of a trailing tile (3: 3x3 tiles, 2: 3x2)
__global__ void __launch_bounds__(768)
k_band_chol_v2(double* __restrict__ band, double* __restrict__ Ginv, double* __restrict__ Y, const int* __restrict__ pairs,
               const int* __restrict__ piv_lo, const int* __restrict__ piv_hi, const int* __restrict__ win_hi,
               const int* __restrict__ merge_from, int N, int b, int* __restrict__ fail_flag,
               const CholWaveMap wmap,                 // role of every physical wave (chol_wave_map; identity = roles in wave order)
               // fused launch of segments and the separators that wait for them (ba_handle.h band_direct): per workgroup the first of two flags to await
               // before its window is loaded (-1: none) and the flag to raise when its rows are in global memory (-1: none); flags hold launch numbers
               const int* __restrict__ await2 = nullptr, const int* __restrict__ signal = nullptr, int* __restrict__ flags = nullptr, int seq = 0) {
    constexpr int BB = DC * DC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int R = b + 1, W = b + 1, RW = W * BB;
    double* sWin = lds;                                     // [R][W][BB] ring of window rows
    double* sP = sWin + (size_t)R * RW;                     // [b][BB]    panel of the current step = L(j+k, j)
    double* sYr = sP + (size_t)b * BB;                      // [R][NR][DC] right-hand-side rows of the window
    double* sYj = sYr + (size_t)R * NR * DC;                // [NR][DC]   final y_j
    double* sG = sYj + NR * DC;                             // [BB]       inverse factor of the current diagonal block
    double* sD = sG + BB;                                   // [BB]       scratch: updated next diagonal block
    int* sPairs = reinterpret_cast<int*>(sD + BB);
    // role index of this (physical) wave and the matching thread index: everything below is written in terms of these two
    const int n = N * DC, nt = blockDim.x, lane = threadIdx.x & 63, nw = nt >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)wmap.v[threadIdx.x >> 6]), tid = wave * 64 + lane;
    const int ntw = nw - 2 - CHOL2_LOADERS;                 // trailing-update waves
    // pivots [r0, r1); the window (panels, trailing update, right-hand sides) runs on to row re >= r1.  re == r1 for a whole
    // component; a SEGMENT of a substructured component (band_sub.h) has re = r1 + b: the rows of the separator behind it
    // receive their factor blocks L(r, k) and their Schur update, and are written back by the epilogue instead of being pivots.
    const int r0 = piv_lo[blockIdx.x], r1 = piv_hi[blockIdx.x], re = win_hi[blockIdx.x];
    const int sig = signal ? signal[blockIdx.x] : -1, aw = await2 ? await2[blockIdx.x] : -1;
    if (r0 >= r1) { if (sig >= 0 && tid == 0) __hip_atomic_store(flags + sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); return; }
    for (int e = tid; e < b * (b + 1) / 2; e += nt) sPairs[e] = pairs[e];
    if (aw >= 0) {                                         // the two segments in front of this separator have written its rows
        if (tid == 0) { while (__hip_atomic_load(flags + aw, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(8);
                        while (__hip_atomic_load(flags + aw + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(8); }
        __syncthreads(); __threadfence();
    }
    // merge_from (twisted components, band_sub.h): this component is a separator of b rows whose second copy, rows mf..mf+b-1 in
    // REVERSED order, holds the Schur update and the forward-substitution share of the reversed segment: block (s, s-d) takes the
    // transpose of the copy's block (b-1-s+d, d).  The whole separator sits in the initial window (b < R), so merging is free here.
    const int mf = merge_from ? merge_from[blockIdx.x] : -1;
    if (mf < 0) {
        // the first window is one contiguous piece of the band: eight loads per lane in flight before the first LDS write (a plain
        // row-by-row copy pays one memory round trip per row: ~10 us of the 56 us of a 32-pivot segment)
        const int nrow0 = min(r0 + R, re) - r0, total = nrow0 * RW;
        const double* src = band + (size_t)r0 * RW;
        for (int base = 0; base < total; base += 8 * nt) {
            double tmp[8];
#pragma unroll
            for (int u = 0; u < 8; u++) tmp[u] = src[min(base + u * nt + tid, total - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = base + u * nt + tid;
                if (idx < total) { const int rr = idx / RW, e = idx - rr * RW; sWin[(size_t)((r0 + rr) % R) * RW + e] = tmp[u]; }
            }
        }
        for (int idx = tid; idx < nrow0 * NR * DC; idx += nt) {
            const int rr = idx / (NR * DC), e = idx - rr * (NR * DC);
            sYr[(size_t)((r0 + rr) % R) * NR * DC + e] = Y[(size_t)(e / DC) * n + (size_t)(r0 + rr) * DC + (e % DC)];
        }
    } else {
        // separator of a twisted component: same batching, two sources per entry (the copy's block (b-1-s+d, d), transposed)
        const int nrow0 = min(r0 + R, re) - r0, total = nrow0 * RW;
        const double* src = band + (size_t)r0 * RW;
        for (int base = 0; base < total; base += 8 * nt) {
            double tmp[8], tmp2[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = min(base + u * nt + tid, total - 1);
                const int s = idx / RW, e = idx - s * RW, d = e / BB, rc = e - d * BB, a = rc / DC, a2 = rc - a * DC;
                tmp[u] = src[idx];
                const int dd = min(d, s);                                  // clamped address; entries with d > s take nothing
                const double v2 = band[((size_t)(mf + b - 1 - s + dd) * W + dd) * BB + a2 * DC + a];
                tmp2[u] = (d <= s) ? v2 : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = base + u * nt + tid;
                if (idx < total) { const int rr = idx / RW, e = idx - rr * RW; sWin[(size_t)((r0 + rr) % R) * RW + e] = tmp[u] + tmp2[u]; }
            }
        }
        for (int idx = tid; idx < nrow0 * NR * DC; idx += nt) {
            const int s = idx / (NR * DC), e = idx - s * (NR * DC);
            sYr[(size_t)((r0 + s) % R) * NR * DC + e] = Y[(size_t)(e / DC) * n + (size_t)(r0 + s) * DC + (e % DC)] + Y[(size_t)(e / DC) * n + (size_t)(mf + b - 1 - s) * DC + (e % DC)];
        }
    }
    __syncthreads();
    if (wave == 0) {                                        // factor the first diagonal block
        double row[DC], g[DC];
        const double* D0 = sWin + (size_t)(r0 % R) * RW;
#pragma unroll
        for (int c = 0; c < DC; c++) row[c] = (lane < DC) ? D0[lane * DC + c] : ((lane == c) ? 1.0 : 0.0);
        if (!wave_chol_inverse<DC>(row, g) && lane == 0) *fail_flag = 1;
        if (lane < DC) {
#pragma unroll
            for (int r = 0; r < DC; r++) sG[r * DC + lane] = g[r];
        }
    }
    __syncthreads();
    const int jm0 = r0 % R;
    const bool is_writer = wave == nw - 1;
    const int lw = wave - 1 - ntw;                          // loader index, valid when 0 <= lw < CHOL2_LOADERS
    // ---- phase B, shared by every role: panel X_k = A_k G^T and y_j = G y_j into LDS
    const int li = lane & 15, lk = lane >> 4;
    // round 4: when every panel entry has a thread of its own (b BB <= threads: always in the shapes ba_handle.h launches) the entry's coordinates are the same in
    // every step -- two integer divisions and the address arithmetic leave the step loop (a step is ~10 ticks per instruction on every wave: profiles/r04_notes.md)
    // MF bit 2 (round 4, "early look-ahead"): the look-ahead wave keeps block 0 of the panel (X_1, 36 entries) for itself and has it -- and with it the update of the
    // next diagonal block -- done BEFORE barrier A, while the other waves compute the rest of the panel; its part after the barrier is the factorisation alone.
    // Needs one panel entry per thread with the look-ahead wave's 28 other lanes left out: b BB <= 36 + threads - 64.
    constexpr bool EARLY = (MF & 4) != 0;
    const bool pb_own = EARLY ? true : (b * BB <= nt);
    const int pb_e = EARLY ? ((wave == 0) ? (lane < BB ? lane : b * BB) : BB + tid - 64) : tid;
    const int pb_k = pb_e / BB, pb_rc = pb_e - pb_k * BB, pb_a = pb_rc / DC, pb_c = pb_rc - pb_a * DC;
    const int pb_offA = (pb_k + 1) * BB + pb_a * DC;
    const double* pb_G = sG + pb_c * DC;
    auto phaseB = [&](int j, int jm, int nb) {
        if constexpr ((MF & 1) != 0) {
            // panel X = A G^T, 16 window rows per wave: X(i, c) = sum_m A(i, m) G(c, m); tile rows 16 wave .. + 15, columns c = li < 6
            const int nrow = nb * DC;
            if (16 * wave < nrow) {
                const int i = 16 * wave + li; const bool vi = i < nrow; const int ic = vi ? i : 0;
                const int k = (ic * 43) >> 8, a = ic - 6 * k;                    // i / 6, i % 6 (exact below 128)
                int sl = jm + 1 + k; if (sl >= R) sl -= R;
                const double* A = sWin + (size_t)sl * RW + (size_t)(k + 1) * BB + a * DC;
                const double a0 = vi ? A[lk] : 0.0, a1 = (vi && lk < 2) ? A[4 + lk] : 0.0;
                const double b0 = (li < DC) ? sG[li * DC + lk] : 0.0, b1 = (li < DC && lk < 2) ? sG[li * DC + 4 + lk] : 0.0;
                typedef double v4d_ __attribute__((ext_vector_type(4)));
                v4d_ acc = {0.0, 0.0, 0.0, 0.0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) { const int row = 16 * wave + lk + 4 * q; if (row < nrow && li < DC) sP[row * DC + li] = acc[q]; }
            }
        } else if (pb_own) {
            if (pb_k < nb) {
                int sl = jm + 1 + pb_k; if (sl >= R) sl -= R;
                const double* A = sWin + (size_t)sl * RW + pb_offA;
                double x = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) x += A[m] * pb_G[m];
                sP[pb_e] = x;
            }
        } else
        for (int e = tid; e < nb * BB; e += nt) {
            const int k = e / BB, rc = e - k * BB, a = rc / DC, c = rc - a * DC;
            int sl = jm + 1 + k; if (sl >= R) sl -= R;
            const double* A = sWin + (size_t)sl * RW + (size_t)(k + 1) * BB + a * DC;
            const double* Gc = sG + c * DC;
            double x = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) x += A[m] * Gc[m];
            sP[e] = x;
        }
        if (tid >= nt - 128 && tid < nt - 128 + NR * DC) {
            const int q = tid - (nt - 128), r = q / DC, c = q - r * DC;
            const double* yr = sYr + (size_t)jm * NR * DC + r * DC;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < DC; m++) s += sG[c * DC + m] * yr[m];
            sYj[q] = s;
        }
    };
    // Each role runs its own copy of the step loop (a barrier only counts arrivals), so that no role carries another role's
    // registers around the back edge: a loop-carried register set that is refreshed by global loads in one branch only
    // becomes copies at the back edge, and the copies wait for the loads.
    if (wave == 0) {
        // ---- look-ahead: next diagonal block, its factor and inverse.  No global memory traffic.
        int jm = jm0;
        if constexpr (EARLY) {
            for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
                const int nb = min(b, re - 1 - j);
                phaseB(j, jm, nb);                                                   // this wave's share: X_1 = A(j+1, j) G_j^T -> block 0 of the panel
                double row[DC], g[DC];
                if (nb >= 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // wave-local: X_1 is in LDS
                    int s1 = jm + 1; if (s1 >= R) s1 -= R;
                    double* dblk = sWin + (size_t)s1 * RW;                          // block (j+1, j+1): final since the trailing update of step j-1
                    double* dout = (j + 1 < r1) ? sD : dblk;                        // last pivot of a segment: the first separator row takes the update in place
#pragma unroll
                    for (int e = lane; e < BB; e += 64) {
                        const int a = e / DC, c = e - a * DC;
                        double v = dblk[e];
#pragma unroll
                        for (int m = 0; m < DC; m++) v -= sP[a * DC + m] * sP[c * DC + m];
                        dout[e] = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int c = 0; c < DC; c++) row[c] = (lane < DC) ? sD[lane * DC + c] : ((lane == c) ? 1.0 : 0.0);
                }
                lds_barrier();
                if (j + 1 < r1) {
                    if (!wave_chol_inverse<DC>(row, g) && lane == 0) *fail_flag = 1;
                    if (lane < DC) {
#pragma unroll
                        for (int r = 0; r < DC; r++) sG[r * DC + lane] = g[r];
                    }
                }
                lds_barrier();
            }
        } else
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            phaseB(j, jm, nb);
            lds_barrier();
            if (j + 1 < r1) {
                int s1 = jm + 1; if (s1 >= R) s1 -= R;
                const double* dblk = sWin + (size_t)s1 * RW;                        // block (j+1, j+1)
#pragma unroll
                for (int e = lane; e < BB; e += 64) {                               // one pass for DC <= 8
                    const int a = e / DC, c = e - a * DC;
                    double v = dblk[e];
#pragma unroll
                    for (int m = 0; m < DC; m++) v -= sP[a * DC + m] * sP[c * DC + m];
                    sD[e] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // wave-local LDS round trip
                double row[DC], g[DC];
#pragma unroll
                for (int c = 0; c < DC; c++) row[c] = (lane < DC) ? sD[lane * DC + c] : ((lane == c) ? 1.0 : 0.0);
                if (!wave_chol_inverse<DC>(row, g) && lane == 0) *fail_flag = 1;
                if (lane < DC) {
#pragma unroll
                    for (int r = 0; r < DC; r++) sG[r * DC + lane] = g[r];
                }
            } else if (nb >= 1) {
                // last pivot of a segment: row j+1 is the first separator row, its diagonal block takes the update in place
                int s1 = jm + 1; if (s1 >= R) s1 -= R;
                double* dblk = sWin + (size_t)s1 * RW;
#pragma unroll
                for (int e = lane; e < BB; e += 64) {
                    const int a = e / DC, c = e - a * DC;
                    double v = dblk[e];
#pragma unroll
                    for (int m = 0; m < DC; m++) v -= sP[a * DC + m] * sP[c * DC + m];
                    dblk[e] = v;
                }
            }
            lds_barrier();
        }
    } else if (wave <= ntw) {
        // ---- trailing update of the window + right-hand sides (LDS only)
        // block tasks: (pair, row part, column part) -> TR x TR outputs in registers; pair 0 = (1,1) belongs to wave 0
        constexpr int TR = (DC % 3 == 0) ? 3 : 1, TC = (DC % 3 == 0) ? TCW : 1, TP = DC / TC, TPB = (DC / TR) * TP;      // TP column parts, DC / TR row parts per block
        const int cw = ntw * 64, ct = tid - 64;
        int jm = jm0;
        if constexpr ((MF & 2) != 0) {
            typedef double v4d_ __attribute__((ext_vector_type(4)));
            for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
                const int nb = min(b, re - 1 - j);
                phaseB(j, jm, nb);
                lds_barrier();
                const int nrow = nb * DC, TQp = (nrow + 15) >> 4, ntile = TQp * (TQp + 1) / 2;
                // three tiles in flight per pass: operand reads of all three, then the six instructions, then the read-modify-writes
                for (int t0 = wave - 1; t0 < ntile; t0 += 3 * ntw) {
                    int TI[3], TJ[3]; bool on[3]; v4d_ acc[3];
                    double a0[3], a1[3], b0[3], b1[3];
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const int t = t0 + u * ntw; on[u] = t < ntile; const int tc = on[u] ? t : 0;
                        int I = 0; while ((I + 1) * (I + 2) / 2 <= tc) I++;
                        TI[u] = I; TJ[u] = tc - I * (I + 1) / 2;
                        const int i = 16 * TI[u] + li, jj = 16 * TJ[u] + li; const bool vi = on[u] && i < nrow, vj = on[u] && jj < nrow;
                        const double* Pi = sP + (vi ? i : 0) * DC; const double* Pj = sP + (vj ? jj : 0) * DC;
                        a0[u] = vi ? Pi[lk] : 0.0; b0[u] = vj ? Pj[lk] : 0.0;
                        a1[u] = (vi && lk < 2) ? Pi[4 + lk] : 0.0; b1[u] = (vj && lk < 2) ? Pj[4 + lk] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) { acc[u] = v4d_{0.0, 0.0, 0.0, 0.0}; acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc[u], 0, 0, 0); }
#pragma unroll
                    for (int u = 0; u < 3; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc[u], 0, 0, 0);
                    // read-modify-write of the twelve accumulators: all addresses first, then all reads, then all writes (pointer-wise the
                    // compiler has to assume that a write may hit the next read and would run the twelve round trips one after the other)
                    double* dst[3][4]; double* dstm[3][4]; bool ok[3][4], mir[3][4]; double cur[3][4], curm[3][4];
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const int c = 16 * TJ[u] + li, kr = ((c * 43) >> 8) + 1, w = c - 6 * (kr - 1);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int r = 16 * TI[u] + lk + 4 * q, ir = ((r * 43) >> 8) + 1, a = r - 6 * (ir - 1);
                            ok[u][q] = on[u] && r < nrow && c < nrow && kr <= ir && !(ir == 1 && kr == 1);      // block (1, 1) belongs to wave 0
                            // diagonal blocks are stored whole: an element whose mirror image lies in a tile above the diagonal (never
                            // enumerated) writes that one too (P P^T is symmetric)
                            mir[u][q] = ok[u][q] && ir == kr && TI[u] > TJ[u];
                            const int irc = ok[u][q] ? ir : 1, krc = ok[u][q] ? kr : 1;
                            int si = jm + irc; if (si >= R) si -= R;
                            double* blk = sWin + (size_t)si * RW + (size_t)(irc - krc) * BB;
                            dst[u][q] = blk + (ok[u][q] ? a * DC + w : 0); dstm[u][q] = blk + (mir[u][q] ? w * DC + a : 0);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++)
#pragma unroll
                        for (int q = 0; q < 4; q++) { cur[u][q] = *dst[u][q]; curm[u][q] = *dstm[u][q]; }
#pragma unroll
                    for (int u = 0; u < 3; u++)
#pragma unroll
                        for (int q = 0; q < 4; q++) { if (ok[u][q]) *dst[u][q] = cur[u][q] - acc[u][q]; if (mir[u][q]) *dstm[u][q] = curm[u][q] - acc[u][q]; }
                }
                for (int qq = ct; qq < nb * DC; qq += cw) {                                    // right-hand sides: y_{j+kr} -= X_kr y_j
                    const int kr = qq / DC + 1, a = qq - (kr - 1) * DC;
                    int sk = jm + kr; if (sk >= R) sk -= R;
                    const double* Lk_ = sP + (size_t)(kr - 1) * BB + a * DC;
#pragma unroll
                    for (int r = 0; r < NR; r++) { double v = 0.0;
#pragma unroll
                        for (int m = 0; m < DC; m++) v += Lk_[m] * sYj[r * DC + m];
                        sYr[(size_t)sk * NR * DC + r * DC + a] -= v; }
                }
                lds_barrier();
            }
        } else {
        // round 4: with one block task per lane (the shapes ba_handle.h launches) the lane's task is the same in every step: its pair (an LDS round trip on the
        // dependent path of the phase), tile coordinates and operand offsets are fixed before the loop
        const int own_t = ct + TPB, own_pr = min(own_t / TPB, b * (b + 1) / 2 - 1), own_sub = own_t - (own_t / TPB) * TPB;
        const int own_pk = sPairs[own_pr], own_ir = own_pk & 0xffff, own_kr = own_pk >> 16;
        const int own_a0 = (own_sub / TP) * TR, own_c0 = (own_sub - (own_sub / TP) * TP) * TC;
        const double* own_Li = sP + (size_t)(own_ir - 1) * BB + own_a0 * DC;
        const double* own_Lk = sP + (size_t)(own_kr - 1) * BB + own_c0 * DC;
        const int own_dst = (own_ir - own_kr) * BB + own_a0 * DC + own_c0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            phaseB(j, jm, nb);
            lds_barrier();
            const int work = (nb * (nb + 1) / 2) * TPB;
            auto block_task_own = [&]() {
                double la[TR][DC], lk[TC][DC];
#pragma unroll
                for (int m = 0; m < DC; m++) {
#pragma unroll
                    for (int u = 0; u < TR; u++) la[u][m] = own_Li[u * DC + m];
#pragma unroll
                    for (int u = 0; u < TC; u++) lk[u][m] = own_Lk[u * DC + m];
                }
                int si = jm + own_ir; if (si >= R) si -= R;
                double* dst = sWin + (size_t)si * RW + own_dst;
#pragma unroll
                for (int u = 0; u < TR; u++)
#pragma unroll
                    for (int w = 0; w < TC; w++) { double v = 0.0;
#pragma unroll
                        for (int m = 0; m < DC; m++) v += la[u][m] * lk[w][m];
                        dst[u * DC + w] -= v; }
            };
            auto block_task = [&](int t) {
                const int pr = t / TPB, sub = t - pr * TPB, a0 = (sub / TP) * TR, c0 = (sub - (sub / TP) * TP) * TC;
                const int pk = sPairs[pr]; const int ir = pk & 0xffff, kr = pk >> 16;
                const double* Li_ = sP + (size_t)(ir - 1) * BB + a0 * DC;
                const double* Lk_ = sP + (size_t)(kr - 1) * BB + c0 * DC;
                double la[TR][DC], lk[TC][DC];
#pragma unroll
                for (int m = 0; m < DC; m++) {
#pragma unroll
                    for (int u = 0; u < TR; u++) la[u][m] = Li_[u * DC + m];
#pragma unroll
                    for (int u = 0; u < TC; u++) lk[u][m] = Lk_[u * DC + m];
                }
                int si = jm + ir; if (si >= R) si -= R;
                double* dst = sWin + (size_t)si * RW + (size_t)(ir - kr) * BB + a0 * DC + c0;
#pragma unroll
                for (int u = 0; u < TR; u++)
#pragma unroll
                    for (int w = 0; w < TC; w++) { double v = 0.0;
#pragma unroll
                        for (int m = 0; m < DC; m++) v += la[u][m] * lk[w][m];
                        dst[u * DC + w] -= v; }
            };
            auto rhs_task = [&](int qq) {                                         // right-hand sides: y_{j+kr} -= X_kr y_j
                const int kr = qq / DC + 1, a = qq - (kr - 1) * DC;
                int sk = jm + kr; if (sk >= R) sk -= R;
                const double* Lk_ = sP + (size_t)(kr - 1) * BB + a * DC;
#pragma unroll
                for (int r = 0; r < NR; r++) { double v = 0.0;
#pragma unroll
                    for (int m = 0; m < DC; m++) v += Lk_[m] * sYj[r * DC + m];
                    sYr[(size_t)sk * NR * DC + r * DC + a] -= v; }
            };
            // one task per lane; the right-hand-side tasks start on a wave of their own when the lanes allow it (a wave that held the tail of
            // the block tasks AND right-hand-side tasks ran both bodies: 2.2k cycles against 1.5k for its neighbours, and the step waits for it)
            const int nblk = work - TPB, nrhs = nb * DC, wbk = (nblk + 63) & ~63;
            if (wbk + nrhs <= cw) {
                if (ct < nblk) block_task_own();
                else if (ct >= wbk && ct - wbk < nrhs) rhs_task(ct - wbk);
            } else {
                for (int t = ct + TPB; t < work + nrhs; t += cw) { if (t < work) block_task(t); else rhs_task(t - work); }
            }
            lds_barrier();
        }
        }
    } else if (!is_writer) {
        // ---- loaders: together they bring in the row that enters the window (RW + NR*DC doubles), two steps ahead, in
        // registers; loads are unconditional from clamped addresses (a select on a loaded value would wait for it at once)
        constexpr int PRE = 5;                              // covers RW + NR*DC <= 64 * CHOL2_LOADERS * PRE; longer rows take the slow tail
        const int le0 = lw * 64 + lane;
        double preA[PRE], preB[PRE];                        // two rows in flight: the factor comes from another XCD's L2 / MALL, more than a step away
#define CHOL2_ISSUE(pre_, jn_)                                                                                        \
        do {                                                                                                          \
            const int jc_ = min((jn_), re - 1);                                                                       \
            _Pragma("unroll") for (int u = 0; u < PRE; u++) {                                                         \
                const int e = le0 + u * 64 * CHOL2_LOADERS;                                                           \
                const int q = max(min(e - RW, NR * DC - 1), 0);                                                       \
                const double* src = (e < RW) ? band + (size_t)jc_ * RW + e : Y + (size_t)(q / DC) * n + (size_t)jc_ * DC + (q % DC); \
                pre_[u] = *src;                                                                                       \
            }                                                                                                         \
        } while (0)
#define CHOL2_STEP(pre_, j_)                                                                                          \
        do {                                                                                                          \
            const int nb = min(b, re - 1 - (j_)), jn = (j_) + R;                                                      \
            phaseB((j_), jm, nb);                                                                                     \
            lds_barrier();                                                                                            \
            /* row j's slot is dead (its blocks left as panels of earlier steps): it takes row j + R */               \
            double* rowj = sWin + (size_t)jm * RW;                                                                    \
            double* yrow = sYr + (size_t)jm * NR * DC;                                                                \
            _Pragma("unroll") for (int u = 0; u < PRE; u++) {                                                         \
                const int e = le0 + u * 64 * CHOL2_LOADERS;                                                           \
                if (e < RW) rowj[e] = pre_[u]; else if (e < RW + NR * DC) yrow[e - RW] = pre_[u];                     \
            }                                                                                                         \
            if (jn < re) for (int e = le0 + PRE * 64 * CHOL2_LOADERS; e < RW + NR * DC; e += 64 * CHOL2_LOADERS) {    \
                if (e < RW) rowj[e] = band[(size_t)jn * RW + e];                                                      \
                else { const int q = e - RW; yrow[q] = Y[(size_t)(q / DC) * n + (size_t)jn * DC + (q % DC)]; }        \
            }                                                                                                         \
            CHOL2_ISSUE(pre_, jn + 2);                                                                                \
            lds_barrier();                                                                                            \
            jm = (jm + 1 == R) ? 0 : jm + 1;                                                                          \
        } while (0)
        CHOL2_ISSUE(preA, r0 + R);
        CHOL2_ISSUE(preB, r0 + R + 1);
        int jm = jm0;
        for (int j = r0; j < r1; j += 2) {
            CHOL2_STEP(preA, j);
            if (j + 1 < r1) CHOL2_STEP(preB, j + 1);
        }
#undef CHOL2_ISSUE
#undef CHOL2_STEP
    } else {
        // ---- writer: panel, y_j and G to global memory (stores only, never waited on)
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            phaseB(j, jm, nb);
            for (int e = lane; e < BB; e += 64) Ginv[(size_t)j * BB + e] = sG[e];    // before wave 0 replaces it
            lds_barrier();
            for (int e = lane; e < nb * BB; e += 64) { const int k = e / BB; band[((size_t)(j + 1 + k) * W + (k + 1)) * BB + (e - k * BB)] = sP[e]; }
            if (lane < NR * DC) Y[(size_t)(lane / DC) * n + (size_t)j * DC + (lane % DC)] = sYj[lane];
            lds_barrier();
        }
    }
    // ---- epilogue of a segment: the window now holds rows [r1, re) = the separator behind it, reduced by this segment
    if (re > r1) {
        __syncthreads();
        for (int row = r1; row < re; row++) {
            const int nin = (row - r1 + 1) * BB;                                   // blocks (row, r1..row): d = 0..row-r1
            const double* src = sWin + (size_t)(row % R) * RW;
            for (int e = tid; e < nin; e += nt) band[(size_t)row * RW + e] = src[e];
            for (int e = tid; e < NR * DC; e += nt) Y[(size_t)(e / DC) * n + (size_t)row * DC + (e % DC)] = sYr[(size_t)(row % R) * NR * DC + e];
        }
    }
    if (sig >= 0) {                                        // this segment's share of its separator is in global memory
        __threadfence(); __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Back substitution Y <- L^-T Y.  grid (components, NR), ONE wave per block.  Task t = (d-1)*DC + a (d = 1..b) owns the pending
// sum of row j-(d-1), component a; a lane carries tasks t = lane + 64 s, s < NS (b*DC <= 64 NS).  The factor streams from
// global memory BACK_PD steps ahead (a step is shorter than one memory latency), loads unconditional from clamped addresses.
// NS task sets per lane: 1 for b * DC <= 64 (half-width <= 10 at 6-dof blocks: the second set of loads / sums is compiled out), 2 up to 128, 3 up to 192
// (round 4: half-widths 22..30, the packed-window factorisation of band_kernels2p.h, twisted components included)
// tw_mode (round 4, twisted components: seg_0 | sep | seg_1 reversed | copy of sep): the separator's own back substitution used to be a launch of its own in front of
// this one (b steps + a launch gap in a dependent stream).  Mode 1 (seg_0): the rows [r1, re) ARE the separator and are solved here, the sweep simply starts at its last
// row.  Mode 2 (seg_1): the wave first solves the separator proper (rows gf .. gf+b-1, an isolated component) into LDS -- the same arithmetic as mode 1's, redundantly --
// and takes its given rows from there.  Mode 0 / nullptr: as before.
template <int DC, int NS>
__global__ void __launch_bounds__(64)
k_band_back_v2(const double* __restrict__ band, const double* __restrict__ Ginv, double* __restrict__ Y, const int* __restrict__ piv_lo,
               const int* __restrict__ piv_hi, const int* __restrict__ win_hi, const int* __restrict__ given_from, int N, int b,
               const int* __restrict__ tw_mode = nullptr) {
    __shared__ double xs[32 * DC];                                     // mode 2: the separator's solution (b <= 30 rows)
    const int n = N * DC, lane = threadIdx.x;
    // rows [r1, re) (the separator behind a segment, band_sub.h) already hold their solution: they only feed the pending sums
    // given_from (reversed segment of a twisted component): its given rows are a reversed copy of the separator at rows gf..gf+b-1,
    // row j of the copy = row gf + (re-1-j) of the separator proper, where the solution is
    const int mode = tw_mode ? tw_mode[blockIdx.x] : 0;
    if (piv_lo[blockIdx.x] >= piv_hi[blockIdx.x]) return;
    double* y = Y + (size_t)blockIdx.y * n;
    const int r0 = piv_lo[blockIdx.x], r1 = piv_hi[blockIdx.x], re = win_hi[blockIdx.x];
    const int gf = given_from ? given_from[blockIdx.x] : -1;
    if (mode == 2) {
        band_back_sweep<DC, NS, true, false>(band, Ginv, y, xs, gf, gf + b, gf + b, -1, b, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        band_back_sweep<DC, NS, false, true>(band, Ginv, y, xs, r0, r1, re, gf, b, lane);
    } else band_back_sweep<DC, NS, false, false>(band, Ginv, y, xs, r0, mode == 1 ? re : r1, re, mode == 1 ? -1 : gf, b, lane);
}

}  // namespace ssfm
