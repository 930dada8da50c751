// spherical_sfm_amd -- k_band_chol_v2p: the LDS-resident block-band Cholesky of band_kernels2.h with a PACKED window, for bands
// too wide for the square ring (half-width 22..30 at 6x6 blocks).
//
// k_band_chol_v2 keeps the window as a ring of b+1 rows x b+1 block slots: (b+1)^2 x 288 B = 140 KB at half-width 21, the most a CU
// holds.  Half of it is dead at any time: block (i, i-d) enters with its row at step i-b and is consumed as a panel input at step
// i-d, so it lives b+1-d steps.  Blocks of one diagonal d that are alive together number b+1-d, hence
//     slot(i, d) = base(d) + i mod (b+1-d),   base(d) = sum_{d' < d} (b+1-d'),
// is a collision-free packing with (b+1)(b+2)/2 slots -- the live triangle exactly: 109 KB at half-width 26, 143 KB at 30.  (Block
// (i, d) and its successor on the slot, (i + b+1-d, d), do not overlap: the latter enters at step i+1-d, the former died at i-d.)
// A connected camera ring with tracks of L frames has half-width 2 (L-1): 26 for tracks of 14 frames, where the square ring would
// need 210 KB and the global-memory kernel ran at 6.2 us per pivot (profiles/r04_notes.md).
//
// Everything else is k_band_chol_v2's VALU path: wave 0 look-ahead (factor + inverse of the next diagonal block), trailing update on
// 3x3 register tiles, two loader waves, one writer, panel = multiplication by G, two barriers per step.  A wide window has more tile
// tasks than lanes, so a lane owns up to NTASK FIXED tasks (pair, tile, operand offsets and the modulo counter of its slot are set
// up once; no division, no table lookup in the step loop), and the panel product NPB fixed entries per thread.
#pragma once
#include "band_kernels2.h"

namespace ssfm {

__device__ __host__ __forceinline__ int win2p_base(int d, int b) { return d * (b + 1) - d * (d - 1) / 2; }
__device__ __forceinline__ int win2p_slot(int i, int d, int b) { return win2p_base(d, b) + i % (b + 1 - d); }
inline size_t chol2p_lds_bytes(int b, int NR, bool t6 = false) {
    constexpr int DC = 6, BB = 36;
    return ((size_t)(b + 1) * (b + 2) / 2 * BB + (size_t)b * BB * (t6 ? 2 : 1) + (size_t)(b + 1) * NR * DC + NR * DC + 2 * BB) * sizeof(double) + ((size_t)b * (b + 1) / 2 + 2) * sizeof(int);
}

//   band / Ginv / Y / pairs / piv_lo / piv_hi / win_hi / merge_from / await2 / signal / flags: exactly as k_band_chol_v2.
// blockDim.x = 64 * (1 + ntw + 2 + 1), ntw trailing waves with NTASK * 64 * ntw >= 4 (b (b+1) / 2 - 1) + 6 b; NPB * blockDim.x >= 36 b;
// PRE * 128 >= 36 (b+1) + 6 NR.
// T6: the trailing update on 6x6 tiles, one block per lane (operands as ds_read_b128 from a transposed copy of the panel, the block read and written once as 18 + 18
// ds_*_b128): 404 KB of LDS traffic per step at half-width 26 against 606 KB for the 3x3 tiles, and one task per lane instead of three.
template <int NR, int NTASK, int NPB, int PRE, bool T6 = false>
__global__ void __launch_bounds__(T6 ? 768 : 1024)      // T6: twelve waves (eight trailing x 64 lanes >= blocks + right-hand sides up to half-width 29) leave 168 VGPRs for the 36 accumulators
k_band_chol_v2p(double* __restrict__ band, double* __restrict__ Ginv, double* __restrict__ Y, const int* __restrict__ pairs,
                const int* __restrict__ piv_lo, const int* __restrict__ piv_hi, const int* __restrict__ win_hi,
                const int* __restrict__ merge_from, int N, int b, int* __restrict__ fail_flag,
                const int* __restrict__ await2 = nullptr, const int* __restrict__ signal = nullptr, int* __restrict__ flags = nullptr, int seq = 0) {
    constexpr int DC = 6, BB = 36, LOADERS = 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int R = b + 1, RW = R * BB, nslots = R * (R + 1) / 2;
    double* sWin = lds;                                     // [nslots][BB] packed window
    double* sP = sWin + (size_t)nslots * BB;                // [b][BB]    panel of the current step = L(j+k, j)
    double* sYr = sP + (size_t)b * BB;                      // [R][NR][DC] right-hand-side rows of the window (ring)
    double* sYj = sYr + (size_t)R * NR * DC;                // [NR][DC]   final y_j
    double* sG = sYj + NR * DC;                             // [BB]       inverse factor of the current diagonal block
    double* sD = sG + BB;                                   // [BB]       scratch: updated next diagonal block
    double* sPT = sD + BB;                                  // [b][BB]    T6: the panel transposed, X_k[m][a] at m*6 + a
    int* sPairs = reinterpret_cast<int*>(sPT + (T6 ? (size_t)b * BB : 0));
    const int n = N * DC, nt = blockDim.x, lane = threadIdx.x & 63, nw = nt >> 6, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ntw = nw - 2 - LOADERS;
    const int r0 = piv_lo[blockIdx.x], r1 = piv_hi[blockIdx.x], re = win_hi[blockIdx.x];
    const int sig = signal ? signal[blockIdx.x] : -1, aw = await2 ? await2[blockIdx.x] : -1;
    if (r0 >= r1) { if (sig >= 0 && tid == 0) __hip_atomic_store(flags + sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); return; }
    for (int e = tid; e < b * (b + 1) / 2; e += nt) sPairs[e] = pairs[e];
    if (aw >= 0) {
        if (tid == 0) { while (__hip_atomic_load(flags + aw, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(8);
                        while (__hip_atomic_load(flags + aw + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(8); }
        __syncthreads(); __threadfence();
    }
    const int mf = merge_from ? merge_from[blockIdx.x] : -1;
    {
        // first window: rows r0 .. r0 + nrow0 - 1, of each only the blocks whose column is >= r0 (the others are not part of this window and their
        // slots belong to live blocks); eight loads per lane in flight
        const int nrow0 = min(r0 + R, re) - r0, total = nrow0 * RW;
        const double* src = band + (size_t)r0 * RW;
        for (int base = 0; base < total; base += 8 * nt) {
            double tmp[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = min(base + u * nt + tid, total - 1);
                double v = src[idx];
                if (mf >= 0) {      // separator of a twisted component: block (s, s-d) also takes the transpose of the copy's block (b-1-s+d, d)
                    const int s = idx / RW, e = idx - s * RW, d = e / BB, rc = e - d * BB, a = rc / DC, a2 = rc - a * DC, dd = min(d, s);
                    const double v2 = band[((size_t)(mf + b - 1 - s + dd) * R + dd) * BB + a2 * DC + a];
                    v += (d <= s) ? v2 : 0.0;
                }
                tmp[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = base + u * nt + tid;
                if (idx < total) { const int rr = idx / RW, e = idx - rr * RW, d = e / BB; if (d <= rr) sWin[(size_t)win2p_slot(r0 + rr, d, b) * BB + (e - d * BB)] = tmp[u]; }
            }
        }
        for (int idx = tid; idx < nrow0 * NR * DC; idx += nt) {
            const int s = idx / (NR * DC), e = idx - s * (NR * DC);
            double v = Y[(size_t)(e / DC) * n + (size_t)(r0 + s) * DC + (e % DC)];
            if (mf >= 0) v += Y[(size_t)(e / DC) * n + (size_t)(mf + b - 1 - s) * DC + (e % DC)];
            sYr[(size_t)((r0 + s) % R) * NR * DC + e] = v;
        }
    }
    __syncthreads();
    if (wave == 0) {                                        // factor the first diagonal block
        double row[DC], g[DC];
        const double* D0 = sWin + (size_t)win2p_slot(r0, 0, b) * BB;
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
    const int lw = wave - 1 - ntw;                          // loader index, valid when 0 <= lw < LOADERS
    // ---- phase B, every role: panel X_k = A_k G^T, NPB fixed entries per thread; entry u = (block k, row a, column c) reads block (j+1+k, j) = diagonal k+1
    int pbk[NPB], pbo[NPB], pbc[NPB], pbm[NPB], pbcnt[NPB];
#pragma unroll
    for (int u = 0; u < NPB; u++) {
        const int e = tid + u * nt, k = min(e / BB, b - 1), rc = e - (e / BB) * BB, a = rc / DC;
        pbk[u] = (e < b * BB) ? k : b;                      // b: no entry
        pbo[u] = a * DC; pbc[u] = (rc - a * DC) * DC;
        pbm[u] = b - k;                                     // blocks of diagonal k+1 alive together
        pbcnt[u] = (r0 + 1 + k) % pbm[u];
    }
    auto phaseB = [&](int jm, int nb) {
#pragma unroll
        for (int u = 0; u < NPB; u++) {
            if (pbk[u] < nb) {
                const double* A = sWin + (size_t)(win2p_base(pbk[u] + 1, b) + pbcnt[u]) * BB + pbo[u];
                const double* Gc = sG + pbc[u];
                double x = 0.0;
#pragma unroll
                for (int m = 0; m < DC; m++) x += A[m] * Gc[m];
                sP[tid + u * nt] = x;
                if (T6) sPT[pbk[u] * BB + (pbc[u] / DC) * DC + pbo[u] / DC] = x;       // X_k[a][c] -> [k][c][a]
            }
            pbcnt[u] = (pbcnt[u] + 1 == pbm[u]) ? 0 : pbcnt[u] + 1;
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
    if (wave == 0) {
        // ---- look-ahead: next diagonal block (diagonal 0: slot (j+1) mod R), its factor and inverse
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            phaseB(jm, nb);
            lds_barrier();
            int s1 = jm + 1; if (s1 >= R) s1 -= R;
            if (j + 1 < r1) {
                const double* dblk = sWin + (size_t)s1 * BB;                        // base(0) = 0
#pragma unroll
                for (int e = lane; e < BB; e += 64) {
                    const int a = e / DC, c = e - a * DC;
                    double v = dblk[e];
#pragma unroll
                    for (int m = 0; m < DC; m++) v -= sP[a * DC + m] * sP[c * DC + m];
                    sD[e] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double row[DC], g[DC];
#pragma unroll
                for (int c = 0; c < DC; c++) row[c] = (lane < DC) ? sD[lane * DC + c] : ((lane == c) ? 1.0 : 0.0);
                if (!wave_chol_inverse<DC>(row, g) && lane == 0) *fail_flag = 1;
                if (lane < DC) {
#pragma unroll
                    for (int r = 0; r < DC; r++) sG[r * DC + lane] = g[r];
                }
            } else if (nb >= 1) {
                double* dblk = sWin + (size_t)s1 * BB;
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
        // ---- trailing update + right-hand sides: NTASK fixed tasks per lane.  Task t < nblk: 3x3 tile (pair, row part, column part) of block (j+ir, j+kr);
        // nblk <= t < nblk + b DC: right-hand-side row
        constexpr int TR = 3, TP = 2, TPB = T6 ? 1 : 4;
        const int cw = ntw * 64, ct = tid - 64, npair = b * (b + 1) / 2, nblk = npair * TPB - TPB;
        int kind[NTASK], t_ir[NTASK], t_kr[NTASK], offi[NTASK], offk[NTASK], offd[NTASK], tm[NTASK], tcnt[NTASK];
#pragma unroll
        for (int u = 0; u < NTASK; u++) {
            const int t = ct + u * cw;
            kind[u] = (t < nblk) ? 0 : (t < nblk + b * DC) ? 1 : 2;
            if (kind[u] == 0) {
                const int tt = t + TPB, pr = tt / TPB, sub = tt - pr * TPB, a0 = T6 ? 0 : (sub / TP) * TR, c0 = T6 ? 0 : (sub - (sub / TP) * TP) * TR;
                const int pk = sPairs[pr], ir = pk & 0xffff, kr = pk >> 16, d = ir - kr;
                t_ir[u] = ir; t_kr[u] = kr; offi[u] = (ir - 1) * BB + a0 * DC; offk[u] = (kr - 1) * BB + c0 * DC;
                offd[u] = win2p_base(d, b) * BB + a0 * DC + c0; tm[u] = b + 1 - d; tcnt[u] = (r0 + ir) % tm[u];
            } else if (kind[u] == 1) {
                const int qq = t - nblk, kr = qq / DC + 1, a = qq - (kr - 1) * DC;
                t_ir[u] = kr; t_kr[u] = kr; offi[u] = a; offk[u] = (kr - 1) * BB + a * DC; offd[u] = 0; tm[u] = 1; tcnt[u] = 0;
            } else { t_ir[u] = b + 1; t_kr[u] = 0; offi[u] = offk[u] = offd[u] = 0; tm[u] = 1; tcnt[u] = 0; }
        }
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            phaseB(jm, nb);
            lds_barrier();
#pragma unroll
            for (int u = 0; u < NTASK; u++) {
                if (T6 && kind[u] == 0) {
                    if (t_ir[u] <= nb) {
                        typedef double d2_ __attribute__((ext_vector_type(2)));
                        double* dst = sWin + (size_t)tcnt[u] * BB + offd[u];
                        const double* pi = sPT + offi[u];
                        const double* pk = sPT + offk[u];
                        double acc[6][6];
#pragma unroll
                        for (int a = 0; a < 6; a++)
#pragma unroll
                            for (int c = 0; c < 6; c += 2) { const d2_ v = *reinterpret_cast<const d2_*>(dst + a * 6 + c); acc[a][c] = v.x; acc[a][c + 1] = v.y; }
#pragma unroll
                        for (int m = 0; m < 6; m++) {
                            double la[6], lk[6];
#pragma unroll
                            for (int q = 0; q < 6; q += 2) { const d2_ v = *reinterpret_cast<const d2_*>(pi + m * 6 + q); la[q] = v.x; la[q + 1] = v.y;
                                                             const d2_ w = *reinterpret_cast<const d2_*>(pk + m * 6 + q); lk[q] = w.x; lk[q + 1] = w.y; }
#pragma unroll
                            for (int a = 0; a < 6; a++)
#pragma unroll
                                for (int c = 0; c < 6; c++) acc[a][c] -= la[a] * lk[c];
                        }
#pragma unroll
                        for (int a = 0; a < 6; a++)
#pragma unroll
                            for (int c = 0; c < 6; c += 2) *reinterpret_cast<d2_*>(dst + a * 6 + c) = d2_{acc[a][c], acc[a][c + 1]};
                    }
                    tcnt[u] = (tcnt[u] + 1 == tm[u]) ? 0 : tcnt[u] + 1;
                } else if (kind[u] == 0) {
                    if (t_ir[u] <= nb) {
                        const double* Li_ = sP + offi[u];
                        const double* Lk_ = sP + offk[u];
                        double la[TR][DC], lk[TR][DC];
#pragma unroll
                        for (int q = 0; q < TR; q++)
#pragma unroll
                            for (int m = 0; m < DC; m++) { la[q][m] = Li_[q * DC + m]; lk[q][m] = Lk_[q * DC + m]; }
                        double* dst = sWin + (size_t)tcnt[u] * BB + offd[u];
#pragma unroll
                        for (int q = 0; q < TR; q++)
#pragma unroll
                            for (int w = 0; w < TR; w++) { double v = 0.0;
#pragma unroll
                                for (int m = 0; m < DC; m++) v += la[q][m] * lk[w][m];
                                dst[q * DC + w] -= v; }
                    }
                    tcnt[u] = (tcnt[u] + 1 == tm[u]) ? 0 : tcnt[u] + 1;
                } else if (kind[u] == 1) {
                    if (t_kr[u] <= nb) {                     // right-hand sides: y_{j+kr} -= X_kr y_j
                        int sk = jm + t_kr[u]; if (sk >= R) sk -= R;
                        const double* Lk_ = sP + offk[u];
#pragma unroll
                        for (int r = 0; r < NR; r++) { double v = 0.0;
#pragma unroll
                            for (int m = 0; m < DC; m++) v += Lk_[m] * sYj[r * DC + m];
                            sYr[(size_t)sk * NR * DC + r * DC + offi[u]] -= v; }
                    }
                }
            }
            lds_barrier();
        }
    } else if (!is_writer) {
        // ---- loaders: the row that enters the window, two steps ahead in registers; element e of the row image = block e / BB (diagonal d), entry e % BB, then
        // the right-hand sides.  Block (j + R, d) goes to slot base(d) + (j + R) mod (b+1-d): its predecessor there, block (j + d, d), was this step's panel input
        const int le0 = lw * 64 + lane;
        double preA[PRE], preB[PRE];
        int lm[PRE], lcnt[PRE], loff[PRE];
#pragma unroll
        for (int u = 0; u < PRE; u++) {
            const int e = le0 + u * 64 * LOADERS, d = min(e / BB, b);
            lm[u] = b + 1 - d; lcnt[u] = (r0 + R) % lm[u]; loff[u] = win2p_base(d, b) * BB + (e - (e / BB) * BB);
        }
#define CHOL2P_ISSUE(pre_, jn_)                                                                                       \
        do {                                                                                                          \
            const int jc_ = min((jn_), re - 1);                                                                       \
            _Pragma("unroll") for (int u = 0; u < PRE; u++) {                                                         \
                const int e = le0 + u * 64 * LOADERS;                                                                 \
                const int q = max(min(e - RW, NR * DC - 1), 0);                                                       \
                const double* src = (e < RW) ? band + (size_t)jc_ * RW + e : Y + (size_t)(q / DC) * n + (size_t)jc_ * DC + (q % DC); \
                pre_[u] = *src;                                                                                       \
            }                                                                                                         \
        } while (0)
#define CHOL2P_STEP(pre_, j_)                                                                                         \
        do {                                                                                                          \
            const int nb = min(b, re - 1 - (j_)), jn = (j_) + R;                                                      \
            phaseB(jm, nb);                                                                                           \
            lds_barrier();                                                                                            \
            double* yrow = sYr + (size_t)jm * NR * DC;                                                                \
            _Pragma("unroll") for (int u = 0; u < PRE; u++) {                                                         \
                const int e = le0 + u * 64 * LOADERS;                                                                 \
                if (jn < re) { if (e < RW) sWin[(size_t)lcnt[u] * BB + loff[u]] = pre_[u]; else if (e < RW + NR * DC) yrow[e - RW] = pre_[u]; } \
                lcnt[u] = (lcnt[u] + 1 == lm[u]) ? 0 : lcnt[u] + 1;                                                   \
            }                                                                                                         \
            CHOL2P_ISSUE(pre_, jn + 2);                                                                               \
            lds_barrier();                                                                                            \
            jm = (jm + 1 == R) ? 0 : jm + 1;                                                                          \
        } while (0)
        CHOL2P_ISSUE(preA, r0 + R);
        CHOL2P_ISSUE(preB, r0 + R + 1);
        int jm = jm0;
        for (int j = r0; j < r1; j += 2) {
            CHOL2P_STEP(preA, j);
            if (j + 1 < r1) CHOL2P_STEP(preB, j + 1);
        }
#undef CHOL2P_ISSUE
#undef CHOL2P_STEP
    } else {
        // ---- writer: panel, y_j and G to global memory (stores only, never waited on)
        int jm = jm0;
        for (int j = r0; j < r1; j++, jm = (jm + 1 == R) ? 0 : jm + 1) {
            const int nb = min(b, re - 1 - j);
            phaseB(jm, nb);
            for (int e = lane; e < BB; e += 64) Ginv[(size_t)j * BB + e] = sG[e];    // before wave 0 replaces it
            lds_barrier();
            for (int e = lane; e < nb * BB; e += 64) { const int k = e / BB; band[((size_t)(j + 1 + k) * R + (k + 1)) * BB + (e - k * BB)] = sP[e]; }
            if (lane < NR * DC) Y[(size_t)(lane / DC) * n + (size_t)j * DC + (lane % DC)] = sYj[lane];
            lds_barrier();
        }
    }
    // ---- epilogue of a segment: the window now holds rows [r1, re) = the separator behind it, reduced by this segment
    if (re > r1) {
        __syncthreads();
        for (int row = r1; row < re; row++) {
            const int nin = (row - r1 + 1) * BB;                                   // blocks (row, r1..row): d = 0..row-r1
            for (int e = tid; e < nin; e += nt) { const int d = e / BB; band[(size_t)row * RW + e] = sWin[(size_t)win2p_slot(row, d, b) * BB + (e - d * BB)]; }
            for (int e = tid; e < NR * DC; e += nt) Y[(size_t)(e / DC) * n + (size_t)row * DC + (e % DC)] = sYr[(size_t)(row % R) * NR * DC + e];
        }
    }
    if (sig >= 0) {
        __threadfence(); __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace ssfm
