// spherical_sfm_amd -- HIP kernels of the bundle-adjustment hot path (gfx950 / CDNA4, wave64, fp64).
//
// What the reference does inside ceres::Solve for SfM::Optimize (src/sfm.cpp:273-276) per LM iteration
//   evaluate residuals + Jacobians (Jets)  ->  loss corrector  ->  Jacobi scaling  ->
//   Schur-eliminate point blocks  ->  solve reduced camera system  ->  back-substitute  ->  candidate cost
// is laid out here as a handful of streaming passes.  Nothing stores a per-observation Jacobian: every pass
// re-linearises the observation it touches from the (L2-resident) camera table and its point, which costs
// ~150 fp64 flops and saves 176 B/observation of HBM traffic per pass.
//
// Data layout in HBM (all fp64 / int32):
//   cam[Nc*6]  [t;r]               rot[Nc*27] per-camera R, Rd, M (ssfm_math.h angle_axis_derivative_aid)
//   pts[nP*3]                      obs_xy[M] double2, obs_cam[M], obs_pt[M]  point-major, pt_start[nP+1]
//   cam_start[Nc+1], cam_obs[M]    camera-major view of the same observations
//   scale_cam[Nc*6], scale_pt[nP*3], scale_f   Jacobi column scales, 0 for parameters that are constant
//   gp[nP*3]                                   per-point J_p^T r  ((V + D^2)^-1 lives in the record PS only)
//   S_val[nnzb*DC*DC]              block-CSR reduced camera system (row_ptr/col_idx), rhs[Nc*DC+1]
// DC = 3 when every translation is fixed (spherical BA: only r varies), 6 otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include "ssfm_math.h"
#include "det_acc.h"

namespace ssfm {

static __global__ void k_profile_pad() {}      // absorbs the dispatch latency of an idle queue in front of a profiled launch (ba_solver.hip)

// slots of the per-iteration scalar block (device doubles)
enum {
    SC_COST = 0,      // 1/2 sum rho at x                       (sharded by point)
    SC_FJJ = 1,       // sum (s_f J_f)^2                        (sharded)
    SC_FJR = 2,       // sum s_f J_f r                          (sharded)
    SC_FWW = 3,       // sum_p Wf V^-1 Wf^T                     (sharded)
    SC_FWG = 4,       // sum_p Wf V^-1 g_p                      (sharded)
    SC_MODEL = 5,     // sum m (r + m/2)                        (sharded)
    SC_STEP2_PT = 6,  // |delta|^2 over points                  (sharded)
    SC_XN2_PT = 7,    // |candidate|^2 over points              (sharded)
    SC_CAND_COST = 8, // 1/2 sum rho at candidate               (sharded)
    SC_X0N2_PT = 9,   // |x|^2 over points (iteration 0)        (sharded)
    SC_NSUM = 10,
    SC_GMAX = 10,     // max |gradient| (uint64 bit pattern)    (max-reduced)
    SC_STEP2_CAM = 11, SC_XN2_CAM = 12, SC_X0N2_CAM = 13,   // replicated camera/focal parts
    SC_TOTAL = 16
};
// Same-address floating-point atomics serialise at ~12 ns each (measured: k_point_lin 41 us with 391 workgroups x 6 atomics,
// 110 us with 1563 x 6), so the per-workgroup partial sums go to one of SC_NSLOT replicas of the scalar block (workgroup id
// mod SC_NSLOT); readers fold the replicas (k_finalize_S on the device, the host after the copy, k_scal_fold before a collective).
constexpr int SC_NSLOT = 64;
__device__ __forceinline__ double* scal_slot(double* scal) { return scal + (size_t)(blockIdx.x & (SC_NSLOT - 1)) * SC_TOTAL; }
// SSFM_DETERMINISTIC (det_acc.h): the scalar block's sums go to long accumulators instead -- entry (replica, k) of the block at `base` has its LA_STRIDE words at
// lacc[(replica SC_TOTAL + k) LA_STRIDE]; lacc == nullptr: the floating-point atomic
struct DetScal { const double* base = nullptr; long long* lacc = nullptr; };
__device__ __forceinline__ void sadd(const DetScal& ds, double* p, double v) {
    if (!ds.lacc) { unsafeAtomicAdd(p, v); return; }
    lacc_add(ds.lacc + (size_t)(p - ds.base) * LA_STRIDE, v);
}
// limbs -> doubles, and the limbs cleared for the next assembly.  Workgroups [0, gz): entries [0, n) of the zone at `zone` (two limbs each; a poisoned assembly decodes
// to NaN); the last workgroup: the scalars k with bit k of kmask set, summed over the SC_NSLOT replicas of the long accumulators, into replica 0 of `scal`.
static __global__ void __launch_bounds__(256)
k_det_decode(double* __restrict__ zone, long long* __restrict__ limb, size_t n, int gz, double* __restrict__ scal, long long* __restrict__ lacc, unsigned kmask) {
    if ((int)blockIdx.x < gz) {
        const bool poisoned = limb[-1] != 0;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gz * blockDim.x) {
            const long long hi = limb[2 * i], lo = limb[2 * i + 1];
            zone[i] = poisoned ? __builtin_nan("") : zdecode(hi, lo);
            if (hi != 0 || lo != 0) { limb[2 * i] = 0; limb[2 * i + 1] = 0; }
        }
        return;
    }
    // one wave per scalar in turn: lane = replica
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int k = w; k < SC_TOTAL; k += nw) {
        if (!((kmask >> k) & 1u)) continue;
        long long* a = lacc + ((size_t)lane * SC_TOTAL + k) * LA_STRIDE;
        long long tot[LA_STRIDE];
#pragma unroll
        for (int j = 0; j < LA_STRIDE; j++) tot[j] = a[j];                        // (all limbs in flight before the first clear: see publish_body)
#pragma unroll
        for (int j = 0; j < LA_STRIDE; j++) if (tot[j] != 0) a[j] = 0;
#pragma unroll
        for (int j = 0; j < LA_STRIDE; j++) {
            long long v = tot[j];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);          // integer sums: exact, any order
            tot[j] = v;
        }
        if (lane == 0) scal[k] = lacc_value(tot, tot[LA_NL]);
    }
}
enum { PCG_RZ = 0, PCG_BN2 = 1, PCG_RR = 2, PCG_DONE = 3, PCG_ITERS = 4, PCG_BREAKDOWN = 5, PCG_TOTAL = 8 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Transposing wave reduction: every lane enters with N <= 64 values; lane L leaves with the wave-wide sum of value wave_tr_index()
// = its six lane bits reversed (0 when that index is >= N).  Halving butterfly: at every step a lane keeps the even- or the
// odd-indexed half of its values (by one lane bit), sends the other half to its partner and adds what it receives -- N/2 + N/4 + ...
// < N exchanges instead of 6 N for N separate wave_sum calls.
template <int N, int MASK>
struct WaveTR {
    static __device__ __forceinline__ double run(double (&v)[N]) {
        constexpr int H = (N + 1) / 2;
        const bool up = (threadIdx.x & MASK) != 0;
        double w[H];
#pragma unroll
        for (int j = 0; j < H; j++) {
            const double lo = v[2 * j], hi = (2 * j + 1 < N) ? v[2 * j + 1] : 0.0;
            const double mine = up ? hi : lo, send = up ? lo : hi;
            w[j] = mine + __shfl_xor(send, MASK, 64);
        }
        return WaveTR<H, MASK / 2>::run(w);
    }
};
template <int N>
struct WaveTR<N, 0> { static __device__ __forceinline__ double run(double (&v)[N]) { static_assert(N == 1, "at most 64 values"); return v[0]; } };
template <int N>
__device__ __forceinline__ double wave_transpose_sum(double (&v)[N]) { return WaveTR<N, 32>::run(v); }
__device__ __forceinline__ int wave_tr_index() {
    const int l = threadIdx.x & 63;
    return ((l >> 5) & 1) | (((l >> 4) & 1) << 1) | (((l >> 3) & 1) << 2) | (((l >> 2) & 1) << 3) | (((l >> 1) & 1) << 4) | ((l & 1) << 5);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum of N values per thread; result valid in thread 0.  red: LDS scratch [N * (blockDim/64)]
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* red) {
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double t = wave_transpose_sum(v);                // value i of the wave in the lane with wave_tr_index() == i
    const int slot = wave_tr_index();
    if (slot < N) red[slot * nw + w] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) { double s = 0; for (int k = 0; k < nw; k++) s += red[i * nw + k]; v[i] = s; }
    }
    __syncthreads();
}
__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// ---- one observation: robustified residual + Jacobian blocks (unscaled) ---------------------------
// reference ReprojectionError (src/sfm.cpp:38-63) + Ceres corrector (rho'' <= 0 => sqrt(rho') scaling)
struct ObsLin {
    double r[2], Jf[2], Jt[2][3], Jr[2][3], Jp[2][3], half_rho;
};
// fp64 reciprocal / reciprocal square root from the hardware estimates + two Newton steps each (the IEEE division and sqrt sequences
// are 14-15 instructions; every re-linearised observation paid one division for 1/z, one for rho' and one sqrt)
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}
// The same to full precision with ONE third-order step: v_rsq_f64 is good to 5e-8 (scripts/lab/rsq_lab.hip), e = 1 - d y^2, y (1 + e/2 + 3 e^2 / 8) leaves an
// error of order e^3 = 1e-22.  Five dependent instructions instead of seven: for the factor-and-invert chain of the band kernels, where instructions are time.
__device__ __forceinline__ double fast_rsqrt3(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}
// sqrt(rho'(s)) of robust_loss (ssfm_math.h): Cauchy 1/sqrt(1 + s/a^2) is ONE reciprocal square root
__device__ __forceinline__ double loss_sqrt_weight(int type, double a, double s) {
    if (type == 1) return fast_rsqrt(1.0 + s * fast_rcp(a * a));
    if (type == 2) return sqrt(fast_rsqrt(1.0 + s * fast_rcp(a * a)));      // SoftLOne: rho' = (1 + s/a^2)^(-1/2)
    return 1.0;
}
__device__ __forceinline__ bool project(double f, const double* t, const double* R, const double* X, double ox, double oy,
                                        double& xp, double& yp, double& iz, double& r0, double& r1) {
    const double p0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double p1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double p2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    iz = fast_rcp(p2); xp = p0 * iz; yp = p1 * iz;
    r0 = f * xp - ox; r1 = f * yp - oy;
    return true;
}
__device__ __forceinline__ double obs_cost(double f, const double* t, const double* R, const double* X, double ox, double oy, int loss, double la) {
    double xp, yp, iz, r0, r1; project(f, t, R, X, ox, oy, xp, yp, iz, r0, r1);
    double rho0, rho1; robust_loss(loss, la, r0 * r0 + r1 * r1, rho0, rho1);
    return 0.5 * rho0;
}
template <bool NEED_T>
__device__ __forceinline__ void lin_obs(double f, const double* t, const double* rot, const double* X, double ox, double oy,
                                        int loss, double la, ObsLin& L) {
    const double* R = rot; const double* Rd = rot + 9; const double* Mm = rot + 18;
    double xp, yp, iz, r0, r1; project(f, t, R, X, ox, oy, xp, yp, iz, r0, r1);
    double rho0, rho1; robust_loss(loss, la, r0 * r0 + r1 * r1, rho0, rho1);      // only rho(s) is used from here: dead code where the cost is not
    const double sr = loss_sqrt_weight(loss, la, r0 * r0 + r1 * r1);
    L.half_rho = 0.5 * rho0;
    L.r[0] = sr * r0; L.r[1] = sr * r1;
    L.Jf[0] = sr * xp; L.Jf[1] = sr * yp;
    const double a = sr * f * iz;
    const double A0[3] = {a, 0.0, -a * xp}, A1[3] = {0.0, a, -a * yp};
    if (NEED_T) { for (int k = 0; k < 3; k++) { L.Jt[0][k] = A0[k]; L.Jt[1][k] = A1[k]; } }
    double B0[3], B1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        L.Jp[0][k] = A0[0] * R[k] + A0[2] * R[6 + k];
        L.Jp[1][k] = A1[1] * R[3 + k] + A1[2] * R[6 + k];
        B0[k] = A0[0] * Rd[k] + A0[2] * Rd[6 + k];
        B1[k] = A1[1] * Rd[3 + k] + A1[2] * Rd[6 + k];
    }
    // Jr = -(A Rd) [X]x M.  B . (X x m) = (B x X) . m: two cross products with X instead of three
    const double w0[3] = {B0[1] * X[2] - B0[2] * X[1], B0[2] * X[0] - B0[0] * X[2], B0[0] * X[1] - B0[1] * X[0]};
    const double w1[3] = {B1[1] * X[2] - B1[2] * X[1], B1[2] * X[0] - B1[0] * X[2], B1[0] * X[1] - B1[1] * X[0]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        L.Jr[0][k] = -(w0[0] * Mm[k] + w0[1] * Mm[3 + k] + w0[2] * Mm[6 + k]);
        L.Jr[1][k] = -(w1[0] * Mm[k] + w1[1] * Mm[3 + k] + w1[2] * Mm[6 + k]);
    }
}
// point-side view of the same linearisation (no camera blocks, R only): residual, focal column, point block
struct ObsPoint { double r[2], Jf[2], Jp[2][3], half_rho; };
__device__ __forceinline__ void lin_obs_point(double f, const double* t, const double* R, const double* X, double ox, double oy, int loss, double la,
                                              ObsPoint& L) {
    double xp, yp, iz, r0, r1; project(f, t, R, X, ox, oy, xp, yp, iz, r0, r1);
    double rho0, rho1; robust_loss(loss, la, r0 * r0 + r1 * r1, rho0, rho1);      // only rho(s) is used from here: dead code where the cost is not
    const double sr = loss_sqrt_weight(loss, la, r0 * r0 + r1 * r1);
    L.half_rho = 0.5 * rho0;
    L.r[0] = sr * r0; L.r[1] = sr * r1;
    L.Jf[0] = sr * xp; L.Jf[1] = sr * yp;
    const double a = sr * f * iz, a02 = -a * xp, a12 = -a * yp;
#pragma unroll
    for (int k = 0; k < 3; k++) { L.Jp[0][k] = a * R[k] + a02 * R[6 + k]; L.Jp[1][k] = a * R[3 + k] + a12 * R[6 + k]; }
}
// scaled camera block Jc[2][DC] of an observation (DC=6: [t r], DC=3: r only)
template <int DC>
__device__ __forceinline__ void cam_block(const ObsLin& L, const double* sc6, double (&Jc)[2][DC]) {
    if (DC == 6) {
#pragma unroll
        for (int k = 0; k < 3; k++) { Jc[0][k] = L.Jt[0][k] * sc6[k]; Jc[1][k] = L.Jt[1][k] * sc6[k];
                                      Jc[0][3 + k] = L.Jr[0][k] * sc6[3 + k]; Jc[1][3 + k] = L.Jr[1][k] * sc6[3 + k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) { Jc[0][k] = L.Jr[0][k] * sc6[3 + k]; Jc[1][k] = L.Jr[1][k] * sc6[3 + k]; }
    }
}

// unscaled camera block (the pair kernel applies the Jacobi scales once per folded block instead of once per pair)
template <int DC>
__device__ __forceinline__ void cam_block_raw(const ObsLin& L, double (&Jc)[2][DC]) {
    if (DC == 6) {
#pragma unroll
        for (int k = 0; k < 3; k++) { Jc[0][k] = L.Jt[0][k]; Jc[1][k] = L.Jt[1][k]; Jc[0][3 + k] = L.Jr[0][k]; Jc[1][3 + k] = L.Jr[1][k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) { Jc[0][k] = L.Jr[0][k]; Jc[1][k] = L.Jr[1][k]; }
    }
}

// ---- K0: per-camera rotation tables ----------------------------------------------------------------
static __global__ void k_cam_rot(const double* __restrict__ cam, double* __restrict__ rot, int Nc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Nc) return;
    double aa[3] = {cam[c * 6 + 3], cam[c * 6 + 4], cam[c * 6 + 5]};
    double R[9], Rd[9], M[9];
    angle_axis_derivative_aid(aa, R, Rd, M);
    for (int i = 0; i < 9; i++) { rot[c * 27 + i] = R[i]; rot[c * 27 + 9 + i] = Rd[i]; rot[c * 27 + 18 + i] = M[i]; }
}

// the first launch of a solve: rotation tables + the clears the iteration-0 sums need (two hipMemsetAsync launches less per solve)
static __global__ void k_cam_rot0(const double* __restrict__ cam, double* __restrict__ rot, int Nc, double* __restrict__ z0, int n0, double* __restrict__ z1, int n1) {
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n0; i += blockDim.x) z0[i] = 0.0;
        for (int i = threadIdx.x; i < n1; i += blockDim.x) z1[i] = 0.0;
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Nc) return;
    double aa[3] = {cam[c * 6 + 3], cam[c * 6 + 4], cam[c * 6 + 5]};
    double R[9], Rd[9], M[9];
    angle_axis_derivative_aid(aa, R, Rd, M);
    for (int i = 0; i < 9; i++) { rot[c * 27 + i] = R[i]; rot[c * 27 + 9 + i] = Rd[i]; rot[c * 27 + 18 + i] = M[i]; }
}

// ---- one-time: squared column norms of the unscaled robustified Jacobian (Jacobi scaling, iteration 0)
// (each kernel also writes the Jacobi scale of the columns it owns: mask * 1/(1 + sqrt(norm^2)), problem_impl.cc / trust_region_minimizer.cc of Ceres 2.2)
// points + focal: one lane per point;  cameras: one workgroup per camera over its observation list (no atomics)
static __device__ __forceinline__ void colnorm_pt_body(const int bx, double* __restrict__ red, const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
                          const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
                          const int* __restrict__ pt_start, int nP, int loss, double la,
                          double* __restrict__ diag_pt, double* __restrict__ diag_f,
                          const double* __restrict__ mask_pt, double* __restrict__ scale_pt, int jacobi, double* __restrict__ df_part) {
    const int p = bx * blockDim.x + threadIdx.x;
    double df[1] = {0.0};
    if (p < nP) {
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double f = focal[0];
        double dp[3] = {0, 0, 0};
        for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
            const int c = obs_cam[j]; const double2 o = obs_xy[j];
            ObsLin L; lin_obs<false>(f, cam + 6 * c, rot + 27 * c, X, o.x, o.y, loss, la, L);
            for (int k = 0; k < 3; k++) dp[k] += L.Jp[0][k] * L.Jp[0][k] + L.Jp[1][k] * L.Jp[1][k];
            df[0] += L.Jf[0] * L.Jf[0] + L.Jf[1] * L.Jf[1];
        }
        for (int k = 0; k < 3; k++) { diag_pt[3 * p + k] = dp[k]; scale_pt[3 * p + k] = mask_pt[3 * p + k] * (jacobi ? 1.0 / (1.0 + sqrt(dp[k])) : 1.0); }
    }
    block_sum<1>(df, red);
    if (threadIdx.x == 0) { if (df_part) df_part[bx] = df[0]; else unsafeAtomicAdd(diag_f, df[0]); }     // df_part (deterministic mode): k_startup_tail adds the parts in order
}
static __device__ __forceinline__ void colnorm_cam_body(const int bx, double* __restrict__ red, const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
              const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_pt,
              const int* __restrict__ cam_start, const int* __restrict__ cam_obs, int loss, double la, double* __restrict__ diag_cam,
              const double* __restrict__ mask_cam, double* __restrict__ scale_cam, int jacobi) {   // scale_cam == nullptr: the norms still need a sum over ranks
    const int c = bx;
    double d[6] = {0, 0, 0, 0, 0, 0};
    const double f = focal[0];
    for (int q = cam_start[c] + threadIdx.x; q < cam_start[c + 1]; q += blockDim.x) {
        const int j = cam_obs[q], p = obs_pt[j];
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double2 o = obs_xy[j];
        ObsLin L; lin_obs<true>(f, cam + 6 * c, rot + 27 * c, X, o.x, o.y, loss, la, L);
        for (int k = 0; k < 3; k++) { d[k] += L.Jt[0][k] * L.Jt[0][k] + L.Jt[1][k] * L.Jt[1][k]; d[3 + k] += L.Jr[0][k] * L.Jr[0][k] + L.Jr[1][k] * L.Jr[1][k]; }
    }
    block_sum<6>(d, red);
    if (threadIdx.x == 0) for (int k = 0; k < 6; k++) {
        diag_cam[c * 6 + k] = d[k];
        if (scale_cam) scale_cam[c * 6 + k] = mask_cam[c * 6 + k] * (jacobi ? 1.0 / (1.0 + sqrt(d[k])) : 1.0);
    }
}
static __global__ void __launch_bounds__(256)
k_colnorm_cam(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
              const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_pt,
              const int* __restrict__ cam_start, const int* __restrict__ cam_obs, int loss, double la, double* __restrict__ diag_cam,
              const double* __restrict__ mask_cam, double* __restrict__ scale_cam, int jacobi) {
    __shared__ double red[6 * 4];
    colnorm_cam_body((int)blockIdx.x, red, cam, rot, pts, focal, obs_xy, obs_pt, cam_start, cam_obs, loss, la, diag_cam, mask_cam, scale_cam, jacobi);
}
// both column-norm passes of a solve's start in ONE launch (round 6): workgroups [0, gp) take the points, the rest one camera each -- the two passes are independent and
// each was a launch of its own in the dependent stream (12 + 17 us per solve)
static __global__ void __launch_bounds__(256)
k_colnorm_both(int gp, const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts, const double* __restrict__ focal,
               const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam, const int* __restrict__ pt_start, int nP, int loss, double la,
               double* __restrict__ diag_pt, double* __restrict__ diag_f, const double* __restrict__ mask_pt, double* __restrict__ scale_pt, int jacobi, double* __restrict__ df_part,
               const int* __restrict__ obs_pt, const int* __restrict__ cam_start, const int* __restrict__ cam_obs, double* __restrict__ diag_cam,
               const double* __restrict__ mask_cam, double* __restrict__ scale_cam) {
    __shared__ double red[6 * 4];
    if ((int)blockIdx.x < gp) colnorm_pt_body((int)blockIdx.x, red, cam, rot, pts, focal, obs_xy, obs_cam, pt_start, nP, loss, la, diag_pt, diag_f, mask_pt, scale_pt, jacobi, df_part);
    else colnorm_cam_body((int)blockIdx.x - gp, red, cam, rot, pts, focal, obs_xy, obs_pt, cam_start, cam_obs, loss, la, diag_cam, mask_cam, scale_cam, jacobi);
}
static __global__ void k_make_scale(const double* __restrict__ diag, const double* __restrict__ mask, double* __restrict__ scale, int n, int jacobi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scale[i] = mask[i] * (jacobi ? 1.0 / (1.0 + sqrt(diag[i])) : 1.0);
}

// ---- K1: point pass.  One lane per point: V = sum Jp^T Jp (+D^2), V^-1, g_p, focal coupling ----------
// Ceres SchurEliminator "chunk" work for the e-block, with the LM diagonal D_p^2 = clamp(diag V)/radius.
template <int OBS_UNROLL>
static __global__ void __launch_bounds__(256)
k_point_lin(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
            const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
            const int* __restrict__ pt_start, int nP, const double* __restrict__ scale_pt, const double* __restrict__ scale_f,
            int loss, double la, double radius, double min_diag, double max_diag,
            double* __restrict__ Vs, double* __restrict__ gp, double* __restrict__ scal,
            const double* __restrict__ spec, long long* __restrict__ lacc = nullptr) {
    const DetScal ds{scal, lacc};
    // spec (speculative launch behind k_publish of the previous iteration): [go, radius] as decided on the device
    if (spec) { if (spec[0] == 0.0) return; radius = spec[1]; }
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};   // cost, FJJ, FJR, FWW, FWG
    double gmax = 0.0;
    if (p < nP) {
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double sp[3] = {scale_pt[3 * p], scale_pt[3 * p + 1], scale_pt[3 * p + 2]};
        const double f = focal[0], sf = scale_f[0];
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, wf[3] = {0, 0, 0};
        // observations in groups of OBS_UNROLL: indices first, then every camera's [t | R], then the arithmetic, so that the
        // dependent gathers of a group are in flight together (a point has 3-10 observations; lanes beyond the list recompute
        // the last one with zero weight)
        const int js = pt_start[p], je = pt_start[p + 1];
        int cn[OBS_UNROLL]; double2 on[OBS_UNROLL];
#pragma unroll
        for (int u = 0; u < OBS_UNROLL; u++) { const int jj = min(js + u, je - 1); cn[u] = obs_cam[jj]; on[u] = obs_xy[jj]; }
        for (int jb = js; jb < je; jb += OBS_UNROLL) {
            int cc[OBS_UNROLL]; double2 oo[OBS_UNROLL]; double tR[OBS_UNROLL][12];
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) { cc[u] = cn[u]; oo[u] = on[u]; }
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
#pragma unroll
                for (int k = 0; k < 3; k++) tR[u][k] = cam[6 * (size_t)cc[u] + k];
#pragma unroll
                for (int k = 0; k < 9; k++) tR[u][3 + k] = rot[27 * (size_t)cc[u] + k];
            }
            // the indices of the next group travel with this group's camera tables: one dependent round trip per group
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) { const int jj = min(jb + OBS_UNROLL + u, je - 1); cn[u] = obs_cam[jj]; on[u] = obs_xy[jj]; }
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
                const double wgt = (jb + u < je) ? 1.0 : 0.0;
                ObsPoint L; lin_obs_point(f, tR[u], tR[u] + 3, X, oo[u].x, oo[u].y, loss, la, L);
                acc[0] += wgt * L.half_rho;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const double j0 = L.Jp[a][0] * sp[0] * wgt, j1 = L.Jp[a][1] * sp[1] * wgt, j2 = L.Jp[a][2] * sp[2] * wgt, jf = L.Jf[a] * sf * wgt;
                    V[0] += j0 * j0; V[1] += j0 * j1; V[2] += j0 * j2; V[3] += j1 * j1; V[4] += j1 * j2; V[5] += j2 * j2;
                    g[0] += j0 * L.r[a]; g[1] += j1 * L.r[a]; g[2] += j2 * L.r[a];
                    wf[0] += jf * j0; wf[1] += jf * j1; wf[2] += jf * j2;
                    acc[1] += jf * jf; acc[2] += jf * L.r[a];
                }
            }
        }
        if (sp[0] > 0.0) {
            gmax = fmax(fabs(g[0] / sp[0]), fmax(fabs(g[1] / sp[1]), fabs(g[2] / sp[2])));
            V[0] += fmin(fmax(V[0], min_diag), max_diag) / radius;
            V[3] += fmin(fmax(V[3], min_diag), max_diag) / radius;
            V[5] += fmin(fmax(V[5], min_diag), max_diag) / radius;
        } else { V[0] = V[3] = V[5] = 1.0; }   // constant point: identity block, zero coupling
        double Vi[6]; sym3_inverse(V, Vi);
        const double u0 = wf[0] * Vi[0] + wf[1] * Vi[1] + wf[2] * Vi[2];
        const double u1 = wf[0] * Vi[1] + wf[1] * Vi[3] + wf[2] * Vi[4];
        const double u2 = wf[0] * Vi[2] + wf[1] * Vi[4] + wf[2] * Vi[5];
        acc[3] = u0 * wf[0] + u1 * wf[1] + u2 * wf[2];
        acc[4] = u0 * g[0] + u1 * g[1] + u2 * g[2];
        // (V^-1 itself and w_f are not stored: nothing reads w_f, and k_point_backsub recovers V^-1 b from the record below -- 9 doubles per
        // point less to write back at the end of a kernel whose run time follows its store volume)
        // per-point record of the Schur kernels, Jacobi point scales folded in so that they need no scale loads:
        //   [ diag(s) V^-1 diag(s) (6) | s o (V^-1 g) (3) | s o (V^-1 w_f) (3) ]
        double* ps = Vs + 12 * (size_t)p;
        ps[0] = Vi[0] * sp[0] * sp[0]; ps[1] = Vi[1] * sp[0] * sp[1]; ps[2] = Vi[2] * sp[0] * sp[2];
        ps[3] = Vi[3] * sp[1] * sp[1]; ps[4] = Vi[4] * sp[1] * sp[2]; ps[5] = Vi[5] * sp[2] * sp[2];
        ps[6] = sp[0] * (Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2]); ps[7] = sp[1] * (Vi[1] * g[0] + Vi[3] * g[1] + Vi[4] * g[2]);
        ps[8] = sp[2] * (Vi[2] * g[0] + Vi[4] * g[1] + Vi[5] * g[2]);
        ps[9] = sp[0] * u0; ps[10] = sp[1] * u1; ps[11] = sp[2] * u2;
        for (int k = 0; k < 3; k++) gp[3 * p + k] = g[k];
    }
    // every wave folds its own five sums (one transposing reduction) and issues its own atomics into the slot replica of its wave index:
    // no LDS, no workgroup barrier at the end of the kernel (the barrier made every wave wait for the slowest of its workgroup: 2.4 us of a
    // wave's 9 at config 2; scripts/lab/point_lab.hip)
    static_assert(SC_COST == 0 && SC_FJJ == 1 && SC_FJR == 2 && SC_FWW == 3 && SC_FWG == 4, "the five sums are the first five scalars");
    const double t = wave_transpose_sum(acc);
    gmax = wave_max(gmax);
    const int slot = wave_tr_index();
    double* sl = scal + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (SC_NSLOT - 1)) * SC_TOTAL;
    if (slot < 5) sadd(ds, &sl[slot], t);
    if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&sl[SC_GMAX], gmax);
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup w runs on XCD w % 8) and every XCD has its own L2.  Wave-task lists
// are ordered by camera, and neighbouring cameras share points: this maps the workgroups of one XCD onto one CONTIGUOUS eighth of
// the list, so that the points a camera range re-reads are found in that XCD's L2 instead of being fetched once per XCD.
__device__ __forceinline__ int xcd_contiguous_block(int w, int nwg) {
    const int x = w & 7, idx = w >> 3;
    // workgroups on XCD y: (nwg - y + 7) / 8;  base = how many sit on XCDs 0..x-1
    int base = 0;
#pragma unroll
    for (int y = 0; y < 7; y++) if (y < x) base += (nwg - y + 7) >> 3;
    return base + idx;
}

// ---- camera-side sums, second generation: diagonal blocks of the reduced system, right-hand side, focal border ---------
// One WAVE per task = a run of <= 4 batches (64 observations each) of one camera's observation list; no LDS, no barriers,
// gathers of batch i+1 in flight during batch i, per-point record PS = [diag(s) V^-1 diag(s) | s o V^-1 g | s o V^-1 w_f].
//   S[c,c]  += sum_j Jc^T Jc - W PS_V W^T        (W = Jc^T Jp, unscaled on the point side)
//   rhs_c   += sum_j Jc^T r - W PS_g ;  S_fc += sum_j Jc^T J_f - W PS_w ;  diag U (for the LM diagonal) and Jc^T r on their own
// Everything is accumulated with atomics into buffers zeroed once per pass.
template <int DC>
__global__ void __launch_bounds__(256, 2)
k_cam_sums2(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
            const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ cam_obs,
            const int* __restrict__ cam_obs_pt, const int* __restrict__ task_cam, const int* __restrict__ task_q0,
            const int* __restrict__ task_q1, int ntasks, const int* __restrict__ row_ptr, const int* __restrict__ diag_slot,
            const double* __restrict__ scale_cam, const double* __restrict__ scale_f, const double* __restrict__ PS, int loss, double la,
            double* __restrict__ S_val, double* __restrict__ rhs, double* __restrict__ Udiag, double* __restrict__ Sfc,
            double* __restrict__ gcraw, const unsigned char* __restrict__ pt_skip, DetZone dz = DetZone{}) {
    constexpr int BB = DC * DC;
    constexpr int NU = DC * (DC + 1) / 2;
    constexpr int NSM = NU + 4 * DC;           // [S_cc (upper) | Jc^T r | -W g' | focal coupling | diag U]   (<= 64 for DC <= 6)
    const int lane = threadIdx.x & 63;
    const int task = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (task >= ntasks) return;
    const int c = __builtin_amdgcn_readfirstlane(task_cam[task]);
    const int q0 = __builtin_amdgcn_readfirstlane(task_q0[task]), q1 = __builtin_amdgcn_readfirstlane(task_q1[task]);
    const double f = focal[0], sf = scale_f[0];
    double sm[NSM];
#pragma unroll
    for (int i = 0; i < NSM; i++) sm[i] = 0.0;
#define CS2_LOAD(qb_, X_, P_, o_, wgt_)                                                                               \
    do {                                                                                                              \
        const int q_ = (qb_) + lane;                                                                                  \
        wgt_ = (q_ < q1) ? 1.0 : 0.0;                                                                                 \
        const int qc_ = min(q_, q1 - 1);                                                                              \
        const int j_ = cam_obs[qc_], p_ = cam_obs_pt[qc_];                                                            \
        if (pt_skip && pt_skip[p_]) wgt_ = 0.0;                 /* a point of a signature group: k_schur_gram has its sums */ \
        _Pragma("unroll") for (int k = 0; k < 3; k++) X_[k] = pts[3 * (size_t)p_ + k];                                \
        _Pragma("unroll") for (int k = 0; k < 12; k++) P_[k] = PS[12 * (size_t)p_ + k];                               \
        o_ = obs_xy[j_];                                                                                              \
    } while (0)
#define CS2_COMPUTE(X_, P_, o_, wgt_)                                                                                 \
    do {                                                                                                              \
        ObsLin L_; lin_obs<DC == 6>(f, cam + 6 * c, rot + 27 * c, X_, o_.x, o_.y, loss, la, L_);                      \
        double Jc_[2][DC]; cam_block<DC>(L_, scale_cam + 6 * c, Jc_);                                                 \
        const double jf0 = L_.Jf[0] * sf, jf1 = L_.Jf[1] * sf;                                                        \
        _Pragma("unroll") for (int a = 0; a < DC; a++) { Jc_[0][a] *= wgt_; Jc_[1][a] *= wgt_; }                      \
        /* everything through 2-vectors: Jc^T (I - Jp PS_V Jp^T) Jc, Jc^T (Jp PS_g), Jc^T (J_f - Jp PS_w); W = Jc^T Jp is never formed */ \
        double Q_[2][3], y_[2], z_[2];                                                                                \
        _Pragma("unroll") for (int r = 0; r < 2; r++) {                                                               \
            const double j0 = L_.Jp[r][0], j1 = L_.Jp[r][1], j2 = L_.Jp[r][2];                                        \
            Q_[r][0] = j0 * P_[0] + j1 * P_[1] + j2 * P_[2];                                                          \
            Q_[r][1] = j0 * P_[1] + j1 * P_[3] + j2 * P_[4];                                                          \
            Q_[r][2] = j0 * P_[2] + j1 * P_[4] + j2 * P_[5];                                                          \
            y_[r] = j0 * P_[6] + j1 * P_[7] + j2 * P_[8];                                                             \
            z_[r] = ((r == 0) ? jf0 : jf1) - (j0 * P_[9] + j1 * P_[10] + j2 * P_[11]);                                \
        }                                                                                                             \
        const double c00 = 1.0 - (Q_[0][0] * L_.Jp[0][0] + Q_[0][1] * L_.Jp[0][1] + Q_[0][2] * L_.Jp[0][2]);          \
        const double c01 = -(Q_[0][0] * L_.Jp[1][0] + Q_[0][1] * L_.Jp[1][1] + Q_[0][2] * L_.Jp[1][2]);               \
        const double c11 = 1.0 - (Q_[1][0] * L_.Jp[1][0] + Q_[1][1] * L_.Jp[1][1] + Q_[1][2] * L_.Jp[1][2]);          \
        int u = 0;                                                                                                    \
        _Pragma("unroll") for (int a = 0; a < DC; a++) {                                                              \
            const double e0 = c00 * Jc_[0][a] + c01 * Jc_[1][a], e1 = c01 * Jc_[0][a] + c11 * Jc_[1][a];              \
            sm[NU + a] += Jc_[0][a] * L_.r[0] + Jc_[1][a] * L_.r[1];                                                  \
            sm[NU + DC + a] -= Jc_[0][a] * y_[0] + Jc_[1][a] * y_[1];                                                 \
            sm[NU + 2 * DC + a] += Jc_[0][a] * z_[0] + Jc_[1][a] * z_[1];                                             \
            sm[NU + 3 * DC + a] += Jc_[0][a] * Jc_[0][a] + Jc_[1][a] * Jc_[1][a];                                     \
            _Pragma("unroll") for (int b = a; b < DC; b++) sm[u++] += e0 * Jc_[0][b] + e1 * Jc_[1][b];                \
        }                                                                                                             \
    } while (0)
    double Xa[3], Pa[12], wa; double2 oa;
    double Xb[3], Pb[12], wb; double2 ob;
    CS2_LOAD(q0, Xa, Pa, oa, wa);
    for (int qb = q0; qb < q1; qb += 128) {
        CS2_LOAD(qb + 64, Xb, Pb, ob, wb);
        CS2_COMPUTE(Xa, Pa, oa, wa);
        CS2_LOAD(qb + 128, Xa, Pa, oa, wa);
        if (qb + 64 < q1) CS2_COMPUTE(Xb, Pb, ob, wb);
    }
#undef CS2_LOAD
#undef CS2_COMPUTE
    // fold: value i ends up in the lane with wave_tr_index() == i, which owns its destination(s)
    const double mine = wave_transpose_sum(sm);
    const int slot = wave_tr_index();
    if (slot < NU) {
        int a = 0, rem = slot; while (rem >= DC - a) { rem -= DC - a; a++; }
        const int b = a + rem;
        double* blk = S_val + ((size_t)row_ptr[c] + diag_slot[c]) * BB;
        zadd(dz, &blk[a * DC + b], mine); if (b != a) zadd(dz, &blk[b * DC + a], mine);
    } else if (slot < NU + DC) { const int a = slot - NU; zadd(dz, &rhs[c * DC + a], mine); zadd(dz, &gcraw[c * DC + a], mine); }
    else if (slot < NU + 2 * DC) zadd(dz, &rhs[c * DC + slot - NU - DC], mine);
    else if (slot < NU + 3 * DC) zadd(dz, &Sfc[c * DC + slot - NU - 2 * DC], mine);
    else if (slot < NSM) zadd(dz, &Udiag[c * DC + slot - NU - 3 * DC], mine);
}

// ---- Schur complement from the slot-sorted pair lists, second generation ------------------------------------------
// One WAVE per task = a run of <= 16 batches (64 pairs each) of one camera row; no LDS, no barriers.  The gathers of batch
// bt+1 are issued before the arithmetic of batch bt (two register sets, loop unrolled by two so that neither set is copied at
// the back edge), loads are unconditional from clamped indices (padding lanes carry a zero weight), the point index rides in
// the pair list (one dependent load less) and the Jacobi point scales are pre-folded into Vs.  Blocks are folded across the
// wave at every slot change and added to S with global atomics (a few dozen per task).
// OCC = waves per SIMD the register allocation is held to (EXPERIMENT, VERDICT r2 #6a: SSFM_PAIRS_OCC3=1 runs the 6-dof kernel at 3 waves per
// SIMD, i.e. <= 168 VGPRs instead of 224; profiles/r03_notes.md has the measurement)
template <int DC, int OCC = ((DC == 3) ? 3 : 2)>
__global__ void __launch_bounds__(256, OCC)
k_schur_pairs2(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
               const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ row_ptr,
               const int* __restrict__ col_idx, const int* __restrict__ task_cam, const int* __restrict__ task_b0,
               const int* __restrict__ task_b1, int ntasks, const int* __restrict__ batch_slot, const int* __restrict__ pair_j,
               const int* __restrict__ pair_j2, const int* __restrict__ pair_p, const double* __restrict__ scale_cam,
               const double* __restrict__ Vs, int loss, double la, double* __restrict__ S_val, DetZone dz = DetZone{}) {
    constexpr int BB = DC * DC;
    const int lane = threadIdx.x & 63;
    const int task = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (task >= ntasks) return;
    const int c = __builtin_amdgcn_readfirstlane(task_cam[task]);
    const int b0 = __builtin_amdgcn_readfirstlane(task_b0[task]), b1 = __builtin_amdgcn_readfirstlane(task_b1[task]);
    const int rb = __builtin_amdgcn_readfirstlane(row_ptr[c]);
    const double f = focal[0];
    // two register sets, spelled out as plain locals (a struct handed to lambdas by reference ends up in scratch memory)
#define SP2_LOAD(bt_, X_, V_, o_, o2_, wgt_)                                                                         \
    do {                                                                                                              \
        const size_t e_ = (size_t)(bt_) * 64 + lane;                                                                  \
        const int j_ = pair_j[e_], j2_ = pair_j2[e_], p_ = pair_p[e_];                                                \
        const int jc_ = max(j_, 0), j2c_ = max(j2_, 0), pc_ = max(p_, 0);                                             \
        wgt_ = (j_ >= 0) ? 1.0 : 0.0;                                                                                 \
        _Pragma("unroll") for (int k = 0; k < 3; k++) X_[k] = pts[3 * (size_t)pc_ + k];                               \
        _Pragma("unroll") for (int k = 0; k < 6; k++) V_[k] = Vs[12 * (size_t)pc_ + k];                               \
        o_ = obs_xy[jc_]; o2_ = obs_xy[j2c_];                                                                         \
    } while (0)
    double blk[DC][DC];
#pragma unroll
    for (int a = 0; a < DC; a++)
#pragma unroll
        for (int b = 0; b < DC; b++) blk[a][b] = 0.0;
    int cur = -1;
    const int tr_slot = wave_tr_index();
    auto fold = [&]() {                                    // wave-uniform: add the finished block to S (transposing reduction: element e lands in one lane)
        double* dst = S_val + ((size_t)rb + cur) * BB;
        double flat[BB];
#pragma unroll
        for (int a = 0; a < DC; a++)
#pragma unroll
            for (int b = 0; b < DC; b++) { flat[a * DC + b] = blk[a][b]; blk[a][b] = 0.0; }
        const double v = wave_transpose_sum(flat);
        if (tr_slot < BB) {                                // Jacobi scales of both cameras, once per folded block
            constexpr int off = (DC == 6) ? 0 : 3;
            const int a = tr_slot / DC, b = tr_slot - a * DC, c2f = col_idx[rb + cur];
            zadd(dz, &dst[tr_slot], v * scale_cam[6 * c + off + a] * scale_cam[6 * c2f + off + b]);
        }
    };
#define SP2_COMPUTE(bt_, X_, V_, o_, o2_, wgt_)                                                                       \
    do {                                                                                                              \
        const int slot_ = __builtin_amdgcn_readfirstlane(batch_slot[bt_]);                                            \
        if (slot_ != cur) { if (cur >= 0) fold(); cur = slot_; }                                                      \
        const int c2_ = __builtin_amdgcn_readfirstlane(col_idx[rb + slot_]);                                          \
        /* block -= Jc_i^T (Jp_i Vs Jp_j^T) Jc_j : the 2x2 core first, W = Jc^T Jp is never formed; unscaled Jc (scales at the  \
           fold); for 6-dof blocks d/dt = [a 0 -a x; 0 a -a y]: its two zeros are skipped explicitly */               \
        double Jc_[2][DC], Q_[2][3];                                                                                  \
        {                                                                                                             \
            ObsLin L_; lin_obs<DC == 6>(f, cam + 6 * c, rot + 27 * c, X_, o_.x, o_.y, loss, la, L_);                  \
            cam_block_raw<DC>(L_, Jc_);                                                                               \
            _Pragma("unroll") for (int r = 0; r < 2; r++) {                                                           \
                const double j0 = L_.Jp[r][0] * wgt_, j1 = L_.Jp[r][1] * wgt_, j2 = L_.Jp[r][2] * wgt_;               \
                Q_[r][0] = j0 * V_[0] + j1 * V_[1] + j2 * V_[2];                                                      \
                Q_[r][1] = j0 * V_[1] + j1 * V_[3] + j2 * V_[4];                                                      \
                Q_[r][2] = j0 * V_[2] + j1 * V_[4] + j2 * V_[5];                                                      \
            }                                                                                                         \
        }                                                                                                             \
        ObsLin L2_; lin_obs<DC == 6>(f, cam + 6 * c2_, rot + 27 * c2_, X_, o2_.x, o2_.y, loss, la, L2_);              \
        double Jc2_[2][DC]; cam_block_raw<DC>(L2_, Jc2_);                                                             \
        double C_[2][2], D_[2][DC];                                                                                   \
        _Pragma("unroll") for (int r = 0; r < 2; r++)                                                                 \
            _Pragma("unroll") for (int q = 0; q < 2; q++)                                                             \
                C_[r][q] = Q_[r][0] * L2_.Jp[q][0] + Q_[r][1] * L2_.Jp[q][1] + Q_[r][2] * L2_.Jp[q][2];               \
        _Pragma("unroll") for (int r = 0; r < 2; r++)                                                                 \
            _Pragma("unroll") for (int b = 0; b < DC; b++) {                                                          \
                if (DC == 6 && b == 0) D_[r][b] = C_[r][0] * Jc2_[0][0];                                              \
                else if (DC == 6 && b == 1) D_[r][b] = C_[r][1] * Jc2_[1][1];                                         \
                else D_[r][b] = C_[r][0] * Jc2_[0][b] + C_[r][1] * Jc2_[1][b];                                        \
            }                                                                                                         \
        _Pragma("unroll") for (int a = 0; a < DC; a++)                                                                \
            _Pragma("unroll") for (int b = 0; b < DC; b++) {                                                          \
                if (DC == 6 && a == 0) blk[a][b] -= Jc_[0][0] * D_[0][b];                                             \
                else if (DC == 6 && a == 1) blk[a][b] -= Jc_[1][1] * D_[1][b];                                        \
                else blk[a][b] -= Jc_[0][a] * D_[0][b] + Jc_[1][a] * D_[1][b];                                        \
            }                                                                                                         \
    } while (0)
    double Xa[3], Va[6], wa; double2 oa, o2a;
    double Xb[3], Vb[6], wb; double2 ob, o2b;
    SP2_LOAD(b0, Xa, Va, oa, o2a, wa);
    for (int bt = b0; bt < b1; bt += 2) {
        SP2_LOAD(min(bt + 1, b1 - 1), Xb, Vb, ob, o2b, wb);
        SP2_COMPUTE(bt, Xa, Va, oa, o2a, wa);
        SP2_LOAD(min(bt + 2, b1 - 1), Xa, Va, oa, o2a, wa);
        if (bt + 1 < b1) SP2_COMPUTE(bt + 1, Xb, Vb, ob, o2b, wb);
    }
#undef SP2_LOAD
#undef SP2_COMPUTE
    if (cur >= 0) fold();
}

// ---- Schur complement of SIGNATURE GROUPS on the matrix cores (round 3) -----------------------------------------------------------------
// A task = a run of consecutive points observed by exactly the same K <= 8 cameras (BAFlat::gr_rec).  With V^-1 = L L^T per point and the half
// products Y_k = Jc_k^T Jp_k L (DC x 3) of its K observations stacked into a (DC K) x 3 matrix per point, the blocks of all K (K - 1) / 2 camera pairs AND
// the Schur corrections of the K diagonal blocks are the blocks of ONE Gram matrix  G = sum_points Y Y^T  -- every observation is linearised ONCE (the
// pair kernel re-linearises it K - 1 times, k_cam_sums2 once more) and the products run as v_mfma_f64_16x16x4_f64 tiles.  FP64 MFMA has the VALU's flop
// rate on gfx950; what it buys is issue slots: one instruction per 1024 multiply-adds with its operands from LDS, against ~470 VALU instructions per 64 pairs.
// (Round 6, measured: a wave does NOT overlap its fp64 matrix products with its own fp64 vector work -- the hand-interleaved, double-buffered loop of
// profiles/r06_notes.md r06k ran the 14 sub-chunks of a config-2 task in 24.3 us against 22.6 one after the other: both use the same fp64 units, so a sub-chunk costs
// its ~2000 vector cycles PLUS its ~1450 matrix cycles whatever the order.)
// One wave per task, sub-chunks of 8 points: lane (point = lane & 7, k = lane >> 3) linearises observation k of its point (camera records staged in LDS
// once per task) and lays Y out in LDS as sY[row = DC k + d][3 point + c] (24 columns, leading dimension 28: the fragment loads touch every bank twice,
// the hardware minimum for 64 x 8 bytes); then the wave runs the tiles of the lower triangle over the 6 k-steps, accumulating across sub-chunks.
// While the matrix pipe works the lane also keeps the camera-side sums of ITS camera in registers (what k_cam_sums2 computes from a second
// linearisation): Jc^T Jc, Jc^T r, Jc^T Jp V^-1 g, the focal coupling -- folded over the 8 lanes of a camera once per task.
// At the end every lane adds the entries of G it holds to S with the Jacobi scales of both cameras (unscaled Jc inside, scales at the fold): blocks
// (a > b) to their slot, blocks (a = a) to the diagonal block of camera a.
// Rows >= DC K of a tile hold whatever LDS holds: row i of Y only reaches row i and column i of G, and those entries are never emitted -- so nothing is
// zeroed and a launch sizes LDS for the largest K of its task range (DC Kmax rows + GRAM_TAIL: 13.4 KB at K = 8).  NT = 16-row tiles in use: the tasks
// are sorted by K and every tile count has its own launch over a contiguous task range (DC K <= 16: one product per k-step, <= 32: three, else six).
// The points' records of the NEXT sub-chunk are loaded before the tiles of the current one run (the observations of a group are consecutive, K per point:
// no dependent index load), so the matrix pipe covers the load latency; the waves of a SIMD cover each other's linearisation.
// History (profiles/r03_notes.md): 32-point sub-chunks on half the lanes 70.7 us at config 2; 16 points on all lanes 57; + prefetch, no zeroing 39.5;
// emission from LDS instead of per-entry global index loads 35.3; 8-point sub-chunks (12 instead of 7 waves per CU) 380 -> 355 us at the configs[4] size;
// with the camera-side sums (k_cam_sums2 no longer runs for these cameras) 48.7 us / 400 us against 39.5 + 20.5 / 931 + 414 for the pair path.
#ifndef SSFM_GRAM_LD
#define SSFM_GRAM_LD 28      // row stride of the half products in LDS, doubles (swept in round 4: scripts/gpu_gram_ld.sh, profiles/r04_notes.md)
#endif
constexpr int GRAM_CAMREC = 34, GRAM_LD = SSFM_GRAM_LD, GRAM_SUB = GRAM_SUB_PTS, GRAM_TAIL = GRAM_KMAX * GRAM_CAMREC + 48 + GRAM_NPAIR / 2 + GRAM_KMAX / 2;
// transposing reduction over the 8 lanes that differ in lane bits 0..2: N values per lane in, ceil(N / 8) out; out[j] of a lane is the 8-lane sum of
// value 8 j + 4 (lane & 1) + 2 ((lane >> 1) & 1) + ((lane >> 2) & 1)
template <int N, int MASK>
struct OctTR {
    static __device__ __forceinline__ void run(double (&v)[N], double* out) {
        constexpr int H = (N + 1) / 2;
        const bool up = (threadIdx.x & MASK) != 0;
        double w[H];
#pragma unroll
        for (int j = 0; j < H; j++) {
            const double lo = v[2 * j], hi = (2 * j + 1 < N) ? v[2 * j + 1] : 0.0;
            const double mine = up ? hi : lo, send = up ? lo : hi;
            w[j] = mine + __shfl_xor(send, MASK, 64);
        }
        OctTR<H, MASK / 2>::run(w, out);
    }
};
template <int N>
struct OctTR<N, 0> { static __device__ __forceinline__ void run(double (&v)[N], double* out) {
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = v[j]; } };
// W waves per workgroup, one task and one LDS slice each; the waves never talk to each other, so the two hand-overs per sub-chunk are wave-local
// (LDS operations of one wave complete in order; the fences keep the compiler from moving them) instead of workgroup barriers.
__device__ __forceinline__ void wave_lds_handover() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// entry (r, i) of the unscaled 6-dof camera block that is zero by construction: d/dt = [a 0 -a x; 0 a -a y]
template <int DC> __device__ __forceinline__ constexpr bool jc_zero(int r, int i) { return DC == 6 && ((r == 0 && i == 1) || (r == 1 && i == 0)); }
// The cameras of a task (gr_rec[4 ..]) as a LANE VECTOR -- lane i < GRAM_KMAX holds camera i, read with one ds_bpermute -- and the gathers of the per-camera tables
// with every load of a table in flight before its first LDS store (round 6).  Rounds 3-5 kept the ids in scalar registers behind a select chain; the compiler turned
// that chain into a scratch array, and the head of every task was, per 64 table entries, scratch load -> wait -> global load -> wait -> LDS store: five to seven
// dependent round trips before the first observation could be linearised.
__device__ __forceinline__ int gram_cam_of(int camv, int k) { return __shfl(camv, k, 64); }
// table entries e < K PER: camera k = e / PER, entry i = e - PER k = (i < NA) ? A[6 c + i] : B[27 c + i - NA], to dst[k STRIDE + i]
template <int PER, int STRIDE, int NA>
__device__ __forceinline__ void gram_gather(const double* __restrict__ A, const double* __restrict__ B, int camv, int K, int lane, double* __restrict__ dst) {
    constexpr int NIT = (GRAM_KMAX * PER + 63) / 64;
    double v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int e = lane + 64 * it, k = e / PER, i = e - PER * k;
        const int c = gram_cam_of(camv, min(k, K - 1));                 // (past the table: a valid address, the value is dropped)
        const double* src = (i < NA) ? A + 6 * (size_t)c + i : B + 27 * (size_t)c + (i - NA);
        v[it] = *src;
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int e = lane + 64 * it, k = e / PER, i = e - PER * k;
        if (e < K * PER) dst[k * STRIDE + i] = v[it];
    }
}
// Round 5, FUSE (every point of the problem sits in a signature group): the kernel also does k_point_lin's work -- the lanes of a point fold their V = sum Jp^T Jp, g_p and
// focal coupling over the K observation lanes (three xor exchanges), every lane damps and inverts the 3x3 block itself, lane 0 of the point stores the record PS and g_p
// for the back substitution and the wave adds the five point-pass sums (cost, focal sums) and the gradient maximum to its scalar slot at the end.  k_point_lin does not
// run then (15.5 us of a 173 us iteration at config 2, 180 of 1550 at the configs[4] size), PS is not read back and the observations are read one pass less.
// spec (speculative launch behind k_publish, like k_point_lin's): [go, radius] as decided on the device.
struct GramFuse { const double* scale_pt; double radius, min_diag, max_diag; double* PS_out; double* gp_out; double* scal; const double* spec; int emit_skip = 0; };
// one wave task of k_schur_gram / k_schur_gram_any (below): the task's tile shape (NT, TI) is a template parameter, the task index an argument
template <int DC, int NT, int TI, bool FUSE>
__device__ __forceinline__ void
schur_gram_task(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts, const double* __restrict__ focal,
                const double2* __restrict__ obs_xy, const int* __restrict__ gr_rec, const double* __restrict__ scale_cam, const double* __restrict__ scale_f,
                const double* __restrict__ PS, int loss, double la, int rows_alloc, int focal_free, const int task, double* __restrict__ S_val, double* __restrict__ rhs,
                double* __restrict__ Udiag, double* __restrict__ Sfc, double* __restrict__ gcraw, long long* __restrict__ dbg, const GramFuse& fz, const DetZone& dz) {
    double fz_radius = fz.radius;
    if (FUSE && fz.spec) { if (fz.spec[0] == 0.0) return; fz_radius = fz.spec[1]; }
    constexpr int BB = DC * DC, off = (DC == 6) ? 0 : 3;                // NT = row tiles of 16 in use: the launch covers the tasks with 16 (NT - 1) < DC K <= 16 NT
    constexpr int NU = DC * (DC + 1) / 2, NS = NU + 3 * DC, NO = (NS + 7) / 8;          // camera-side sums: [Jc^T Jc (upper) | Jc^T r | -Jc^T Jp V^-1 g | Jc^T (J_f - Jp V^-1 w_f)]
    typedef double v4d_ __attribute__((ext_vector_type(4)));
    const long long t_0 = dbg ? wall_clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) double sY[];          // per wave: [rows_alloc][GRAM_LD] | camera records [GRAM_KMAX][GRAM_CAMREC] | scales | slots | diagonal slots
    const int lane = threadIdx.x & 63;
    double* sYw = sY + (size_t)(threadIdx.x >> 6) * (rows_alloc * GRAM_LD + GRAM_TAIL);
    double* sCam = sYw + rows_alloc * GRAM_LD;
    double* sScale = sCam + GRAM_KMAX * GRAM_CAMREC;                     // [DC K] Jacobi scale of Gram row DC k + d
    int* sSlot = (int*)(sScale + 48);                                    // [GRAM_NPAIR] pair slots, then [GRAM_KMAX] diagonal blocks
    int* sDiag = sSlot + GRAM_NPAIR;
    const int* rec = gr_rec + (size_t)task * GRAM_REC;                   // one record per task (ba_flatten.h): wave-uniform, scalar loads
    const int p0 = __builtin_amdgcn_readfirstlane(rec[0]), cnt = __builtin_amdgcn_readfirstlane(rec[1]), K = __builtin_amdgcn_readfirstlane(rec[2]);
    const int j00 = __builtin_amdgcn_readfirstlane(rec[3]);
    const double f = focal[0];
    const int camv = rec[4 + (lane & (GRAM_KMAX - 1))];                 // the task's cameras as a lane vector (gram_cam_of)
    gram_gather<33, GRAM_CAMREC, 6>(cam, rot, camv, K, lane, sCam);
    { const int k = min(lane / DC, K - 1), c = gram_cam_of(camv, k); const double sc = scale_cam[6 * c + off + lane - DC * (lane / DC)]; if (lane < DC * K) sScale[lane] = sc; }
    if (lane < GRAM_NPAIR + GRAM_KMAX) sSlot[lane] = rec[12 + lane];
    v4d_ acc[NT * (NT + 1) / 2];
#pragma unroll
    for (int t = 0; t < NT * (NT + 1) / 2; t++) acc[t] = v4d_{0.0, 0.0, 0.0, 0.0};
    // TI > 0: 16 NT < DC K <= 16 NT + 4 (six cameras: 36 rows at 6 dof, 18 at 3).  The last rows would cost NT + 1 more 16x16 products per k-step (64 cycles each) for
    // a handful of useful entries; they go through v_mfma_f64_4x4x4_4b_f64 instead (22 cycles, four 4x4 blocks per instruction): rows 16 NT .. + 3 against the
    // ceil(DC K / 4) column groups of four = TI instructions.  Block b of instruction u is column group g = 4 u + b; operands A_b[i][k] = Y[16 NT + i][k] on lane
    // i + 4 b + 16 k, B_b[k][j] = Y[4 g + j][k] on lane j + 4 b + 16 k, result D_b[i][j] on lane j + 4 b + 16 i (profiles/r03_notes.md).  Rows / columns >= DC K hold
    // whatever LDS holds and only reach entries that are dropped.
    double tacc[TI > 0 ? TI : 1] = {};
    double sm[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) sm[i] = 0.0;
    const int li = lane & 15, lk = lane >> 4;                          // fragment / accumulator coordinates of the tiles
    const int lp = lane & (GRAM_SUB - 1), lq = lane >> 3;               // linearisation: point of the sub-chunk, observation (= camera of the group) of this lane
    const int kq = min(lq, K - 1);                                      // clamped: a lane whose camera does not exist repeats the last one, stores nothing and sums nothing
    const double sf = scale_f[0];
    // point record of a sub-chunk: X, the scaled V^-1 (6) with V^-1 g (3) and V^-1 w_f (3), this lane's observation
    double X[3], V[12]; double2 ob;
    double spt[3] = {0.0, 0.0, 0.0};                                     // FUSE: Jacobi scales of the point (what k_point_lin read)
    double pacc[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, pgmax = 0.0;             // FUSE: cost, FJJ, FJR, FWW, FWG of this lane; gradient maximum
#define GRAM_LOAD(s0_)                                                                                                            \
    do {                                                                                                                          \
        const int q_ = min((s0_) + lp, cnt - 1);                                                                                  \
        _Pragma("unroll") for (int k = 0; k < 3; k++) X[k] = pts[3 * (size_t)(p0 + q_) + k];                                      \
        if (FUSE) { _Pragma("unroll") for (int k = 0; k < 3; k++) spt[k] = fz.scale_pt[3 * (size_t)(p0 + q_) + k]; }              \
        else {                                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < 9; k++) V[k] = PS[12 * (size_t)(p0 + q_) + k];                                      \
        if (focal_free) { _Pragma("unroll") for (int k = 9; k < 12; k++) V[k] = PS[12 * (size_t)(p0 + q_) + k]; }                 \
        }                                                                                                                         \
        ob = obs_xy[j00 + (size_t)q_ * K + kq];                                                                                   \
    } while (0)
    GRAM_LOAD(0);
    long long t_1 = 0, t_2 = 0;
    for (int s0 = 0; s0 < cnt; s0 += GRAM_SUB) {
        wave_lds_handover();                                             // the tiles of the previous sub-chunk have read sY (first pass: sCam is written)
        {
            const bool valid = s0 + lp < cnt;
            const double* crec = sCam + kq * GRAM_CAMREC;
            if (FUSE) {
                // the point pass (k_point_lin): this lane's observation through the point-side view of the linearisation FIRST (r, J_f, J_p: 11 values; the camera blocks
                // are formed afterwards, so that they are not alive during the fold -- together with the 39 camera sums and the tile accumulators they spilled 49-80 registers),
                // folded over the K lanes of its point
                const double wgt = (valid && lq < K) ? 1.0 : 0.0;
                double pv[12];
                {
                    ObsPoint Lp; lin_obs_point(f, crec, crec + 6, X, ob.x, ob.y, loss, la, Lp);
                    double j0[2], j1[2], j2[2], jf[2];
#pragma unroll
                    for (int a = 0; a < 2; a++) { j0[a] = Lp.Jp[a][0] * spt[0] * wgt; j1[a] = Lp.Jp[a][1] * spt[1] * wgt; j2[a] = Lp.Jp[a][2] * spt[2] * wgt; jf[a] = Lp.Jf[a] * sf * wgt; }
                    pv[0] = j0[0] * j0[0] + j0[1] * j0[1]; pv[1] = j0[0] * j1[0] + j0[1] * j1[1]; pv[2] = j0[0] * j2[0] + j0[1] * j2[1];
                    pv[3] = j1[0] * j1[0] + j1[1] * j1[1]; pv[4] = j1[0] * j2[0] + j1[1] * j2[1]; pv[5] = j2[0] * j2[0] + j2[1] * j2[1];
                    pv[6] = j0[0] * Lp.r[0] + j0[1] * Lp.r[1]; pv[7] = j1[0] * Lp.r[0] + j1[1] * Lp.r[1]; pv[8] = j2[0] * Lp.r[0] + j2[1] * Lp.r[1];
                    pv[9] = jf[0] * j0[0] + jf[1] * j0[1]; pv[10] = jf[0] * j1[0] + jf[1] * j1[1]; pv[11] = jf[0] * j2[0] + jf[1] * j2[1];
                    pacc[0] += wgt * Lp.half_rho; pacc[1] += jf[0] * jf[0] + jf[1] * jf[1]; pacc[2] += jf[0] * Lp.r[0] + jf[1] * Lp.r[1];
                }
#pragma unroll
                for (int i = 0; i < 12; i++) { double t = pv[i]; t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64); pv[i] = t; }
                double Vd[6] = {pv[0], pv[1], pv[2], pv[3], pv[4], pv[5]};
                if (spt[0] > 0.0) {
                    if (valid && lq == 0) pgmax = fmax(pgmax, fmax(fabs(pv[6] / spt[0]), fmax(fabs(pv[7] / spt[1]), fabs(pv[8] / spt[2]))));
                    Vd[0] += fmin(fmax(Vd[0], fz.min_diag), fz.max_diag) / fz_radius;
                    Vd[3] += fmin(fmax(Vd[3], fz.min_diag), fz.max_diag) / fz_radius;
                    Vd[5] += fmin(fmax(Vd[5], fz.min_diag), fz.max_diag) / fz_radius;
                } else { Vd[0] = Vd[3] = Vd[5] = 1.0; }                   // constant point: identity block, zero coupling
                double Vi[6]; sym3_inverse(Vd, Vi);
                const double u0 = pv[9] * Vi[0] + pv[10] * Vi[1] + pv[11] * Vi[2];
                const double u1 = pv[9] * Vi[1] + pv[10] * Vi[3] + pv[11] * Vi[4];
                const double u2 = pv[9] * Vi[2] + pv[10] * Vi[4] + pv[11] * Vi[5];
                V[0] = Vi[0] * spt[0] * spt[0]; V[1] = Vi[1] * spt[0] * spt[1]; V[2] = Vi[2] * spt[0] * spt[2];
                V[3] = Vi[3] * spt[1] * spt[1]; V[4] = Vi[4] * spt[1] * spt[2]; V[5] = Vi[5] * spt[2] * spt[2];
                V[6] = spt[0] * (Vi[0] * pv[6] + Vi[1] * pv[7] + Vi[2] * pv[8]); V[7] = spt[1] * (Vi[1] * pv[6] + Vi[3] * pv[7] + Vi[4] * pv[8]);
                V[8] = spt[2] * (Vi[2] * pv[6] + Vi[4] * pv[7] + Vi[5] * pv[8]);
                V[9] = spt[0] * u0; V[10] = spt[1] * u1; V[11] = spt[2] * u2;
                if (valid && lq == 0) {
                    pacc[3] += u0 * pv[9] + u1 * pv[10] + u2 * pv[11];
                    pacc[4] += u0 * pv[6] + u1 * pv[7] + u2 * pv[8];
                    double* ps = fz.PS_out + 12 * (size_t)(p0 + s0 + lp);
#pragma unroll
                    for (int k = 0; k < 12; k++) ps[k] = V[k];
#pragma unroll
                    for (int k = 0; k < 3; k++) fz.gp_out[3 * (size_t)(p0 + s0 + lp) + k] = pv[6 + k];
                }
            }
            ObsLin Lk; lin_obs<DC == 6>(f, crec, crec + 6, X, ob.x, ob.y, loss, la, Lk);
            // Cholesky factor of the scaled V^-1 (all zero for a fixed point or a lane past the end of the task: its columns of Y are zero)
            double L00 = 0, L10 = 0, L20 = 0, L11 = 0, L21 = 0, L22 = 0;
            if (valid && V[0] > 0.0) {
                const double i0 = fast_rsqrt(V[0]);
                L00 = V[0] * i0; L10 = V[1] * i0; L20 = V[2] * i0;
                const double d1 = V[3] - L10 * L10;
                if (d1 > 0.0) {
                    const double i1 = fast_rsqrt(d1);
                    L11 = d1 * i1; L21 = (V[4] - L20 * L10) * i1;
                    const double d2 = V[5] - L20 * L20 - L21 * L21;
                    if (d2 > 0.0) L22 = d2 * fast_rsqrt(d2);
                }
            }
            double Jc[2][DC]; cam_block_raw<DC>(Lk, Jc);
            double T[2][3];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                T[r][0] = Lk.Jp[r][0] * L00 + Lk.Jp[r][1] * L10 + Lk.Jp[r][2] * L20;
                T[r][1] = Lk.Jp[r][1] * L11 + Lk.Jp[r][2] * L21;
                T[r][2] = Lk.Jp[r][2] * L22;
            }
            if (lq < K) {
                double* dst = sYw + (size_t)(DC * kq) * GRAM_LD + 3 * lp;
#pragma unroll
                for (int d = 0; d < DC; d++)
#pragma unroll
                    for (int cc = 0; cc < 3; cc++) {
                        double y = 0.0;
                        if (!jc_zero<DC>(0, d)) y += Jc[0][d] * T[0][cc];
                        if (!jc_zero<DC>(1, d)) y += Jc[1][d] * T[1][cc];
                        dst[d * GRAM_LD + cc] = y;
                    }
                if (valid) {                                             // camera-side sums of camera kq (k_cam_sums2's, from the same linearisation)
                    double yv[2], zv[2];
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        yv[r] = Lk.Jp[r][0] * V[6] + Lk.Jp[r][1] * V[7] + Lk.Jp[r][2] * V[8];
                        zv[r] = focal_free ? Lk.Jf[r] * sf - (Lk.Jp[r][0] * V[9] + Lk.Jp[r][1] * V[10] + Lk.Jp[r][2] * V[11]) : 0.0;
                    }
                    int u = 0;
#pragma unroll
                    for (int a2 = 0; a2 < DC; a2++) {
#pragma unroll
                        for (int b2 = a2; b2 < DC; b2++) {
                            if (!jc_zero<DC>(0, a2) && !jc_zero<DC>(0, b2)) sm[u] += Jc[0][a2] * Jc[0][b2];
                            if (!jc_zero<DC>(1, a2) && !jc_zero<DC>(1, b2)) sm[u] += Jc[1][a2] * Jc[1][b2];
                            u++;
                        }
#pragma unroll
                        for (int r = 0; r < 2; r++)
                            if (!jc_zero<DC>(r, a2)) {
                                sm[NU + a2] += Jc[r][a2] * Lk.r[r];
                                sm[NU + DC + a2] -= Jc[r][a2] * yv[r];
                                if (focal_free) sm[NU + 2 * DC + a2] += Jc[r][a2] * zv[r];
                            }
                    }
                }
            }
        }
        if (dbg && s0 == 0) t_1 = wall_clock64();                       // first sub-chunk linearised
        if (s0 + GRAM_SUB < cnt) GRAM_LOAD(s0 + GRAM_SUB);              // in flight while the tiles run
        wave_lds_handover();
        // tiles of the lower triangle: G(ti, tj) += Y(ti rows) Y(tj rows)^T over the 24 columns of this sub-chunk, 4 per instruction
        // (only the row tiles that hold cameras: the tasks are sorted by K and every tile count has its own launch -- DC K <= 16: one product per k-step,
        // <= 32: three, else six.  Guards in one loop cost the full case 6 %; two or three loop variants in one kernel 35-40 %: the accumulators left their AGPRs)
#pragma unroll
        for (int st = 0; st < 3 * GRAM_SUB / 4; st++) {
            double fr[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) fr[t] = sYw[(size_t)min(16 * t + li, rows_alloc - 1) * GRAM_LD + 4 * st + lk];
            int tix = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
#pragma unroll
                for (int tj = 0; tj <= ti; tj++) { acc[tix] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[ti], fr[tj], acc[tix], 0, 0, 0); tix++; }
        }
        if constexpr (TI > 0) {
            double a4[3 * GRAM_SUB / 4], b4[TI][3 * GRAM_SUB / 4];
#pragma unroll
            for (int st = 0; st < 3 * GRAM_SUB / 4; st++) {
                a4[st] = sYw[(size_t)min(16 * NT + (lane & 3), rows_alloc - 1) * GRAM_LD + 4 * st + lk];
#pragma unroll
                for (int u = 0; u < TI; u++) {
                    const int g = 4 * u + ((lane >> 2) & 3);                                   // column group of this lane's block
                    b4[u][st] = sYw[(size_t)min(4 * g + (lane & 3), rows_alloc - 1) * GRAM_LD + 4 * st + lk];
                }
            }
#pragma unroll
            for (int st = 0; st < 3 * GRAM_SUB / 4; st++)
#pragma unroll
                for (int u = 0; u < TI; u++) tacc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[st], b4[u][st], tacc[u], 0, 0, 0);
        }
    }
#undef GRAM_LOAD
    if (dbg) t_2 = wall_clock64();
    if (fz.emit_skip && (task % fz.emit_skip) != 0) { if (dbg && lane == 0) { long long* d = dbg + 4 * (size_t)task; d[0] = t_0; d[1] = t_1; d[2] = t_2; d[3] = wall_clock64(); } return; }   // EXPERIMENT (lab)
    // ---- emission (round 6: by BLOCKS, through LDS).  Rounds 3-5 went from the accumulators straight to the atomics: every (tile, register) entry worked out its camera
    // pair, slot and transposition with integer divisions behind divergent guards -- ~130 instructions per entry, ~2000 per task, 7-8 us of a 39 us task with nothing else
    // on its SIMD (SSFM_GRAM_STAMPS: a single emitting task took 8.3 us).  Now the scaled, negated tiles go to LDS as the lower triangle of G (bands of 16 rows: band t
    // has 16 (t + 1) + 1 columns; a store is lane coordinates + a constant), Jc^T Jc of every camera is added to its diagonal block there, and the K (K + 1) / 2 blocks
    // leave in block order: entry e of block (a, b) is G[6 a + e / DC][6 b + e % DC] -- consecutive lanes, consecutive addresses of S.
    wave_lds_handover();
    double* sG = sYw;                                                    // over sY and the camera records (both done): gram_g_off(DC K) <= rows_alloc GRAM_LD + GRAM_KMAX GRAM_CAMREC
    auto g_off = [](int R) { const int t = R >> 4; return 128 * t * t + 144 * t + (R & 15) * (16 * t + 17); };
    {
        double cs[NT], rs[NT][4];
#pragma unroll
        for (int t = 0; t < NT; t++) { cs[t] = sScale[min(16 * t + li, DC * GRAM_KMAX - 1)];
#pragma unroll
                                       for (int reg = 0; reg < 4; reg++) rs[t][reg] = sScale[min(16 * t + lk + 4 * reg, DC * GRAM_KMAX - 1)]; }
        int tix = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++) {
#pragma unroll
                for (int reg = 0; reg < 4; reg++)                        // (rows / columns >= DC K: whatever the scales' padding gives -- no block reads them)
                    if (16 * ti + lk + 4 * reg < rows_alloc) sG[(128 * ti * ti + 144 * ti) + (lk + 4 * reg) * (16 * ti + 17) + 16 * tj + li] = -acc[tix][reg] * rs[ti][reg] * cs[tj];
                tix++;
            }
        if constexpr (TI > 0) {
            const int R = 16 * NT + (lane >> 4);
            const double rsc = sScale[min(R, DC * GRAM_KMAX - 1)];
#pragma unroll
            for (int u = 0; u < TI; u++) {
                const int C = 16 * u + ((lane >> 2) & 3) * 4 + (lane & 3);
                sG[(128 * NT * NT + 144 * NT) + (lane >> 4) * (16 * NT + 17) + C] = -tacc[u] * rsc * sScale[min(C, DC * GRAM_KMAX - 1)];
            }
        }
    }
    // atomics-free emission (round 6, det_acc.h: DetZone::part): this task's stretch of the partial buffer -- blocks, then the cameras' vectors
    double* const PB = dz.part ? dz.part + dz.part_off[task] : nullptr;
    double* const PV = PB ? PB + (K * (K + 1) / 2) * BB : nullptr;
    const int cam_k = gram_cam_of(camv, kq);                             // (every lane takes part in the exchange)
    wave_lds_handover();
    // camera-side sums: fold over the 8 point lanes of every camera; the lane that holds value i adds it where k_cam_sums2 would -- except Jc^T Jc, which joins the
    // diagonal block of its camera in G (lower triangle: one owner lane per entry)
    {
        double out[NO];
        OctTR<NS, 4>::run(sm, out);
        if (lq < K) {
            const int c = cam_k;
            const double* sc = sScale + DC * lq;
            const int i0 = 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
#pragma unroll
            for (int j = 0; j < NO; j++) {
                const int i = 8 * j + i0;
                const double v = out[j];
                if (i < NU) {
                    int a = 0, rem = i; while (rem >= DC - a) { rem -= DC - a; a++; }
                    const int b = a + rem;                                // b >= a: the entry (row DC lq + b, column DC lq + a) of the lower triangle
                    const double w = v * sc[a] * sc[b];
                    sG[g_off(DC * lq + b) + DC * lq + a] += w;
                    if (b == a) { if (PV) PV[lq * 5 * DC + a] = w; else zadd(dz, &Udiag[c * DC + a], w); }
                } else if (i < NU + DC) { const int a = i - NU; if (PV) { PV[lq * 5 * DC + DC + a] = v * sc[a]; PV[lq * 5 * DC + 3 * DC + a] = v * sc[a]; } else { zadd(dz, &rhs[c * DC + a], v * sc[a]); zadd(dz, &gcraw[c * DC + a], v * sc[a]); } }
                else if (i < NU + 2 * DC) { const int a = i - NU - DC; if (PV) PV[lq * 5 * DC + 2 * DC + a] = v * sc[a]; else zadd(dz, &rhs[c * DC + a], v * sc[a]); }
                else if (i < NS && focal_free) { const int a = i - NU - 2 * DC; if (PV) PV[lq * 5 * DC + 4 * DC + a] = v * sc[a]; else zadd(dz, &Sfc[c * DC + a], v * sc[a]); }
            }
        }
    }
    wave_lds_handover();
    if constexpr (2 * BB > 64) {
        // one block per instruction on lanes [0, BB): the block's cameras, slot and orientation are wave-uniform (scalar loop counters, one broadcast LDS read),
        // the lane's entry (da, db) never changes -- an iteration is a dozen vector instructions and one atomic over BB consecutive doubles of S
        const int e = min(lane, BB - 1), da = e / DC, db = e - da * DC, eT = db * DC + da;
        const bool on = lane < BB;
        for (int a = 0; a < K; a++) {
            const int Ra = DC * a + da;
            for (int b = 0; b <= a; b++) {
                const int C = DC * b + db;
                const double v = sG[(b < a) ? g_off(Ra) + C : g_off(max(Ra, C)) + min(Ra, C)];
                const int sl = (b < a) ? sSlot[a * (a - 1) / 2 + b] : sDiag[a];
                const int eo = (b < a && (sl & (1 << 30))) ? eT : e;      // the block as S stores it
                if (on) {
                    if (PB) PB[(a * (a + 1) / 2 + b) * BB + eo] = v;       // (fold lists, ba_flatten.h: block (a, b) of the task at (a (a + 1) / 2 + b) BB)
                    else zadd(dz, &(S_val + (size_t)(sl & 0x3fffffff) * BB)[eo], v);
                }
            }
        }
    } else {
        const int nent = (K * (K + 1) / 2) * BB;
        for (int idx = lane; idx < nent; idx += 64) {
            const int blk = idx / BB, e = idx - blk * BB, da = e / DC, db = e - da * DC;
            const int a = (int)((__fsqrt_rn((float)(8 * blk + 1)) - 1.0f) * 0.5f);        // blk = a (a + 1) / 2 + b, b <= a (8 blk + 1 = (2 a + 1)^2 + 8 b: exact in float)
            const int b = blk - a * (a + 1) / 2;
            const int R = DC * a + da, C = DC * b + db;
            const double v = sG[g_off(max(R, C)) + min(R, C)];           // (max / min: the upper triangle of a diagonal block)
            if (b < a) {
                const int sl = sSlot[a * (a - 1) / 2 + b];
                const int eo = (sl & (1 << 30)) ? (db * DC + da) : e;      // the block as S stores it
                if (PB) PB[blk * BB + eo] = v;
                else zadd(dz, &(S_val + (size_t)(sl & 0x3fffffff) * BB)[eo], v);
            } else if (PB) PB[idx] = v;
            else zadd(dz, &(S_val + (size_t)sDiag[a] * BB)[e], v);
        }
    }
    if (FUSE) {                                                          // the point pass's sums and gradient maximum, one wave = one scalar slot (k_point_lin's rule)
        const double t = wave_transpose_sum(pacc);
        pgmax = wave_max(pgmax);
        const int slot = wave_tr_index();
        double* sl = fz.scal + (size_t)(task & (SC_NSLOT - 1)) * SC_TOTAL;
        if (slot < 5) unsafeAtomicAdd(&sl[slot], t);
        if (lane == 0 && pgmax > 0.0) atomic_max_nonneg(&sl[SC_GMAX], pgmax);
    }
    if (dbg && lane == 0) { long long* d = dbg + 4 * (size_t)task; d[0] = t_0; d[1] = t_1; d[2] = t_2; d[3] = wall_clock64(); }   // SSFM_GRAM_STAMPS (timing study)
}

// one launch per tile class: tasks [task0, ntasks) all have 16 (NT - 1) < DC K <= 16 NT (+ 4 with TI > 0)
template <int DC, int NT, int TI = 0, bool FUSE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))      // <= 256 registers (vector + accumulation): at one wave per SIMD config 2's 1800 tasks need two rounds
k_schur_gram(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts, const double* __restrict__ focal,
             const double2* __restrict__ obs_xy, int ntasks, const int* __restrict__ gr_rec, const double* __restrict__ scale_cam, const double* __restrict__ scale_f,
             const double* __restrict__ PS, int loss, double la, int rows_alloc, int focal_free, int task0, double* __restrict__ S_val, double* __restrict__ rhs,
             double* __restrict__ Udiag, double* __restrict__ Sfc, double* __restrict__ gcraw, long long* __restrict__ dbg, GramFuse fz = GramFuse{}, DetZone dz = DetZone{}) {
    const int task = task0 + __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));   // [task0, ntasks): this launch's tile class
    if (task >= ntasks) return;
    schur_gram_task<DC, NT, TI, FUSE>(cam, rot, pts, focal, obs_xy, gr_rec, scale_cam, scale_f, PS, loss, la, rows_alloc, focal_free, task, S_val, rhs, Udiag, Sfc, gcraw, dbg, fz, dz);
}
// Round 5: ALL tile classes in one launch -- tracks of mixed length (a real sequence: 3 ... 8 cameras per point) gave one launch per class, each with the ~25 us latency
// floor of a wave task, one after the other on the stream (4 x 25 us against 44 + 21 us through the pair lists: the planner refused the groups, profiles/r04_notes.md
// r04d).  Here every wave picks the instantiation of its task's class (wave-uniform branch on K; registers and LDS = those of the largest class), so the classes run
// side by side and the launch costs the longest task once.  A problem with ONE class (config 2, configs[4]) keeps the specialised kernel above.
template <int DC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
k_schur_gram_any(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts, const double* __restrict__ focal,
                 const double2* __restrict__ obs_xy, int ntasks, const int* __restrict__ gr_rec, const double* __restrict__ scale_cam, const double* __restrict__ scale_f,
                 const double* __restrict__ PS, int loss, double la, int rows_alloc, int focal_free, int use_t4, double* __restrict__ S_val, double* __restrict__ rhs,
                 double* __restrict__ Udiag, double* __restrict__ Sfc, double* __restrict__ gcraw, DetZone dz = DetZone{}) {
    // the tasks are sorted by K ascending: the longest ones (most tiles) first
    const int task = ntasks - 1 - __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (task < 0) return;
    const int rows = DC * __builtin_amdgcn_readfirstlane(gr_rec[(size_t)task * GRAM_REC + 2]);
    const GramFuse fz{};
#define SSFM_GRAM_ANY(NT_, TI_) schur_gram_task<DC, NT_, TI_, false>(cam, rot, pts, focal, obs_xy, gr_rec, scale_cam, scale_f, PS, loss, la, rows_alloc, focal_free, task, S_val, rhs, Udiag, Sfc, gcraw, nullptr, fz, dz)
    if (rows <= 16) SSFM_GRAM_ANY(1, 0);
    else if (rows <= 20 && use_t4) SSFM_GRAM_ANY(1, 2);
    else if (rows <= 32) SSFM_GRAM_ANY(2, 0);
    else if (DC == 6 && rows <= 36 && use_t4) SSFM_GRAM_ANY(2, 3);
    else if (DC == 6) SSFM_GRAM_ANY(3, 0);
#undef SSFM_GRAM_ANY
}

// EXPERIMENT (SSFM_PAIRS_Y_PROBE=1, not part of the solve): the pair pass if every observation carried a stored half product
// Y_i = Jc_i^T Jp_i L (DC x 3, V^-1 = L L^T) -- a pair would be blk -= Y_i Y_j^T (DC*DC*3 multiply-adds) and two DC*3*8-byte reads instead of two
// re-linearisations.  Same wave tasks, same slot folds and atomics as k_schur_pairs2, into a scratch copy of S; Y holds arbitrary data.
// Measured next to the real kernel under rocprofv3 (profiles/r02_pairs_y_probe.md): the reads lose what the arithmetic saves.
template <int DC>
__global__ void __launch_bounds__(256, 3)
k_pairs_y_probe(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const int* __restrict__ task_cam, const int* __restrict__ task_b0,
                const int* __restrict__ task_b1, int ntasks, const int* __restrict__ batch_slot, const int* __restrict__ pair_j,
                const int* __restrict__ pair_j2, const double* __restrict__ scale_cam, const double* __restrict__ Y, double* __restrict__ S_scratch) {
    constexpr int BB = DC * DC, YW = DC * 3;
    const int lane = threadIdx.x & 63;
    const int task = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (task >= ntasks) return;
    const int c = __builtin_amdgcn_readfirstlane(task_cam[task]);
    const int b0 = __builtin_amdgcn_readfirstlane(task_b0[task]), b1 = __builtin_amdgcn_readfirstlane(task_b1[task]);
    const int rb = __builtin_amdgcn_readfirstlane(row_ptr[c]);
    double blk[DC][DC];
#pragma unroll
    for (int a = 0; a < DC; a++)
#pragma unroll
        for (int b = 0; b < DC; b++) blk[a][b] = 0.0;
    int cur = -1;
    const int tr_slot = wave_tr_index();
    auto fold = [&]() {
        double* dst = S_scratch + ((size_t)rb + cur) * BB;
        double flat[BB];
#pragma unroll
        for (int a = 0; a < DC; a++)
#pragma unroll
            for (int b = 0; b < DC; b++) { flat[a * DC + b] = blk[a][b]; blk[a][b] = 0.0; }
        const double v = wave_transpose_sum(flat);
        if (tr_slot < BB) {
            constexpr int off = (DC == 6) ? 0 : 3;
            const int a = tr_slot / DC, b = tr_slot - a * DC, c2f = col_idx[rb + cur];
            unsafeAtomicAdd(&dst[tr_slot], v * scale_cam[6 * c + off + a] * scale_cam[6 * c2f + off + b]);
        }
    };
#define PY_LOAD(bt_, A_, B_, wgt_)                                                                                    \
    do {                                                                                                              \
        const size_t e_ = (size_t)(bt_) * 64 + lane;                                                                  \
        const int j_ = pair_j[e_], j2_ = pair_j2[e_];                                                                 \
        wgt_ = (j_ >= 0) ? 1.0 : 0.0;                                                                                 \
        const double2* ya_ = reinterpret_cast<const double2*>(Y + (size_t)max(j_, 0) * YW);                           \
        const double2* yb_ = reinterpret_cast<const double2*>(Y + (size_t)max(j2_, 0) * YW);                          \
        _Pragma("unroll") for (int k = 0; k < YW / 2; k++) { const double2 t_ = ya_[k]; A_[2 * k] = t_.x; A_[2 * k + 1] = t_.y; }  \
        _Pragma("unroll") for (int k = 0; k < YW / 2; k++) { const double2 t_ = yb_[k]; B_[2 * k] = t_.x; B_[2 * k + 1] = t_.y; }  \
    } while (0)
#define PY_COMPUTE(bt_, A_, B_, wgt_)                                                                                 \
    do {                                                                                                              \
        const int slot_ = __builtin_amdgcn_readfirstlane(batch_slot[bt_]);                                            \
        if (slot_ != cur) { if (cur >= 0) fold(); cur = slot_; }                                                      \
        _Pragma("unroll") for (int a = 0; a < DC; a++) {                                                              \
            const double a0 = A_[3 * a] * wgt_, a1 = A_[3 * a + 1] * wgt_, a2 = A_[3 * a + 2] * wgt_;                 \
            _Pragma("unroll") for (int b = 0; b < DC; b++) blk[a][b] -= a0 * B_[3 * b] + a1 * B_[3 * b + 1] + a2 * B_[3 * b + 2];  \
        }                                                                                                             \
    } while (0)
    double Aa[YW], Ba[YW], wa, Ab[YW], Bb[YW], wb;
    PY_LOAD(b0, Aa, Ba, wa);
    for (int bt = b0; bt < b1; bt += 2) {
        PY_LOAD(min(bt + 1, b1 - 1), Ab, Bb, wb);
        PY_COMPUTE(bt, Aa, Ba, wa);
        PY_LOAD(min(bt + 2, b1 - 1), Aa, Ba, wa);
        if (bt + 1 < b1) PY_COMPUTE(bt + 1, Ab, Bb, wb);
    }
#undef PY_LOAD
#undef PY_COMPUTE
    if (cur >= 0) fold();
}

// symmetric mat-vec with only the lower triangle stored: q_c = sum_{s in row c} S_s p_col(s) + sum_{t in trans(c)} S_t^T p_row(t)
template <int DC>
__global__ void __launch_bounds__(256)
k_sym_matvec(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const int* __restrict__ trans_ptr,
             const int* __restrict__ trans_blk, const int* __restrict__ trans_row, const double* __restrict__ S_val,
             const double* __restrict__ Sfc, const double* __restrict__ p, int Nc, const double* __restrict__ pcg,
             double* __restrict__ q, double* __restrict__ pqpart) {
    if (pcg[PCG_DONE] != 0.0) return;
    __shared__ double part[4][64];
    constexpr int BB = DC * DC;
    constexpr int LW = (64 / DC) * DC;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + w;
    if (c < Nc) {
        const int rb = row_ptr[c], nnb = row_ptr[c + 1] - rb, tb = trans_ptr[c], nt = trans_ptr[c + 1] - tb;
        double s = 0.0;
        if (lane < LW) {
            const int a = lane % DC;
            for (int idx = lane; idx < (nnb + nt) * DC; idx += LW) {
                const int b = idx / DC;
                if (b < nnb) {
                    const double* row = S_val + ((size_t)(rb + b)) * BB + a * DC;
                    const double* pv = p + col_idx[rb + b] * DC;
#pragma unroll
                    for (int k = 0; k < DC; k++) s += row[k] * pv[k];
                } else {
                    const int t = tb + (b - nnb);
                    const double* blk = S_val + (size_t)trans_blk[t] * BB;      // block (row r, col c): use its transpose
                    const double* pv = p + trans_row[t] * DC;
#pragma unroll
                    for (int k = 0; k < DC; k++) s += blk[k * DC + a] * pv[k];
                }
            }
        }
        part[w][lane] = s;
    }
    __syncthreads();
    if (c < Nc && lane < DC) {
        double s = 0.0;
        for (int l = lane; l < LW; l += DC) s += part[w][l];
        s += Sfc[c * DC + lane] * p[Nc * DC];
        q[c * DC + lane] = s;
        part[w][lane] = s * p[c * DC + lane];
    }
    __syncthreads();
    if (c < Nc && lane == 0) { double s = 0.0; for (int a = 0; a < DC; a++) s += part[w][a]; pqpart[c] = s; }
}

// focal arrow + q = S x in one launch: every workgroup recomputes phi = (rho - S_fc.V) / (S_ff - S_fc.U) (two short dot products),
// forms x = [V - U phi ; phi] on the fly (V, U in elimination order) and multiplies its four block rows; x is written out for the
// kernels that follow.  Replaces k_band_combine + k_sym_matvec for the lower-triangle storage.
template <int DC>
__global__ void __launch_bounds__(256)
k_arrow_matvec(const double* __restrict__ V, const double* __restrict__ U, const double* __restrict__ Sfc, const double* __restrict__ Sff,
               const double* __restrict__ rho_ptr, const int* __restrict__ pos, const int* __restrict__ row_ptr, const int* __restrict__ col_idx,
               const int* __restrict__ trans_ptr, const int* __restrict__ trans_blk, const int* __restrict__ trans_row,
               const double* __restrict__ S_val, int Nc, double* __restrict__ x, double* __restrict__ q) {
    __shared__ double red[2 * 4];
    __shared__ double part[4][64];
    __shared__ double sphi;
    constexpr int BB = DC * DC;
    constexpr int LW = (64 / DC) * DC;
    const int n = Nc * DC, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc[2] = {0, 0};
    for (int t = threadIdx.x; t < n; t += blockDim.x) { const int c = t / DC, a = t - c * DC; const int pi = pos[c] * DC + a; acc[0] += Sfc[t] * V[pi]; acc[1] += Sfc[t] * U[pi]; }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) sphi = (rho_ptr[0] - acc[0]) / (Sff[0] - acc[1]);
    __syncthreads();
    const double phi = sphi;
    const int c = blockIdx.x * 4 + w;
    if (c < Nc) {
        const int rb = row_ptr[c], nnb = row_ptr[c + 1] - rb, tb = trans_ptr[c], nt = trans_ptr[c + 1] - tb;
        double s = 0.0;
        if (lane < LW) {
            const int a = lane % DC;
            for (int idx = lane; idx < (nnb + nt) * DC; idx += LW) {
                const int b = idx / DC;
                const int col = (b < nnb) ? col_idx[rb + b] : trans_row[tb + (b - nnb)];
                const int pc = pos[col] * DC;
                double xv[DC];
#pragma unroll
                for (int k = 0; k < DC; k++) xv[k] = V[pc + k] - U[pc + k] * phi;
                if (b < nnb) {
                    const double* row = S_val + ((size_t)(rb + b)) * BB + a * DC;
#pragma unroll
                    for (int k = 0; k < DC; k++) s += row[k] * xv[k];
                } else {
                    const double* blk = S_val + (size_t)trans_blk[tb + (b - nnb)] * BB;      // block (row r, col c): use its transpose
#pragma unroll
                    for (int k = 0; k < DC; k++) s += blk[k * DC + a] * xv[k];
                }
            }
        }
        part[w][lane] = s;
    }
    __syncthreads();
    if (c < Nc && lane < DC) {
        double s = 0.0;
        for (int l = lane; l < LW; l += DC) s += part[w][l];
        q[c * DC + lane] = s + Sfc[c * DC + lane] * phi;
        const int pi = pos[c] * DC + lane;
        x[c * DC + lane] = V[pi] - U[pi] * phi;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) x[n] = phi;
}

// ---- K2b: after the (optional) all-reduce: LM diagonal on the camera blocks, block-Jacobi inverse, focal row
template <int DC>
__global__ void k_finalize_S(const int* __restrict__ row_ptr, const int* __restrict__ diag_slot, const double* __restrict__ scale_cam,
                             const double* __restrict__ scale_f, const double* __restrict__ Udiag, const double* __restrict__ gcraw,
                             double radius, double min_diag, double max_diag, int Nc,
                             double* __restrict__ S_val, double* __restrict__ Minv, double* __restrict__ rhs,
                             double* __restrict__ Sff, double* __restrict__ scal) {
    constexpr int BB = DC * DC; constexpr int off = (DC == 6) ? 0 : 3;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    double gmax = 0.0;
    if (c < Nc) {
        double* blk = S_val + ((size_t)row_ptr[c] + diag_slot[c]) * BB;
        double A[DC][DC];
#pragma unroll
        for (int a = 0; a < DC; a++) {
            const double s = scale_cam[c * 6 + off + a];
            const double d2 = (s > 0.0) ? fmin(fmax(Udiag[c * DC + a], min_diag), max_diag) / radius : 1.0;
            blk[a * DC + a] += d2;
            if (s > 0.0) gmax = fmax(gmax, fabs(gcraw[c * DC + a] / s));
        }
#pragma unroll
        for (int a = 0; a < DC; a++)
#pragma unroll
            for (int b = 0; b < DC; b++) A[a][b] = blk[a * DC + b];
        // in-register Cholesky A = L L^T, then inverse = L^-T L^-1
        double Lm[DC][DC];
#pragma unroll
        for (int i = 0; i < DC; i++)
#pragma unroll
            for (int j = 0; j < DC; j++) Lm[i][j] = 0.0;
#pragma unroll
        for (int j = 0; j < DC; j++) {
            double d = A[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= Lm[j][k] * Lm[j][k];
            d = sqrt(d); Lm[j][j] = d;
#pragma unroll
            for (int i = j + 1; i < DC; i++) {
                double s = A[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) s -= Lm[i][k] * Lm[j][k];
                Lm[i][j] = s / d;
            }
        }
        double Li[DC][DC];   // L^-1 (lower)
#pragma unroll
        for (int i = 0; i < DC; i++)
#pragma unroll
            for (int j = 0; j < DC; j++) Li[i][j] = 0.0;
#pragma unroll
        for (int j = 0; j < DC; j++) {
            Li[j][j] = 1.0 / Lm[j][j];
#pragma unroll
            for (int i = j + 1; i < DC; i++) {
                double s = 0.0;
#pragma unroll
                for (int k = j; k < i; k++) s -= Lm[i][k] * Li[k][j];
                Li[i][j] = s / Lm[i][i];
            }
        }
#pragma unroll
        for (int a = 0; a < DC; a++)
#pragma unroll
            for (int b = 0; b < DC; b++) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < DC; k++) s += Li[k][a] * Li[k][b];   // (L^-T L^-1)[a][b]; terms with k < max(a,b) are zero
                Minv[(size_t)c * BB + a * DC + b] = s;
            }
    }
    gmax = wave_max(gmax);
    if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&scal[SC_GMAX], gmax);
    if (blockIdx.x == 0 && threadIdx.x < 64) {             // wave 0 of workgroup 0 folds the replicas of the focal sums (blockDim = 64)
        const double* sl = scal + (size_t)(threadIdx.x & (SC_NSLOT - 1)) * SC_TOTAL;
        const double fjj = wave_sum(sl[SC_FJJ]), fww = wave_sum(sl[SC_FWW]), fjr = wave_sum(sl[SC_FJR]), fwg = wave_sum(sl[SC_FWG]);
        if (threadIdx.x == 0) {
            const double sf = scale_f[0];
            if (sf > 0.0) {
                Sff[0] = fjj + fmin(fmax(fjj, min_diag), max_diag) / radius - fww;
                rhs[Nc * DC] = fjr - fwg;
                atomic_max_nonneg(&scal[SC_GMAX], fabs(fjr / sf));
            } else { Sff[0] = 1.0; rhs[Nc * DC] = 0.0; }
        }
    }
}

// ---- PCG on the reduced system (block-CSR S + dense focal border), block-Jacobi preconditioner -------
// One wave per block row for the mat-vec; all vector work + both dot products in ONE single-workgroup
// kernel, so an iteration is two launches and no host round trip; a device flag ends the loop.
template <int DC>
__global__ void __launch_bounds__(1024)
k_pcg_init(const double* __restrict__ rhs, const double* __restrict__ Minv, const double* __restrict__ Sff, int Nc,
           double* __restrict__ x, double* __restrict__ r, double* __restrict__ z, double* __restrict__ p, double* __restrict__ pcg) {
    __shared__ double red[2 * 16];
    const int n = Nc * DC;
    double acc[2] = {0, 0};
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        double zi;
        if (i < n) {
            const int c = i / DC, a = i - c * DC; zi = 0.0;
#pragma unroll
            for (int b = 0; b < DC; b++) zi += Minv[(size_t)c * DC * DC + a * DC + b] * rhs[c * DC + b];
        } else zi = rhs[n] / Sff[0];
        const double ri = rhs[i];
        x[i] = 0.0; r[i] = ri; z[i] = zi; p[i] = zi;
        acc[0] += ri * zi; acc[1] += ri * ri;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        pcg[PCG_RZ] = acc[0]; pcg[PCG_BN2] = acc[1]; pcg[PCG_RR] = acc[1]; pcg[PCG_ITERS] = 0.0; pcg[PCG_BREAKDOWN] = 0.0;
        pcg[PCG_DONE] = (acc[1] == 0.0) ? 1.0 : 0.0;
    }
}

template <int DC>
__global__ void __launch_bounds__(256)
k_pcg_matvec(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const double* __restrict__ S_val,
             const double* __restrict__ Sfc, const double* __restrict__ p, int Nc, const double* __restrict__ pcg,
             double* __restrict__ q, double* __restrict__ pqpart) {
    if (pcg[PCG_DONE] != 0.0) return;
    __shared__ double part[4][64];
    constexpr int BB = DC * DC;
    constexpr int LW = (64 / DC) * DC;          // lanes used: a multiple of DC so that a lane keeps its row index
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + w;
    if (c < Nc) {
        const int rb = row_ptr[c], nnb = row_ptr[c + 1] - rb;
        double s = 0.0;
        if (lane < LW) {
            const int a = lane % DC;
            for (int idx = lane; idx < nnb * DC; idx += LW) {
                const int b = idx / DC;
                const double* row = S_val + ((size_t)(rb + b)) * BB + a * DC;
                const double* pv = p + col_idx[rb + b] * DC;
#pragma unroll
                for (int k = 0; k < DC; k++) s += row[k] * pv[k];
            }
        }
        part[w][lane] = s;
    }
    __syncthreads();
    if (c < Nc && lane < DC) {
        double s = 0.0;
        for (int l = lane; l < LW; l += DC) s += part[w][l];
        const double pf = p[Nc * DC];
        s += Sfc[c * DC + lane] * pf;
        q[c * DC + lane] = s;
        part[w][lane] = s * p[c * DC + lane];
    }
    __syncthreads();
    if (c < Nc && lane == 0) { double s = 0.0; for (int a = 0; a < DC; a++) s += part[w][a]; pqpart[c] = s; }
}

template <int DC>
__global__ void __launch_bounds__(1024)
k_pcg_vecops(const double* __restrict__ Minv, const double* __restrict__ Sfc, const double* __restrict__ Sff, int Nc, double tol2,
             double* __restrict__ x, double* __restrict__ r, double* __restrict__ z, double* __restrict__ p,
             const double* __restrict__ q, const double* __restrict__ pqpart, double* __restrict__ pcg) {
    if (pcg[PCG_DONE] != 0.0) return;
    __shared__ double red[2 * 16];
    __shared__ double sh[2];
    const int n = Nc * DC;
    const double pf = p[n];
    // q_f = Sff p_f + Sfc . p_c ;  pq = sum_c pqpart + p_f q_f
    double acc[2] = {0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc[0] += Sfc[i] * p[i];
    for (int c = threadIdx.x; c < Nc; c += blockDim.x) acc[1] += pqpart[c];
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        const double qf = Sff[0] * pf + acc[0];
        const double pq = acc[1] + pf * qf;
        sh[0] = qf; sh[1] = pq;
    }
    __syncthreads();
    const double qf = sh[0], pq = sh[1];
    const double rz = pcg[PCG_RZ];
    if (!(pq > 0.0)) { if (threadIdx.x == 0) { pcg[PCG_DONE] = 1.0; pcg[PCG_BREAKDOWN] = 1.0; } return; }
    const double alpha = rz / pq;
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        const double qi = (i < n) ? q[i] : qf;
        x[i] += alpha * p[i];
        r[i] -= alpha * qi;
    }
    __syncthreads();
    double acc2[2] = {0, 0};
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        double zi;
        if (i < n) {
            const int c = i / DC, a = i - c * DC; zi = 0.0;
#pragma unroll
            for (int b = 0; b < DC; b++) zi += Minv[(size_t)c * DC * DC + a * DC + b] * r[c * DC + b];
        } else zi = r[n] / Sff[0];
        z[i] = zi;
        acc2[0] += r[i] * zi; acc2[1] += r[i] * r[i];
    }
    block_sum<2>(acc2, red);
    if (threadIdx.x == 0) { sh[0] = acc2[0]; sh[1] = acc2[1]; }
    __syncthreads();
    const double rz_new = sh[0], rr = sh[1];
    const double beta = rz_new / rz;
    for (int i = threadIdx.x; i <= n; i += blockDim.x) p[i] = z[i] + beta * p[i];
    if (threadIdx.x == 0) {
        pcg[PCG_RZ] = rz_new; pcg[PCG_RR] = rr; pcg[PCG_ITERS] += 1.0;
        if (rr <= tol2 * pcg[PCG_BN2]) pcg[PCG_DONE] = 1.0;
    }
}

// ---- Schur pair lists on the device (what ba_flatten.h: pair_counts_host / pair_fill_host do on the host) ---------------------
// One thread per camera-major entry q = (observation j of camera c, point p): every other observation j2 of p by a camera c2 that
// precedes c in the elimination order is one pair of row c, slot = position of c2 in row c of S (columns sorted by camera id).
// k_pair_count counts per slot; the host lays out the 64-padded batches (pair_layout); k_pair_fill writes (j, j2, p) through atomic
// cursors that start at the slots' first entries (the order inside a slot is arbitrary: the sums over a slot go through atomics anyway).
// camera-major view of the point-major observation list: cursor[c] starts at cam_start[c]; the order inside a camera is arbitrary
static __global__ void __launch_bounds__(256)
k_cam_lists(int M, const int* __restrict__ obs_cam, const int* __restrict__ obs_pt, unsigned int* __restrict__ cursor, int* __restrict__ cam_obs,
            int* __restrict__ cam_obs_pt) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const unsigned int w = atomicAdd(&cursor[obs_cam[j]], 1u);
    cam_obs[w] = j; cam_obs_pt[w] = obs_pt[j];
}
// The same for <= 4096 cameras with one global atomic per (workgroup, camera) instead of one per observation: a workgroup ranks its
// 4096 observations per camera in LDS, reserves every camera's run with one add, then writes.  (600k same-address-heavy atomics on
// 300 cursors took 0.55 ms at config 2, 12M on 4000 took 6 ms.)
constexpr int CAM_LISTS_MAX_CAMS = 4096, CAM_LISTS_CHUNK = 4096;
static __global__ void __launch_bounds__(256)
k_cam_lists_agg(int M, int Nc, const int* __restrict__ obs_cam, const int* __restrict__ obs_pt, unsigned int* __restrict__ cursor,
                int* __restrict__ cam_obs, int* __restrict__ cam_obs_pt) {
    __shared__ unsigned int cnt[CAM_LISTS_MAX_CAMS];
    constexpr int PER = CAM_LISTS_CHUNK / 256;
    for (int c = threadIdx.x; c < Nc; c += blockDim.x) cnt[c] = 0u;
    __syncthreads();
    const int j0 = blockIdx.x * CAM_LISTS_CHUNK + threadIdx.x;
    int cc[PER]; unsigned int rk[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) { const int j = j0 + u * 256; if (j < M) { cc[u] = obs_cam[j]; rk[u] = atomicAdd(&cnt[cc[u]], 1u); } }
    __syncthreads();
    for (int c = threadIdx.x; c < Nc; c += blockDim.x) { const unsigned int n = cnt[c]; if (n) cnt[c] = atomicAdd(&cursor[c], n); }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++) { const int j = j0 + u * 256; if (j < M) { const unsigned int w = cnt[cc[u]] + rk[u]; cam_obs[w] = j; cam_obs_pt[w] = obs_pt[j]; } }
}
// ... then every camera's segment is put back in ascending observation order (= ascending point order: neighbouring lanes of
// k_cam_sums2 read neighbouring points; the unsorted lists cost it 2 us per launch): bitonic sort in LDS, segments of <= 4096 entries
static __global__ void __launch_bounds__(256)
k_cam_lists_sort(const int* __restrict__ cam_start, const int* __restrict__ obs_pt, int* __restrict__ cam_obs, int* __restrict__ cam_obs_pt) {
    __shared__ int sv[4096];
    const int c = blockIdx.x, q0 = cam_start[c], n = cam_start[c + 1] - q0;
    if (n <= 1 || n > 4096) return;
    int np2 = 1; while (np2 < n) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += blockDim.x) sv[i] = (i < n) ? cam_obs[q0 + i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = threadIdx.x; i < np2; i += blockDim.x) {
                const int l = i ^ jj;
                if (l > i) {
                    const int a = sv[i], b = sv[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { sv[i] = b; sv[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const int j = sv[i]; cam_obs[q0 + i] = j; cam_obs_pt[q0 + i] = obs_pt[j]; }
}
// A workgroup's 256 camera-major entries belong to one or two cameras, so its pairs fall into a short contiguous run of slots
// [row_ptr[first camera], row_ptr[last camera + 1]): they are counted in LDS, every slot's run is reserved with ONE global add, and
// (FILL) a second walk over the same pairs writes them at run start + LDS rank.  One global atomic per pair (all lanes of a
// workgroup on the same ~10 counters) took 0.47 + 0.54 ms at config 2 and 4.9 + 9.5 ms at the configs[4] size.  Runs longer
// than PAIR_LISTS_SLOTS (many tiny cameras in one workgroup) keep the direct atomics.
constexpr int PAIR_LISTS_SLOTS = 2048;
template <bool FILL>
static __global__ void __launch_bounds__(256)
k_pair_lists(int M, const int* __restrict__ cam_obs, const int* __restrict__ cam_obs_pt, const int* __restrict__ obs_cam,
             const int* __restrict__ pt_start, const int* __restrict__ elim_pos, const int* __restrict__ row_ptr, const int* __restrict__ col_idx,
             unsigned int* __restrict__ slot_ctr, int* __restrict__ pair_j, int* __restrict__ pair_j2, int* __restrict__ pair_p,
             const unsigned char* __restrict__ pt_skip = nullptr /* points whose blocks come from k_schur_gram */) {
    __shared__ unsigned int cnt[PAIR_LISTS_SLOTS];
    __shared__ unsigned int base[FILL ? PAIR_LISTS_SLOTS : 1];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int q_first = blockIdx.x * blockDim.x, q_last = min(M, q_first + (int)blockDim.x) - 1;
    const int s0 = row_ptr[obs_cam[cam_obs[q_first]]], s1 = row_ptr[obs_cam[cam_obs[q_last]] + 1];     // camera-major: cameras ascend with q
    const bool local = (s1 - s0) <= PAIR_LISTS_SLOTS;
    if (local) { for (int i = threadIdx.x; i < s1 - s0; i += blockDim.x) cnt[i] = 0u; __syncthreads(); }
    int j = 0, p = 0, c = 0, pc = 0, rb = 0, nnb = 0, ja = 0, jb = 0;
    if (q < M) { j = cam_obs[q]; p = cam_obs_pt[q]; c = obs_cam[j]; pc = elim_pos[c]; rb = row_ptr[c]; nnb = row_ptr[c + 1] - rb; ja = pt_start[p]; jb = pt_start[p + 1];
                 if (pt_skip && pt_skip[p]) jb = ja; }
    for (int j2 = ja; j2 < jb; j2++) {
        const int c2 = obs_cam[j2];
        if (!(elim_pos[c2] < pc)) continue;
        int lo = 0, hi = nnb;                                    // first column >= c2
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (col_idx[rb + mid] < c2) lo = mid + 1; else hi = mid; }
        if (local) atomicAdd(&cnt[rb + lo - s0], 1u);
        else {
            const unsigned int w = atomicAdd(&slot_ctr[rb + lo], 1u);
            if (FILL) { pair_j[w] = j; pair_j2[w] = j2; pair_p[w] = p; }
        }
    }
    if (!local) return;                                          // (uniform over the workgroup)
    __syncthreads();
    for (int i = threadIdx.x; i < s1 - s0; i += blockDim.x) {
        const unsigned int n = cnt[i];
        if (n) { const unsigned int b = atomicAdd(&slot_ctr[s0 + i], n); if (FILL) base[i] = b; }
        if (FILL) cnt[i] = 0u;
    }
    if (!FILL) return;
    __syncthreads();
    for (int j2 = ja; j2 < jb; j2++) {
        const int c2 = obs_cam[j2];
        if (!(elim_pos[c2] < pc)) continue;
        int lo = 0, hi = nnb;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (col_idx[rb + mid] < c2) lo = mid + 1; else hi = mid; }
        const int sl = rb + lo - s0;
        const unsigned int w = base[sl] + atomicAdd(&cnt[sl], 1u);
        pair_j[w] = j; pair_j2[w] = j2; pair_p[w] = p;
    }
}

// ---- K3a: candidate cameras/focal from the PCG solution (replicated on every rank) --------------------
template <int DC>
__global__ void __launch_bounds__(1024)
k_cam_update(const double* __restrict__ cam, const double* __restrict__ focal, const double* __restrict__ scale_cam,
             const double* __restrict__ scale_f, const double* __restrict__ y, int Nc,
             double* __restrict__ cam_c, double* __restrict__ focal_c, double* __restrict__ rot_c, double* __restrict__ scal, int with_rot) {
    __shared__ double red[2 * 16];
    constexpr int off = (DC == 6) ? 0 : 3;
    double acc[2] = {0, 0};
    for (int i = threadIdx.x; i < Nc * 6; i += blockDim.x) {
        const int c = i / 6, k = i - c * 6;
        double v = cam[i];
        const double s = scale_cam[i];
        if (s > 0.0 && k >= off) { const double d = -y[c * DC + (k - off)] * s; v += d; acc[0] += d * d; acc[1] += v * v; }
        cam_c[i] = v;
    }
    if (threadIdx.x == 0) {
        double v = focal[0]; const double s = scale_f[0];
        if (s > 0.0) { const double d = -y[Nc * DC] * s; v += d; acc[0] += d * d; acc[1] += v * v; }
        focal_c[0] = v;
    }
    block_sum<2>(acc, red);                                       // (its barriers also publish cam_c to the whole workgroup)
    if (threadIdx.x == 0) { scal[SC_STEP2_CAM] = acc[0]; scal[SC_XN2_CAM] = acc[1]; }
    // rotation tables of the candidate cameras (what a separate k_cam_rot launch did; large camera sets leave them to k_cam_rot again:
    // one workgroup walking 4000 cameras took 64 us)
    if (with_rot) for (int c = threadIdx.x; c < Nc; c += blockDim.x) {
        const double aa[3] = {cam_c[c * 6 + 3], cam_c[c * 6 + 4], cam_c[c * 6 + 5]};
        double R[9], Rd[9], M[9];
        angle_axis_derivative_aid(aa, R, Rd, M);
        for (int i = 0; i < 9; i++) { rot_c[c * 27 + i] = R[i]; rot_c[c * 27 + 9 + i] = Rd[i]; rot_c[c * 27 + 18 + i] = M[i]; }
    }
}

// The two dot products of the focal step phi = (rho - S_fc . v) / (S_ff - S_fc . u) in `gridDim.x` contiguous parts, for camera sets
// where the per-workgroup recomputation in k_arrow_update (O(Nc) per workgroup, O(Nc^2) in total: 70 us at 4000 cameras) costs more than
// a launch.  The parts are summed in a fixed order by the reader: every rank of a multi-GPU solve gets the same bits.
template <int DC>
__global__ void __launch_bounds__(256)
k_arrow_phi(const double* __restrict__ V, const double* __restrict__ U, const double* __restrict__ Sfc, const int* __restrict__ pos, int Nc,
            double* __restrict__ part) {
    __shared__ double red[2 * 4];
    const int n = Nc * DC, per = (n + gridDim.x - 1) / gridDim.x, t0 = blockIdx.x * per, t1 = min(n, t0 + per);
    double acc[2] = {0, 0};
    for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) { const int c = t / DC, a = t - c * DC; const int pi = pos[c] * DC + a; acc[0] += Sfc[t] * V[pi]; acc[1] += Sfc[t] * U[pi]; }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = acc[0]; part[2 * blockIdx.x + 1] = acc[1]; }
}

// ---- K3a': k_arrow_matvec and k_cam_update in one launch (the usual tail of an LM iteration): every wave owns one camera -- its row of
// q = S x for the residual check, its part of the step x, its candidate parameters and their rotation tables; the two camera norms
// go to the replicated scalar slots by one atomic pair per workgroup.  Saves a single-workgroup launch (7 us) per iteration.
// Round 4: TWO waves per camera (512 threads, four cameras per workgroup) -- the row of q = S x (four dependent gathers) on one, the candidate parameters and their rotation
// tables (a serial sincos chain on one lane) on the other: the kernel's time is the longer of the two instead of their sum (9.3 -> see profiles/r04_notes.md us at config 2).
template <int DC>
__global__ void __launch_bounds__(512)
k_arrow_update(const double* __restrict__ V, const double* __restrict__ U, const double* __restrict__ Sfc, const double* __restrict__ Sff,
               const double* __restrict__ rho_ptr, const int* __restrict__ pos, const int* __restrict__ row_ptr, const int* __restrict__ col_idx,
               const int* __restrict__ trans_ptr, const int* __restrict__ trans_blk, const int* __restrict__ trans_row,
               const double* __restrict__ S_val, int Nc, double* __restrict__ x, double* __restrict__ q,
               const double* __restrict__ cam, const double* __restrict__ focal, const double* __restrict__ scale_cam,
               const double* __restrict__ scale_f, double* __restrict__ cam_c, double* __restrict__ focal_c, double* __restrict__ rot_c,
               double* __restrict__ scal, const double* __restrict__ phi_part, int phi_parts, const int* __restrict__ col_pos, const int* __restrict__ trans_pos,
               long long* __restrict__ lacc = nullptr) {
    const DetScal ds{scal, lacc};
    __shared__ double red[2 * 8];
    __shared__ double part[4][64];
    __shared__ double part2[4][2];
    __shared__ double sphi;
    constexpr int BB = DC * DC;
    constexpr int LW = (64 / DC) * DC;
    constexpr int off = (DC == 6) ? 0 : 3;
    const int n = Nc * DC, w = (threadIdx.x >> 6) & 3, role = threadIdx.x >> 8, lane = threadIdx.x & 63;      // role 0: the matvec row, role 1: candidate + rotation tables
    double phi;
    if (phi_parts < 0) phi = 0.0;                                   // focal fixed: S_fc = 0, S_ff = 1, rho = 0 -- no block-wide dot products, no barrier
    else if (phi_parts > 0) {                                       // large camera sets: the dot products come from k_arrow_phi
        double a0 = 0.0, a1 = 0.0;
        for (int b = 0; b < phi_parts; b++) { a0 += phi_part[2 * b]; a1 += phi_part[2 * b + 1]; }
        phi = (rho_ptr[0] - a0) / (Sff[0] - a1);
    } else {                                                          // small ones: every workgroup recomputes it (no extra launch)
        double acc[2] = {0, 0};
        for (int t = threadIdx.x; t < n; t += blockDim.x) { const int c = t / DC, a = t - c * DC; const int pi = pos[c] * DC + a; acc[0] += Sfc[t] * V[pi]; acc[1] += Sfc[t] * U[pi]; }
        block_sum<2>(acc, red);
        if (threadIdx.x == 0) sphi = (rho_ptr[0] - acc[0]) / (Sff[0] - acc[1]);
        __syncthreads();
        phi = sphi;
    }
    const int c = blockIdx.x * 4 + w;
    if (lane < 2 && role == 1) part2[w][lane] = 0.0;
    if (c < Nc && role == 0) {
        const int rb = row_ptr[c], nnb = row_ptr[c + 1] - rb, tb = trans_ptr[c], nt = trans_ptr[c + 1] - tb;
        double s = 0.0;
        if (lane < LW) {
            const int a = lane % DC;
            for (int idx = lane; idx < (nnb + nt) * DC; idx += LW) {
                const int b = idx / DC;
                const int pc = ((b < nnb) ? col_pos[rb + b] : trans_pos[tb + (b - nnb)]) * DC;      // (= pos[col_idx[..]] / pos[trans_row[..]], precomputed: one dependent gather less)
                double xv[DC];
#pragma unroll
                for (int k = 0; k < DC; k++) xv[k] = V[pc + k] - U[pc + k] * phi;
                if (b < nnb) {
                    const double* row = S_val + ((size_t)(rb + b)) * BB + a * DC;
#pragma unroll
                    for (int k = 0; k < DC; k++) s += row[k] * xv[k];
                } else {
                    const double* blk = S_val + (size_t)trans_blk[tb + (b - nnb)] * BB;      // block (row r, col c): use its transpose
#pragma unroll
                    for (int k = 0; k < DC; k++) s += blk[k * DC + a] * xv[k];
                }
            }
        }
        part[w][lane] = s;
    }
    if (c < Nc && role == 1) {
        // candidate camera: lanes 0..5 hold its six parameters
        double v = 0.0, d2 = 0.0, v2 = 0.0;
        if (lane < 6) {
            v = cam[c * 6 + lane];
            const double sc = scale_cam[c * 6 + lane];
            if (sc > 0.0 && lane >= off) {
                const int pi = pos[c] * DC + (lane - off);
                const double d = -(V[pi] - U[pi] * phi) * sc;
                v += d; d2 = d * d; v2 = v * v;
            }
            cam_c[c * 6 + lane] = v;
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { d2 += __shfl_xor(d2, o, 64); v2 += __shfl_xor(v2, o, 64); }
        const double aa[3] = {__shfl(v, 3, 64), __shfl(v, 4, 64), __shfl(v, 5, 64)};
        if (lane == 0) {
            part2[w][0] = d2; part2[w][1] = v2;
            double R[9], Rd[9], M[9];
            angle_axis_derivative_aid(aa, R, Rd, M);
            for (int i = 0; i < 9; i++) { rot_c[c * 27 + i] = R[i]; rot_c[c * 27 + 9 + i] = Rd[i]; rot_c[c * 27 + 18 + i] = M[i]; }
        }
    }
    __syncthreads();
    if (c < Nc && lane < DC && role == 0) {
        double s = 0.0;
        for (int l = lane; l < LW; l += DC) s += part[w][l];
        q[c * DC + lane] = s + Sfc[c * DC + lane] * phi;
        const int pi = pos[c] * DC + lane;
        x[c * DC + lane] = V[pi] - U[pi] * phi;
    }
    if (threadIdx.x == 0) {
        double a0 = part2[0][0] + part2[1][0] + part2[2][0] + part2[3][0], a1 = part2[0][1] + part2[1][1] + part2[2][1] + part2[3][1];
        if (blockIdx.x == 0) {
            x[n] = phi;
            double vf = focal[0]; const double sf = scale_f[0];
            if (sf > 0.0) { const double d = -phi * sf; vf += d; a0 += d * d; a1 += vf * vf; }
            focal_c[0] = vf;
        }
        double* sl = scal_slot(scal);
        sadd(ds, &sl[SC_STEP2_CAM], a0); sadd(ds, &sl[SC_XN2_CAM], a1);
    }
}

// LmGate: what the host knows when it queues the end of iteration k -- enough for the device to decide "step accepted, go on" (see k_publish below)
struct LmGate {
    int enabled, last_successful;
    double radius, x_norm, function_tolerance, gradient_tolerance, parameter_tolerance, min_relative_decrease, max_radius, min_radius;
};
__device__ __forceinline__ double coherent_load(const double* p) {          // straight from L2: the value another workgroup's atomics / stores left there
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)b);
}
// The end-of-iteration hand-over (fold of the scalar replicas, solver flags, the device's accept decision, sequence number into pinned host memory) as a
// function any workgroup of >= 64 threads can run: k_publish is one launch of it, k_point_backsub's last workgroup runs it in place of that launch.
__device__ __forceinline__ void publish_body(const double* __restrict__ scal, const double* __restrict__ pcg, double* __restrict__ host_out, unsigned long long seq,
                                             const LmGate& gate, double* __restrict__ spec, double* folded /* LDS [SC_TOTAL] */,
                                             long long* __restrict__ lacc = nullptr, unsigned kmask = 0,
                                             // values the calling workgroup has just computed itself (LDS: no trip through global memory inside the kernel): the
                                             // scalars k with bit k of ov_mask set, and the solver words [0, PCG_TOTAL)
                                             const double* ov_scal = nullptr, unsigned ov_mask = 0, const double* ov_pcg = nullptr) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    auto pcg_word = [&](int i) { return (ov_pcg && i < PCG_TOTAL) ? ov_pcg[i] : coherent_load(pcg + i); };
    for (int k = w; k < SC_TOTAL; k += nw) {
        if (ov_scal && ((ov_mask >> k) & 1u)) { if (lane == 0) folded[k] = ov_scal[k]; continue; }
        if (lacc && ((kmask >> k) & 1u)) {
            // deterministic mode (det_acc.h): this scalar's sums sit in the long accumulators of the 64 replicas -- integer wave sums, then the value; the limbs are
            // cleared for the next iteration (what k_det_decode does in a launch of its own for the sums of the assembly)
            long long* a = lacc + ((size_t)lane * SC_TOTAL + k) * LA_STRIDE;
            long long tot[LA_STRIDE];
            // round 6: the eight limbs first, then the clears, then the sums -- load / clear / butterfly per limb in turn was eight dependent round trips per scalar
#pragma unroll
            for (int j = 0; j < LA_STRIDE; j++) tot[j] = __hip_atomic_load(a + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < LA_STRIDE; j++) if (tot[j] != 0) a[j] = 0;
#pragma unroll
            for (int j = 0; j < LA_STRIDE; j++) {
                long long v = tot[j];
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                tot[j] = v;
            }
            if (lane == 0) folded[k] = lacc_value(tot, tot[LA_NL]);
            continue;
        }
        const double v = coherent_load(scal + (size_t)(lane & (SC_NSLOT - 1)) * SC_TOTAL + k);
        const double r = (k == SC_GMAX) ? wave_max(v) : wave_sum(v);
        if (lane == 0) folded[k] = r;
    }
    __syncthreads();
    if (w == 0) {
        if (lane < SC_TOTAL) host_out[lane] = folded[lane];
        else if (lane < SC_TOTAL + PCG_TOTAL + 1) host_out[lane] = pcg_word(lane - SC_TOTAL);
        if (spec && lane == 0) {
            double go = 0.0, new_radius = gate.radius;
            if (gate.enabled) {
                int fail_word; { const double fw = coherent_load(pcg + PCG_TOTAL); __builtin_memcpy(&fail_word, &fw, sizeof(int)); }
                const double x_cost = folded[SC_COST], gmax = folded[SC_GMAX], model = -folded[SC_MODEL], cand = folded[SC_CAND_COST];
                bool ok = isfinite(x_cost) && !(gate.last_successful && gmax <= gate.gradient_tolerance);
                ok = ok && fail_word == 0 && pcg_word(PCG_DONE) != 0.0 && isfinite(model) && model > 0.0 && isfinite(cand);
                const double step_norm = sqrt(folded[SC_STEP2_PT] + folded[SC_STEP2_CAM]);
                ok = ok && !(step_norm <= gate.parameter_tolerance * (gate.x_norm + gate.parameter_tolerance));
                const double cost_change = x_cost - cand;
                ok = ok && !(fabs(cost_change) <= gate.function_tolerance * x_cost);
                const double rel = cost_change / model;
                ok = ok && rel > gate.min_relative_decrease;
                if (ok) {
                    const double t = 2.0 * rel - 1.0;
                    new_radius = fmin(gate.max_radius, gate.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
                    ok = new_radius > gate.min_radius;
                }
                go = ok ? 1.0 : 0.0;
            }
            spec[0] = go; spec[1] = new_radius;
            host_out[SC_TOTAL + PCG_TOTAL + 2] = go; host_out[SC_TOTAL + PCG_TOTAL + 3] = new_radius;
        }
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + SC_TOTAL + PCG_TOTAL + 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- K3b: back-substitution + model cost change + candidate points (one lane per point) ---------------
//   y_p = V^-1 (g_p - sum_j Jp_j^T (Jc_j y_c + Jf_j y_f)),  step = -y,  delta = scale o step
//   model = sum_j m_j (r_j + m_j / 2),  m_j = Jc_j step_c + Jf_j step_f + Jp_j step_p
// residual check of the reduced solve (what k_ref_residual does) by ONE workgroup: r = b - [q | S_fc.y + S_ff y_f], |r|^2 <= tol^2 |b|^2 ?  red: LDS scratch [16]
__device__ __forceinline__ void residual_check_body(int n, const double* __restrict__ y, const double* __restrict__ res_b, const double* __restrict__ res_q,
                                                    const double* __restrict__ res_Sfc, const double* __restrict__ res_Sff, double res_tol2,
                                                    double* __restrict__ res_r, double* __restrict__ res_pcg, double* red) {
    __shared__ double sqf;
    double a0[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) a0[0] += res_Sfc[i] * y[i];
    block_sum<1>(a0, red);
    if (threadIdx.x == 0) sqf = res_Sff[0] * y[n] + a0[0];
    __syncthreads();
    double a2[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        const double qi = (i < n) ? res_q[i] : sqf;
        const double ri = res_b[i] - qi; res_r[i] = ri;
        a2[0] += ri * ri; a2[1] += res_b[i] * res_b[i];
    }
    block_sum<2>(a2, red);
    if (threadIdx.x == 0) {
        res_pcg[PCG_RR] = a2[0]; res_pcg[PCG_BN2] = a2[1]; res_pcg[PCG_ITERS] = 0.0; res_pcg[PCG_BREAKDOWN] = 0.0;
        res_pcg[PCG_DONE] = (a2[0] <= res_tol2 * a2[1]) ? 1.0 : 0.0;
    }
}
// LPP = lanes per point (round 4 experiment, 1 in use): with 2, the pair splits the point's observations (interleaved), folds the eleven per-point sums with one quad-permute
// each and both lanes know the point's step -- half the dependent camera gathers per lane and twice the waves.  Slower at config 2 (25.2 against 23.2 us): the kernel is not
// bound by the length of a lane's chain of round trips (see also the unroll experiments below)
template <int DC, int LPP = 1>
__global__ void __launch_bounds__(256)
k_point_backsub(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
                const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
                const int* __restrict__ pt_start, int nP, const double* __restrict__ scale_cam, const double* __restrict__ scale_pt,
                const double* __restrict__ scale_f, const double* __restrict__ Vs, const double* __restrict__ gp,
                const double* __restrict__ y, int Nc, int loss, double la, const double* __restrict__ cam_c, const double* __restrict__ rot_c,
                const double* __restrict__ focal_c, double* __restrict__ pts_c, double* __restrict__ scal,
                // residual check of the reduced solve (what k_ref_residual does), by one extra workgroup behind the point workgroups when
                // res_r is given: the single-workgroup launch leaves the critical path of the iteration
                const double* __restrict__ res_b, const double* __restrict__ res_q, const double* __restrict__ res_Sfc,
                const double* __restrict__ res_Sff, double res_tol2, double* __restrict__ res_r, double* __restrict__ res_pcg,
                const unsigned char* __restrict__ pt_skip,          // points of signature groups: k_gram_backsub has them
                // fused hand-over (pub_host != nullptr): the workgroup that finishes LAST (a ticket) folds the scalars and publishes them, which
                // takes the k_publish launch (4.7 us in a dependent stream) off the iteration
                int* __restrict__ pub_ticket = nullptr, double* __restrict__ pub_host = nullptr, unsigned long long pub_seq = 0,
                const LmGate pub_gate = LmGate(), double* __restrict__ pub_spec = nullptr, long long* __restrict__ lacc = nullptr) {
    const DetScal ds{scal, lacc};
    __shared__ double red[4 * 4];
    const bool residual_block = res_r && blockIdx.x == gridDim.x - 1;
    if (residual_block) {
        residual_check_body(Nc * DC, y, res_b, res_q, res_Sfc, res_Sff, res_tol2, res_r, res_pcg, red);
        if (!pub_host) return;
    }
    const int gt = blockIdx.x * blockDim.x + threadIdx.x, p = gt / LPP, hl = gt - p * LPP;                 // hl: this lane's share of the point's observations
    double acc[4] = {0, 0, 0, 0};   // model, step2, xn2, candidate cost
    if (!residual_block && p < nP && !(pt_skip && pt_skip[p])) {
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double sp[3] = {scale_pt[3 * p], scale_pt[3 * p + 1], scale_pt[3 * p + 2]};
        const double f = focal[0], sf = scale_f[0], yf = y[Nc * DC];
        const double g3[3] = {gp[3 * p], gp[3 * p + 1], gp[3 * p + 2]};
        double b[3] = {0.0, 0.0, 0.0};
        const int j0 = pt_start[p], j1 = pt_start[p + 1];
        // ONE sweep over the observations: with M_j = a_j + B_j z  (a_j = camera/focal part of J s, B_j = Jp_j diag(s_p), z = point step)
        // the model cost change  -sum M_j.(r_j - M_j/2)  expands into sums that do not depend on z -- sum a.a, sum a.r, B^T a, B^T B --
        // so the second re-linearisation of every observation (after z is known) is not needed
        double Saa = 0.0, Sar = 0.0, Vr[6] = {0, 0, 0, 0, 0, 0};
        // (Groups of two or three observations with all their camera tables in flight were slower -- 21.3 / 23.7 us against 18.8 in
        // scripts/lab/point_lab.hip, as was a six- or eight-fold unroll of k_point_lin: the extra registers cost a wave per SIMD and the
        // kernel is not bound by the length of a lane's chain of round trips.)
        // the camera index and pixel of observation j+1 are fetched while observation j is processed: one dependent round trip
        // per observation (its camera tables) instead of two
        const int jf = min(j0 + hl, j1 - 1);
        int c_nx = obs_cam[jf]; double2 o_nx = obs_xy[jf];
        for (int j = j0 + hl; j < j1; j += LPP) {
            const int c = c_nx; const double2 o = o_nx;
            { const int jn = min(j + LPP, j1 - 1); c_nx = obs_cam[jn]; o_nx = obs_xy[jn]; }
            ObsLin L; lin_obs<DC == 6>(f, cam + 6 * c, rot + 27 * c, X, o.x, o.y, loss, la, L);
            double Jc[2][DC]; cam_block<DC>(L, scale_cam + 6 * c, Jc);
            double m0 = L.Jf[0] * sf * yf, m1 = L.Jf[1] * sf * yf;
#pragma unroll
            for (int a = 0; a < DC; a++) { const double ya = y[c * DC + a]; m0 += Jc[0][a] * ya; m1 += Jc[1][a] * ya; }
            Saa += m0 * m0 + m1 * m1; Sar += m0 * L.r[0] + m1 * L.r[1];
            const double B0[3] = {L.Jp[0][0] * sp[0], L.Jp[0][1] * sp[1], L.Jp[0][2] * sp[2]}, B1[3] = {L.Jp[1][0] * sp[0], L.Jp[1][1] * sp[1], L.Jp[1][2] * sp[2]};
            Vr[0] += B0[0] * B0[0] + B1[0] * B1[0]; Vr[1] += B0[0] * B0[1] + B1[0] * B1[1]; Vr[2] += B0[0] * B0[2] + B1[0] * B1[2];
            Vr[3] += B0[1] * B0[1] + B1[1] * B1[1]; Vr[4] += B0[1] * B0[2] + B1[1] * B1[2]; Vr[5] += B0[2] * B0[2] + B1[2] * B1[2];
#pragma unroll
            for (int k = 0; k < 3; k++) b[k] -= B0[k] * m0 + B1[k] * m1;
        }
        if (LPP == 2) {                                     // fold the pair: both lanes continue with the point's sums
            Saa += __shfl_xor(Saa, 1, 64); Sar += __shfl_xor(Sar, 1, 64);
#pragma unroll
            for (int k = 0; k < 6; k++) Vr[k] += __shfl_xor(Vr[k], 1, 64);
#pragma unroll
            for (int k = 0; k < 3; k++) b[k] += __shfl_xor(b[k], 1, 64);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) b[k] += g3[k];
        // V^-1 b from the record of the Schur kernels, PS_V = diag(s) V^-1 diag(s):  V^-1 b = s^-1 o (PS_V (s^-1 o b))
        const double* Vi = Vs + 12 * (size_t)p;
        double yp[3] = {0.0, 0.0, 0.0};
        if (sp[0] > 0.0) {
            const double is[3] = {fast_rcp(sp[0]), fast_rcp(sp[1]), fast_rcp(sp[2])};
            const double bs[3] = {b[0] * is[0], b[1] * is[1], b[2] * is[2]};
            yp[0] = (Vi[0] * bs[0] + Vi[1] * bs[1] + Vi[2] * bs[2]) * is[0];
            yp[1] = (Vi[1] * bs[0] + Vi[3] * bs[1] + Vi[4] * bs[2]) * is[1];
            yp[2] = (Vi[2] * bs[0] + Vi[4] * bs[1] + Vi[5] * bs[2]) * is[2];
        }
        {
            const double zg = yp[0] * g3[0] + yp[1] * g3[1] + yp[2] * g3[2];
            const double zBa = yp[0] * (g3[0] - b[0]) + yp[1] * (g3[1] - b[1]) + yp[2] * (g3[2] - b[2]);       // z . sum B^T a
            const double zVz = Vr[0] * yp[0] * yp[0] + Vr[3] * yp[1] * yp[1] + Vr[5] * yp[2] * yp[2] + 2.0 * (Vr[1] * yp[0] * yp[1] + Vr[2] * yp[0] * yp[2] + Vr[4] * yp[1] * yp[2]);
            if (hl == 0) acc[0] = -(Sar + zg) + 0.5 * (Saa + 2.0 * zBa + zVz);
        }
        double Xc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double d = -yp[k] * sp[k]; const double v = X[k] + d;
            Xc[k] = v;
            if (hl == 0) { pts_c[3 * p + k] = v; if (sp[k] > 0.0) { acc[1] += d * d; acc[2] += v * v; } }
        }
        // robustified cost at the candidate (cameras, focal, this point): what a separate k_point_cost launch did
        const double fcand = focal_c[0];
        c_nx = obs_cam[jf]; o_nx = obs_xy[jf];
        for (int j = j0 + hl; j < j1; j += LPP) {
            const int c = c_nx; const double2 o = o_nx;
            { const int jn = min(j + LPP, j1 - 1); c_nx = obs_cam[jn]; o_nx = obs_xy[jn]; }
            acc[3] += obs_cost(fcand, cam_c + 6 * c, rot_c + 27 * c, Xc, o.x, o.y, loss, la);
        }
    }
    // per-wave fold and atomics (see k_point_lin): no barrier at the end
    static_assert(SC_STEP2_PT == SC_MODEL + 1 && SC_XN2_PT == SC_MODEL + 2 && SC_CAND_COST == SC_MODEL + 3, "the four sums are consecutive scalars");
    const double t = wave_transpose_sum(acc);
    const int slot = wave_tr_index();
    double* sl = scal + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (SC_NSLOT - 1)) * SC_TOTAL;
    if (slot < 4 && !residual_block) sadd(ds, &sl[SC_MODEL + slot], t);
    if (pub_host) {
        __shared__ int s_last; __shared__ double folded[SC_TOTAL];
        __threadfence();                                    // this workgroup's sums / flags are in L2 before its ticket is
        __syncthreads();
        if (threadIdx.x == 0) s_last = (atomicAdd(pub_ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        if (s_last) {
            __threadfence();
            if (threadIdx.x == 0) *pub_ticket = 0;
            publish_body(scal, res_pcg, pub_host, pub_seq, pub_gate, pub_spec, folded);
        }
    }
}

// ---- back substitution + candidate + both costs for the points of SIGNATURE GROUPS (round 3; the kernel of rounds 3-5, k_gram_backsub, is in the history:
// lane = (point, observation) with the point's 8 lanes 8 apart, two register copies of a sub-chunk's point records, folds by ds_bpermute -- 258 us at the configs[4] size) ----
// What k_point_backsub does with a lane per point and one dependent camera-table gather per observation, in k_schur_gram's layout: wave task = a run of
// points with the same K <= 8 cameras, sub-chunks of 8 points, lane = (point, observation).  The K camera records (current and candidate) and the cameras'
// steps sit in LDS, every lane linearises ONE observation, what the point's step needs is folded over the 8 observation lanes of the point, every lane of the
// point then knows the step and evaluates ITS observation at the candidate.  No dependent index load anywhere (the observations of a group are consecutive,
// K per point); the next sub-chunk's records are in flight during the arithmetic.
constexpr int GBS_CAMC = 12, GBS_TAIL = GRAM_KMAX * (GRAM_CAMREC + GBS_CAMC + 6), GBS_WAVES = 4;
// ---- round 6: the point records staged through LDS and the 8 observation lanes of a point ADJACENT -------------------------------------
// k_gram_backsub kept two copies of a sub-chunk's point records in registers (the one in use and the one in flight: 2 x 17 doubles per lane, every value
// requested by the eight lanes of its point -- sixteen load instructions per sub-chunk) and folds B^T a over lanes 8 / 16 / 32 apart with ds_bpermute (three dependent
// LDS round trips).  Here lane = 8 point + observation: a sub-chunk's observations are ONE coalesced 1 KB load; its 8 point records come in through three loads shared
// by the wave, wait one iteration in registers, go to a wave-private LDS slice (two buffers, GBS2_PT doubles per point: the eight points' 16-byte reads fall into
// disjoint banks) and are read where they are used; the fold over the 8 lanes of a point is three DPP exchanges (quad_perm, quad_perm, row_half_mirror) without LDS.
// The point's step comes from the record PS alone: with w = sum over its observations of Jp^T m (m = the camera / focal part of the model residual) the step in the
// caller's coordinates is  u = PS[6..8] - PS[0..5] w   (PS[0..5] = S V^-1 S, PS[6..8] = S V^-1 g, S = the Jacobi scales: k_point_lin) -- no g_p, no scales, no
// reciprocal; a fixed point has an all-zero record.  Sums in another order than k_gram_backsub's (tests: oracle tolerance).
constexpr int GBS2_PT = 18, GBS2_STAGE = 2 * GRAM_SUB * GBS2_PT, GBS2_TAIL = GBS_TAIL + GBS2_STAGE;
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double fold8_dpp(double v) {                 // sum over the 8 lanes of an aligned group, in every lane of the group
    v += dpp_f64<0xB1>(v);                                               // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E>(v);                                               // quad_perm [2, 3, 0, 1]
    v += dpp_f64<0x141>(v);                                              // row_half_mirror: lane i <- lane 7 - i of the same 8 (the other quad's sum)
    return v;
}
template <int DC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))      // 128 VGPRs; 3 waves per SIMD: 213 us, 5 (96 VGPRs, spills): 474 against 195
k_gram_backsub2(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts, const double* __restrict__ focal,
                const double2* __restrict__ obs_xy, int ntasks, const int* __restrict__ gr_rec, const double* __restrict__ scale_cam,
                const double* __restrict__ scale_f, const double* __restrict__ PS,
                const double* __restrict__ y, int Nc, int loss, double la, const double* __restrict__ cam_c, const double* __restrict__ rot_c,
                const double* __restrict__ focal_c, double* __restrict__ pts_c, double* __restrict__ scal,
                const double* __restrict__ res_b, const double* __restrict__ res_q, const double* __restrict__ res_Sfc,
                const double* __restrict__ res_Sff, double res_tol2, double* __restrict__ res_r, double* __restrict__ res_pcg, long long* __restrict__ lacc = nullptr,
                const int split = 1) {        // split: waves per task, each with a contiguous share of the task's sub-chunks (small problems: more waves per SIMD)
    const DetScal ds{scal, lacc};
    constexpr int off = (DC == 6) ? 0 : 3;
    extern __shared__ __attribute__((aligned(16))) double sB[];          // per wave: k_gram_backsub's records | point records [2][8][GBS2_PT]: X at 0, PS at 4
    if (res_r && blockIdx.x == gridDim.x - 1) { __shared__ double red[4 * 4]; residual_check_body(Nc * DC, y, res_b, res_q, res_Sfc, res_Sff, res_tol2, res_r, res_pcg, red); return; }
    const int nwg = res_r ? (int)gridDim.x - 1 : (int)gridDim.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gtask = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, nwg) * (blockDim.x >> 6) + wave);
    const int task = gtask / split, part = gtask - task * split;
    if (task >= ntasks) return;
    double* sCam = sB + (size_t)wave * GBS2_TAIL;
    double* sCamC = sCam + GRAM_KMAX * GRAM_CAMREC;
    double* sStep = sCamC + GRAM_KMAX * GBS_CAMC;
    double* sPt = sStep + GRAM_KMAX * 6;
    const int* rec = gr_rec + (size_t)task * GRAM_REC;
    const int p0 = __builtin_amdgcn_readfirstlane(rec[0]), cnt = __builtin_amdgcn_readfirstlane(rec[1]), K = __builtin_amdgcn_readfirstlane(rec[2]);
    const int j00 = __builtin_amdgcn_readfirstlane(rec[3]);
    const int sub_per = ((cnt + GRAM_SUB - 1) / GRAM_SUB + split - 1) / split;
    const int s_begin = part * sub_per * GRAM_SUB, s_end = min(cnt, s_begin + sub_per * GRAM_SUB);
    if (s_begin >= cnt) return;
    const int camv = rec[4 + (lane & (GRAM_KMAX - 1))];                 // the task's cameras as a lane vector (gram_cam_of)
    gram_gather<33, GRAM_CAMREC, 6>(cam, rot, camv, K, lane, sCam);
    gram_gather<GBS_CAMC, GBS_CAMC, 3>(cam_c, rot_c, camv, K, lane, sCamC);
    { const int k = min(lane / DC, K - 1), a = lane - DC * (lane / DC), c = gram_cam_of(camv, k); const double st = scale_cam[6 * c + off + a] * y[c * DC + a]; if (lane < DC * K) sStep[6 * k + a] = st; }
    const double f = focal[0], sf = scale_f[0], yf = y[Nc * DC], fcand = focal_c[0];
    const int lq = lane & (GRAM_SUB - 1), lp = lane >> 3;               // observation (= camera of the group), point of the sub-chunk
    const int kq = min(lq, K - 1);
    double acc[4] = {0, 0, 0, 0};   // model, step2, xn2, candidate cost
    // the wave's three record loads of a sub-chunk: A lanes [0, 24) X; B entries [0, 64) of the 8 x 12 doubles of PS, C entries [64, 96)
    const int a_i = lane / 3, a_k = lane - 3 * a_i, b_i = lane / 12, b_k = lane - 12 * b_i, c_i = (64 + lane) / 12, c_k = 64 + lane - 12 * c_i;
    const int a_dst = a_i * GBS2_PT + a_k, b_dst = b_i * GBS2_PT + 4 + b_k, c_dst = c_i * GBS2_PT + 4 + c_k;
    double ra = 0.0, rb = 0.0, rc = 0.0; double2 ob;
#define GBS2_LOAD(s0_)                                                                                                            \
    do {                                                                                                                          \
        if (lane < 24) ra = pts[3 * (size_t)(p0 + min((s0_) + a_i, cnt - 1)) + a_k];                                              \
        rb = PS[12 * (size_t)(p0 + min((s0_) + b_i, cnt - 1)) + b_k];                                                             \
        if (lane < 32) rc = PS[12 * (size_t)(p0 + min((s0_) + c_i, cnt - 1)) + c_k];                                              \
    } while (0)
#define GBS2_STAGE_STORE(buf_)                                                                                                    \
    do {                                                                                                                          \
        double* d_ = sPt + (buf_) * GRAM_SUB * GBS2_PT;                                                                           \
        if (lane < 24) d_[a_dst] = ra;                                                                                            \
        d_[b_dst] = rb;                                                                                                           \
        if (lane < 32) d_[c_dst] = rc;                                                                                            \
    } while (0)
    GBS2_LOAD(s_begin);
    ob = obs_xy[j00 + (size_t)min(s_begin + lp, cnt - 1) * K + kq];
    GBS2_STAGE_STORE(0);
    if (s_begin + GRAM_SUB < s_end) GBS2_LOAD(s_begin + GRAM_SUB);
    wave_lds_handover();
    int buf = 0;
    for (int s0 = s_begin; s0 < s_end; s0 += GRAM_SUB, buf ^= 1) {
        const bool valid = s0 + lp < cnt, act = valid && lq < K;
        const double2 on = ob;
        const size_t pn = (size_t)(p0 + min(s0 + lp, cnt - 1));
        // the next sub-chunk's records (loaded one iteration ago) go to the other buffer -- its readers finished before the previous hand-over --, the one after that is requested
        if (s0 + GRAM_SUB < s_end) {
            GBS2_STAGE_STORE(buf ^ 1);
            ob = obs_xy[j00 + (size_t)min(s0 + GRAM_SUB + lp, cnt - 1) * K + kq];
            if (s0 + 2 * GRAM_SUB < s_end) GBS2_LOAD(s0 + 2 * GRAM_SUB);
        }
        const double* pr = sPt + (buf * GRAM_SUB + lp) * GBS2_PT;
        const double Xn[3] = {pr[0], pr[1], pr[2]};
        double m0 = 0.0, m1 = 0.0, r0 = 0.0, r1 = 0.0, J0[3] = {0, 0, 0}, J1[3] = {0, 0, 0}, w[3] = {0, 0, 0};
        if (act) {
            const double* crec = sCam + kq * GRAM_CAMREC;
            ObsLin L; lin_obs<DC == 6>(f, crec, crec + 6, Xn, on.x, on.y, loss, la, L);
            double Jc[2][DC]; cam_block_raw<DC>(L, Jc);
            m0 = L.Jf[0] * sf * yf; m1 = L.Jf[1] * sf * yf; r0 = L.r[0]; r1 = L.r[1];
#pragma unroll
            for (int a = 0; a < DC; a++) {
                const double ya = sStep[6 * kq + a];
                if (!jc_zero<DC>(0, a)) m0 += Jc[0][a] * ya;
                if (!jc_zero<DC>(1, a)) m1 += Jc[1][a] * ya;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { J0[k] = L.Jp[0][k]; J1[k] = L.Jp[1][k]; w[k] = J0[k] * m0 + J1[k] * m1; }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] = fold8_dpp(w[i]);
        const double P[9] = {pr[4], pr[5], pr[6], pr[7], pr[8], pr[9], pr[10], pr[11], pr[12]};
        const double u[3] = {P[6] - (P[0] * w[0] + P[1] * w[1] + P[2] * w[2]), P[7] - (P[1] * w[0] + P[3] * w[1] + P[4] * w[2]), P[8] - (P[2] * w[0] + P[4] * w[1] + P[5] * w[2])};
        const double Xc[3] = {Xn[0] - u[0], Xn[1] - u[1], Xn[2] - u[2]};
        if (act) {
            const double M0 = m0 + J0[0] * u[0] + J0[1] * u[1] + J0[2] * u[2], M1 = m1 + J1[0] * u[0] + J1[1] * u[1] + J1[2] * u[2];
            acc[0] += -(M0 * r0 + M1 * r1) + 0.5 * (M0 * M0 + M1 * M1);
            const double* cc = sCamC + kq * GBS_CAMC;
            acc[3] += obs_cost(fcand, cc, cc + 3, Xc, on.x, on.y, loss, la);
        }
        if (valid && lq == 0) {                                          // the point's candidate and step norms, once per point (a fixed point: zero record, zero step, not counted)
#pragma unroll
            for (int k = 0; k < 3; k++) pts_c[3 * pn + k] = Xc[k];
            if (P[0] > 0.0) { acc[1] += u[0] * u[0] + u[1] * u[1] + u[2] * u[2]; acc[2] += Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]; }
        }
        wave_lds_handover();                                             // this sub-chunk's records are read, the next one's are written
    }
#undef GBS2_LOAD
#undef GBS2_STAGE_STORE
    const double t = wave_transpose_sum(acc);
    const int slot = wave_tr_index();
    double* sl = scal + (size_t)((blockIdx.x * (blockDim.x >> 6) + wave) & (SC_NSLOT - 1)) * SC_TOTAL;
    if (slot < 4) sadd(ds, &sl[SC_MODEL + slot], t);
}

// ---- K4: robustified cost at a state (one lane per point) ----------------------------------------------
static __global__ void __launch_bounds__(256)
k_point_cost(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
             const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
             const int* __restrict__ pt_start, int nP, int loss, double la, double* __restrict__ out, int slot_stride) {
    __shared__ double red[4];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[1] = {0.0};
    if (p < nP) {
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double f = focal[0];
        for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
            const int c = obs_cam[j]; const double2 o = obs_xy[j];
            acc[0] += obs_cost(f, cam + 6 * c, rot + 27 * c, X, o.x, o.y, loss, la);
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(out + (size_t)(blockIdx.x & (SC_NSLOT - 1)) * slot_stride, acc[0]);   // slot_stride 0: a single scalar
}

// fold the SC_NSLOT replicas of the scalar block into replica 0 (sums; SC_GMAX by max) and clear the others: run before a
// collective reduces the block, by ONE workgroup of SC_TOTAL * 64 lanes
static __global__ void __launch_bounds__(SC_TOTAL * 64)
k_scal_fold(double* __restrict__ scal, double* __restrict__ packed, int rank) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* p = scal + (size_t)(lane & (SC_NSLOT - 1)) * SC_TOTAL + k;
    const double v = *p;
    const double r = (k == SC_GMAX) ? wave_max(v) : wave_sum(v);           // non-negative doubles order like their bit patterns
    __syncthreads();
    *p = (lane == 0) ? r : 0.0;
    // packed (optional): what one sum all-reduce carries -- the SC_NSUM sums, then one gradient-max slot per rank (the others stay
    // zero, so the sum over ranks leaves every rank's maximum side by side and no second, max-type collective is needed)
    if (packed && lane == 0) { if (k < SC_NSUM) packed[k] = r; else if (k == SC_GMAX) packed[SC_NSUM + rank] = r; }
}
// End of an LM iteration: the folded scalars and the solver flags go straight into pinned host memory, then a sequence number the host
// spins on -- no copy engine, no completion interrupt between the last kernel of an iteration and the host's decision.
// ONE workgroup of SC_TOTAL * 64 lanes; wave 0 does all the host writes so that its system-scope fence orders them before the flag.
// LmGate: what the host knows when it queues the end of iteration k -- enough for the device to decide "step accepted, go on" by the
// rules of the host loop (ba_solver.hip) and to compute the next trust-region radius.  The decision only gates a speculative launch of
// the next iteration's k_point_lin (queued before the host has seen the scalars); the host decides for itself and stays authoritative.
static __global__ void __launch_bounds__(SC_TOTAL * 64)
k_publish(const double* __restrict__ scal, const double* __restrict__ pcg, double* __restrict__ host_out, unsigned long long seq,
          const LmGate gate, double* __restrict__ spec, long long* __restrict__ lacc = nullptr, unsigned kmask = 0) {
    __shared__ double folded[SC_TOTAL];
    publish_body(scal, pcg, host_out, seq, gate, spec, folded, lacc, kmask);
}
// after the all-reduce: sums back into replica 0, gradient max = max over the per-rank slots
static __global__ void __launch_bounds__(64)
k_scal_unpack(double* __restrict__ scal, const double* __restrict__ packed, int nranks) {
    const int lane = threadIdx.x;
    if (lane < SC_NSUM) scal[lane] = packed[lane];
    double g = 0.0; for (int j = lane; j < nranks; j += 64) g = fmax(g, packed[SC_NSUM + j]);
    g = wave_max(g);
    if (lane == 0) scal[SC_GMAX] = g;
}

// |x|^2 over free parameters (iteration 0)
static __global__ void k_sqnorm_masked(const double* __restrict__ v, const double* __restrict__ mask, int n, double* __restrict__ out) {
    __shared__ double red[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[1] = {(i < n && mask[i] > 0.0) ? v[i] * v[i] : 0.0};
    block_sum<1>(acc, red);
    if (threadIdx.x == 0 && acc[0] != 0.0) unsafeAtomicAdd(out, acc[0]);
}

// the last launch of iteration 0, one instead of up to six: |x|^2 of the free point and camera(+focal) entries, the focal scale and -- when the camera
// column norms were summed over ranks first -- the camera scales.  Workgroups [0, gpt) take the points, the rest the cameras.
static __global__ void k_startup_tail(const double* __restrict__ pts, const double* __restrict__ mask_pt, int n_pt, int gpt,
                                      const double* __restrict__ cam, const double* __restrict__ mask_cam, int n_cam,
                                      const double* __restrict__ focal, const double* __restrict__ mask_f,
                                      const double* __restrict__ diag_cam, double* __restrict__ scale_cam,     // scale_cam == nullptr: already written
                                      const double* __restrict__ diag_f, double* __restrict__ scale_f, int jacobi, // scale_f == nullptr: scales kept from an earlier solve
                                      double* __restrict__ out_pt, double* __restrict__ out_cam,
                                      const double* __restrict__ df_part = nullptr, int n_df_part = 0,    // deterministic mode: k_colnorm's per-workgroup focal norms, added here in order,
                                      const double* __restrict__ scal_base = nullptr, long long* __restrict__ lacc = nullptr) {   // and the two norms through long accumulators
    const DetScal ds{scal_base, lacc};
    __shared__ double red[4];
    const bool is_pt = (int)blockIdx.x < gpt;
    const int i0 = (is_pt ? blockIdx.x : blockIdx.x - gpt) * blockDim.x + threadIdx.x, stride = (is_pt ? gpt : (int)gridDim.x - gpt) * blockDim.x;
    double acc[1] = {0.0};
    if (is_pt) { for (int i = i0; i < n_pt; i += stride) if (mask_pt[i] > 0.0) acc[0] += pts[i] * pts[i]; }
    else {
        for (int i = i0; i < n_cam; i += stride) {
            if (mask_cam[i] > 0.0) acc[0] += cam[i] * cam[i];
            if (scale_cam) scale_cam[i] = mask_cam[i] * (jacobi ? 1.0 / (1.0 + sqrt(diag_cam[i])) : 1.0);
        }
        if ((int)blockIdx.x == gpt && threadIdx.x == 0) {
            if (mask_f[0] > 0.0) acc[0] += focal[0] * focal[0];
            double dfv = diag_f[0];
            if (df_part) { dfv = 0.0; for (int i = 0; i < n_df_part; i++) dfv += df_part[i]; }
            if (scale_f) scale_f[0] = mask_f[0] * (jacobi ? 1.0 / (1.0 + sqrt(dfv)) : 1.0);
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0 && acc[0] != 0.0) sadd(ds, is_pt ? out_pt : out_cam, acc[0]);
}

// the state [cameras | points | focal] from one set of buffers into another in ONE launch (three copy launches before): reset and the end of a solve
static __global__ void k_copy_state(double* __restrict__ dc, const double* __restrict__ sc, int nc, double* __restrict__ dp, const double* __restrict__ sp, int np,
                                    double* __restrict__ df, const double* __restrict__ sf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) dp[i] = sp[i];
    else if (i - np < nc) dc[i - np] = sc[i - np];
    if (i == 0 && df) df[0] = sf[0];
}

// ---- parity probe: per-observation residual + 2x10 Jacobian (focal | t | r | X), robustified, unscaled
static __global__ void k_eval_dump(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
                            const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
                            const int* __restrict__ obs_pt, int M, int loss, double la, double* __restrict__ res, double* __restrict__ jac) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int c = obs_cam[j], p = obs_pt[j];
    const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
    const double2 o = obs_xy[j];
    ObsLin L; lin_obs<true>(focal[0], cam + 6 * c, rot + 27 * c, X, o.x, o.y, loss, la, L);
    for (int a = 0; a < 2; a++) {
        res[2 * j + a] = L.r[a];
        double* J = jac + 20 * (size_t)j + 10 * a;
        J[0] = L.Jf[a];
        for (int k = 0; k < 3; k++) { J[1 + k] = L.Jt[a][k]; J[4 + k] = L.Jr[a][k]; J[7 + k] = L.Jp[a][k]; }
    }
}

}  // namespace ssfm
