// spherical_sfm_amd -- SphericalEstimator::LeastSquares on the device, written from the reference's source lines:
//
//   src/spherical_estimator.cpp:110-157   LeastSquares: r0 = 0, t0 = (0,0,-1|+1), r1 = decompose(E), t1 = t0; residual blocks over
//                                         (r0, t0, r1, t1, u_i, v_i); constant: u_i, v_i (:140-141), r0 (:143), t0 (:144).
//                                         t1 IS NOT SET CONSTANT: Ceres minimises over the SIX parameters x = [r1; t1] and the caller
//                                         discards t1 (:156 rebuilds E from so3exp(r1) only).
//   src/spherical_estimator.cpp:23-65     SampsonError: R = Rj Ri^T, t = Rj (-Ri^T ti) + tj, E = [t]x R,
//                                         residual = (v.Eu)^2 / (|(Eu)_12|^2 + |(E^T v)_12|^2)   -- ONE residual per ray, already squared
//   :146-150                              TRUST_REGION / DENSE_NORMAL_CHOLESKY, 200 iterations, 10 consecutive invalid steps,
//                                         every other option a Ceres 2.2 default (Jacobi scaling, radius 1e4, tolerances 1e-6/1e-10/1e-8).
//
// (Rounds 1-2 fitted r1 alone -- SURVEY a12 says "only r1 free"; the source does not.  The two minima differ by 3e-5..9e-5 rad at
// 1/1000 ray noise, tests/test_ransac_lsq6_gpu.py.)
//
// With Ri = I and ti = t0 = (0, 0, tz):  t = t1 - tz R e_z,  E = [t]x R.  For a ray pair (u, v) put p = R u, w = v x t:
//     E u = t x p,        E^T v = R^T w,        d = v.(t x p) = p.w
// so the residual and its six partial derivatives need R, dR/dr_k (AngleAxisToRotationMatrix differentiated in closed form, the numbers
// the reference's Jets carry) and t -- 39 wave-uniform doubles -- and per ray:
//     d/dt1_k :  d(Eu) = e_k x p,              d(E^T v) = R^T (v x e_k),                 dd = (p x v)_k
//     d/dr1_k :  d(Eu) = tau_k x p + t x q_k,  d(E^T v) = dR_k^T w + R^T (v x tau_k),    dd = v . d(Eu)
//                with q_k = dR_k u, tau_k = dt/dr_k = -tz dR_k e_z.
// The Levenberg-Marquardt loop restates Ceres' TrustRegionMinimizer / LevenbergMarquardtStrategy rules (listed in oracle/lm.hpp; this is
// separately written code): 21 + 6 + 1 sums per linearisation, 6x6 Cholesky of Js^T Js + D^2 in registers.
#pragma once
#include "ransac_device.h"

namespace ssfm {

// AngleAxisToRotationMatrix (Ceres rotation.h; theta^2 > DBL_EPSILON: Rodrigues, else I + [r]x) and its three partial derivatives,
// row-major.  R = c I + (1 - c) a a^T + s [a]x with a = r / theta:  da/dr_k = (e_k - a a_k) / theta, dc/dr_k = -s a_k, ds/dr_k = c a_k.
__device__ __forceinline__ void angle_axis_matrix_and_derivatives(const double* r, double* R, double (*dR)[9]) {
    const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    if (th2 > DBL_EPSILON) {
        const double th = sqrt(th2), ith = 1.0 / th;
        const double a[3] = {r[0] * ith, r[1] * ith, r[2] * ith};
        const double c = cos(th), s = sin(th), m = 1.0 - c;
        R[0] = c + a[0] * a[0] * m;        R[1] = a[0] * a[1] * m - a[2] * s; R[2] = a[0] * a[2] * m + a[1] * s;
        R[3] = a[0] * a[1] * m + a[2] * s; R[4] = c + a[1] * a[1] * m;        R[5] = a[1] * a[2] * m - a[0] * s;
        R[6] = a[0] * a[2] * m - a[1] * s; R[7] = a[1] * a[2] * m + a[0] * s; R[8] = c + a[2] * a[2] * m;
        if (dR) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double da[3];
#pragma unroll
                for (int i = 0; i < 3; i++) da[i] = (((i == k) ? 1.0 : 0.0) - a[i] * a[k]) * ith;
                const double dc = -s * a[k], ds = c * a[k], dm = -dc;
                double* Q = dR[k];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) Q[3 * i + j] = ((i == j) ? dc : 0.0) + dm * a[i] * a[j] + m * (da[i] * a[j] + a[i] * da[j]);
                const double x0 = ds * a[0] + s * da[0], x1 = ds * a[1] + s * da[1], x2 = ds * a[2] + s * da[2];     // d(s a)
                Q[1] -= x2; Q[2] += x1; Q[3] += x2; Q[5] -= x0; Q[6] -= x1; Q[7] += x0;
            }
        }
    } else {
        R[0] = 1; R[1] = -r[2]; R[2] = r[1]; R[3] = r[2]; R[4] = 1; R[5] = -r[0]; R[6] = -r[1]; R[7] = r[0]; R[8] = 1;
        if (dR) {
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int i = 0; i < 9; i++) dR[k][i] = 0.0;
            dR[0][5] = -1; dR[0][7] = 1; dR[1][2] = 1; dR[1][6] = -1; dR[2][1] = -1; dR[2][3] = 1;
        }
    }
}

// the cooperating threads of one fit: a wave (no barrier, no LDS) or the whole workgroup
struct LsqWave {
    __device__ __forceinline__ int first() const { return threadIdx.x & 63; }
    __device__ __forceinline__ int stride() const { return 64; }
    template <int N> __device__ __forceinline__ void allsum(double (&v)[N]) const {      // every lane leaves with all N sums
        if (N == 1) { v[0] = wave_sum(v[0]); return; }
        const double t = wave_transpose_sum(v);                                          // sum i sits in the lane whose six bits reversed are i
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int holder = ((i & 1) << 5) | ((i & 2) << 3) | ((i & 4) << 1) | ((i & 8) >> 1) | ((i & 16) >> 3) | ((i & 32) >> 5);
            v[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), holder), __builtin_amdgcn_readlane(__double2loint(t), holder));
        }
    }
};
struct LsqBlock {
    double* red;        // LDS double[28 * blockDim/64]
    double* bc;         // LDS double[28]
    __device__ __forceinline__ int first() const { return threadIdx.x; }
    __device__ __forceinline__ int stride() const { return blockDim.x; }
    template <int N> __device__ __forceinline__ void allsum(double (&v)[N]) const {
        block_sum<N>(v, red);
        if (threadIdx.x == 0) for (int i = 0; i < N; i++) bc[i] = v[i];
        __syncthreads();
        for (int i = 0; i < N; i++) v[i] = bc[i];
        __syncthreads();
    }
};

constexpr int LSQ_NP = 6;                 // [r1; t1]
constexpr int LSQ_NA = 21;                // upper triangle of J^T J, row by row
// trace (optional, [10]): x[6], iterations, termination-ish (0 converged / 1 iteration limit / 2 invalid steps / 3 evaluation failure), initial cost, final cost
template <class G>
__device__ void sampson_lsq6(const G& grp, const int* list, int cnt, const double* pu, const double* pv, bool inward, double* E, double* trace) {
    const double tz = inward ? 1.0 : -1.0;
    double x[LSQ_NP];
    decompose_E_dev(E, inward, x);                                  // :115-117  r1 = r of the incoming model
    x[3] = 0.0; x[4] = 0.0; x[5] = tz;                              // :118-119  t1 = (0, 0, -1) or (0, 0, 1)
    double R[9], dR[3][9], t[3];
    double scale[LSQ_NP], A[LSQ_NA], g[LSQ_NP], x_cost = 0.0, cost0 = 0.0;
#pragma unroll
    for (int k = 0; k < LSQ_NP; k++) scale[k] = 1.0;
    bool finite_ok = true;

    // residuals + Jacobian at x (R, dR, t hold its tables) -> Jacobi-scaled Js^T Js, Js^T r, cost; identical in every thread afterwards
    auto linearize = [&]() {
        double acc[LSQ_NA + LSQ_NP + 1];
#pragma unroll
        for (int k = 0; k < LSQ_NA + LSQ_NP + 1; k++) acc[k] = 0.0;
        for (int q = grp.first(); q < cnt; q += grp.stride()) {
            const int i = list[q];
            const double u0 = pu[3 * i], u1 = pu[3 * i + 1], u2 = pu[3 * i + 2], v0 = pv[3 * i], v1 = pv[3 * i + 1], v2 = pv[3 * i + 2];
            const double p0 = R[0] * u0 + R[1] * u1 + R[2] * u2, p1 = R[3] * u0 + R[4] * u1 + R[5] * u2, p2 = R[6] * u0 + R[7] * u1 + R[8] * u2;
            const double w0 = v1 * t[2] - v2 * t[1], w1 = v2 * t[0] - v0 * t[2], w2 = v0 * t[1] - v1 * t[0];          // v x t
            const double e0 = t[1] * p2 - t[2] * p1, e1 = t[2] * p0 - t[0] * p2;                                      // (E u)_{0,1} = (t x p)_{0,1}
            const double f0 = R[0] * w0 + R[3] * w1 + R[6] * w2, f1 = R[1] * w0 + R[4] * w1 + R[7] * w2;              // (E^T v)_{0,1} = (R^T w)_{0,1}
            const double d = p0 * w0 + p1 * w1 + p2 * w2;
            const double den = e0 * e0 + e1 * e1 + f0 * f0 + f1 * f1, iden = 1.0 / den;
            const double res = (d * d) * iden;
            double j[LSQ_NP];
            // residual = d^2 / den:  d res = (2 d dd - res dden) / den,  dden = 2 (e.de + f.df)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double* Q = dR[k];
                const double q0 = Q[0] * u0 + Q[1] * u1 + Q[2] * u2, q1 = Q[3] * u0 + Q[4] * u1 + Q[5] * u2, q2 = Q[6] * u0 + Q[7] * u1 + Q[8] * u2;
                const double s0 = -tz * Q[2], s1 = -tz * Q[5], s2 = -tz * Q[8];                                      // tau_k
                const double de0 = (s1 * p2 - s2 * p1) + (t[1] * q2 - t[2] * q1), de1 = (s2 * p0 - s0 * p2) + (t[2] * q0 - t[0] * q2);
                const double de2 = (s0 * p1 - s1 * p0) + (t[0] * q1 - t[1] * q0);
                const double y0 = v1 * s2 - v2 * s1, y1 = v2 * s0 - v0 * s2, y2 = v0 * s1 - v1 * s0;                  // v x tau_k
                const double df0 = (Q[0] * w0 + Q[3] * w1 + Q[6] * w2) + (R[0] * y0 + R[3] * y1 + R[6] * y2);
                const double df1 = (Q[1] * w0 + Q[4] * w1 + Q[7] * w2) + (R[1] * y0 + R[4] * y1 + R[7] * y2);
                const double dd = v0 * de0 + v1 * de1 + v2 * de2;
                j[k] = ((2.0 * d * dd - res * (2.0 * (e0 * de0 + e1 * de1 + f0 * df0 + f1 * df1))) * iden) * scale[k];
            }
            {   // t1_0: e_0 x p = (0, -p2, p1);  v x e_0 = (0, v2, -v1);  dd = (p x v)_0
                const double de1 = -p2, df0 = R[3] * v2 - R[6] * v1, df1 = R[4] * v2 - R[7] * v1, dd = p1 * v2 - p2 * v1;
                j[3] = ((2.0 * d * dd - res * (2.0 * (e1 * de1 + f0 * df0 + f1 * df1))) * iden) * scale[3];
            }
            {   // t1_1: e_1 x p = (p2, 0, -p0);  v x e_1 = (-v2, 0, v0)
                const double de0 = p2, df0 = R[6] * v0 - R[0] * v2, df1 = R[7] * v0 - R[1] * v2, dd = p2 * v0 - p0 * v2;
                j[4] = ((2.0 * d * dd - res * (2.0 * (e0 * de0 + f0 * df0 + f1 * df1))) * iden) * scale[4];
            }
            {   // t1_2: e_2 x p = (-p1, p0, 0);  v x e_2 = (v1, -v0, 0)
                const double de0 = -p1, de1 = p0, df0 = R[0] * v1 - R[3] * v0, df1 = R[1] * v1 - R[4] * v0, dd = p0 * v1 - p1 * v0;
                j[5] = ((2.0 * d * dd - res * (2.0 * (e0 * de0 + e1 * de1 + f0 * df0 + f1 * df1))) * iden) * scale[5];
            }
            int o = 0;
#pragma unroll
            for (int a = 0; a < LSQ_NP; a++)
#pragma unroll
                for (int b = a; b < LSQ_NP; b++) acc[o++] += j[a] * j[b];
#pragma unroll
            for (int a = 0; a < LSQ_NP; a++) acc[LSQ_NA + a] += j[a] * res;
            acc[LSQ_NA + LSQ_NP] += 0.5 * res * res;
        }
        grp.allsum(acc);
#pragma unroll
        for (int k = 0; k < LSQ_NA; k++) A[k] = acc[k];
#pragma unroll
        for (int k = 0; k < LSQ_NP; k++) g[k] = acc[LSQ_NA + k];
        x_cost = acc[LSQ_NA + LSQ_NP];
        finite_ok = isfinite(x_cost);
    };
    auto tables = [&](const double* xx, bool with_derivatives) {
        angle_axis_matrix_and_derivatives(xx, R, with_derivatives ? dR : nullptr);
        t[0] = xx[3] - tz * R[2]; t[1] = xx[4] - tz * R[5]; t[2] = xx[5] - tz * R[8];
    };
    // packed upper-triangle index of (a, b), a <= b, in a 6 x 6 matrix
    auto ix = [](int a, int b) { return a * LSQ_NP - (a * (a - 1)) / 2 + (b - a); };

    tables(x, true);
    linearize();
    int iteration = 0, status = 3;
    if (finite_ok) {
        cost0 = x_cost;
        // Jacobi scaling from the iteration-0 Jacobian: s_k = 1 / (1 + |J_k|)
#pragma unroll
        for (int k = 0; k < LSQ_NP; k++) scale[k] = 1.0 / (1.0 + sqrt(A[ix(k, k)]));
#pragma unroll
        for (int a = 0; a < LSQ_NP; a++) {
#pragma unroll
            for (int b = a; b < LSQ_NP; b++) A[ix(a, b)] *= scale[a] * scale[b];
            g[a] *= scale[a];
        }
        double radius = 1e4, decrease = 2.0;
        double x_norm = 0.0;
#pragma unroll
        for (int k = 0; k < LSQ_NP; k++) x_norm += x[k] * x[k];
        x_norm = sqrt(x_norm);
        int invalid = 0; bool last_ok = true;
        status = 0;
        while (true) {
            if (iteration >= 200) { status = 1; break; }                               // :148 max_num_iterations
            double gmax = 0.0;
#pragma unroll
            for (int k = 0; k < LSQ_NP; k++) gmax = fmax(gmax, fabs(g[k] / scale[k]));   // gradient of the unscaled problem
            if (last_ok && gmax <= 1e-10) break;
            if (radius <= 1e-32) break;
            iteration++;
            // L L^T = Js^T Js + D^2, D^2 = clamp(diag, 1e-6, 1e32) / radius; lower triangle in Lm (full 6 x 6 for simple indexing)
            double Lm[LSQ_NP][LSQ_NP];
            bool chol_ok = true;
#pragma unroll
            for (int jc = 0; jc < LSQ_NP; jc++) {
                double dj = A[ix(jc, jc)] + fmin(fmax(A[ix(jc, jc)], 1e-6), 1e32) / radius;
#pragma unroll
                for (int k = 0; k < jc; k++) dj -= Lm[jc][k] * Lm[jc][k];
                chol_ok = chol_ok && (dj > 0.0);
                const double l = sqrt(dj);
                Lm[jc][jc] = l;
#pragma unroll
                for (int i = jc + 1; i < LSQ_NP; i++) {
                    double vv = A[ix(jc, i)];
#pragma unroll
                    for (int k = 0; k < jc; k++) vv -= Lm[i][k] * Lm[jc][k];
                    Lm[i][jc] = vv / l;
                }
            }
            double z[LSQ_NP], st[LSQ_NP];
#pragma unroll
            for (int i = 0; i < LSQ_NP; i++) {
                double vv = g[i];
#pragma unroll
                for (int k = 0; k < i; k++) vv -= Lm[i][k] * z[k];
                z[i] = vv / Lm[i][i];
            }
#pragma unroll
            for (int i = LSQ_NP - 1; i >= 0; i--) {
                double vv = z[i];
#pragma unroll
                for (int k = i + 1; k < LSQ_NP; k++) vv -= Lm[k][i] * st[k];
                st[i] = vv / Lm[i][i];
            }
#pragma unroll
            for (int i = 0; i < LSQ_NP; i++) st[i] = -st[i];                            // trust-region step of the scaled problem
            // model cost change -(Js st)^T (r + Js st / 2) = -(g.st + st^T A st / 2)
            double gs = 0.0, sAs = 0.0;
#pragma unroll
            for (int a = 0; a < LSQ_NP; a++) {
                gs += g[a] * st[a];
                sAs += A[ix(a, a)] * st[a] * st[a];
#pragma unroll
                for (int b = a + 1; b < LSQ_NP; b++) sAs += 2.0 * A[ix(a, b)] * st[a] * st[b];
            }
            const double model = -(gs + 0.5 * sAs);
            if (!chol_ok || !(model > 0.0) || !isfinite(model)) {
                if (++invalid >= 10) { status = 2; break; }                           // :149 max_num_consecutive_invalid_steps
                radius /= decrease; decrease *= 2.0; last_ok = false; continue;
            }
            invalid = 0;
            double xc[LSQ_NP], step_norm = 0.0;
#pragma unroll
            for (int k = 0; k < LSQ_NP; k++) { xc[k] = x[k] + st[k] * scale[k]; step_norm += (xc[k] - x[k]) * (xc[k] - x[k]); }
            step_norm = sqrt(step_norm);
            // candidate cost: R, t of the candidate overwrite the tables (dR still belongs to x; a rejected step restores R, t below)
            tables(xc, false);
            double c[1] = {0.0};
            for (int q = grp.first(); q < cnt; q += grp.stride()) {
                const int i = list[q];
                const double u0 = pu[3 * i], u1 = pu[3 * i + 1], u2 = pu[3 * i + 2], v0 = pv[3 * i], v1 = pv[3 * i + 1], v2 = pv[3 * i + 2];
                const double p0 = R[0] * u0 + R[1] * u1 + R[2] * u2, p1 = R[3] * u0 + R[4] * u1 + R[5] * u2, p2 = R[6] * u0 + R[7] * u1 + R[8] * u2;
                const double w0 = v1 * t[2] - v2 * t[1], w1 = v2 * t[0] - v0 * t[2], w2 = v0 * t[1] - v1 * t[0];
                const double e0 = t[1] * p2 - t[2] * p1, e1 = t[2] * p0 - t[0] * p2;
                const double f0 = R[0] * w0 + R[3] * w1 + R[6] * w2, f1 = R[1] * w0 + R[4] * w1 + R[7] * w2;
                const double d = p0 * w0 + p1 * w1 + p2 * w2;
                const double r = (d * d) / (e0 * e0 + e1 * e1 + f0 * f0 + f1 * f1);
                c[0] += 0.5 * r * r;
            }
            grp.allsum(c);
            double cand = c[0];
            if (!isfinite(cand)) cand = 1.79769313486231570815e308;
            const double change = x_cost - cand;
            const double rho = (cand >= 1.79769313486231570815e308) ? -1.79769313486231570815e308 : change / model;
            const bool stop = (step_norm <= 1e-8 * (x_norm + 1e-8)) || (fabs(change) <= 1e-6 * x_cost);    // parameter / function tolerance: candidate not taken
            if (!stop && rho > 1e-3) {
                double xp[LSQ_NP];
#pragma unroll
                for (int k = 0; k < LSQ_NP; k++) { xp[k] = x[k]; x[k] = xc[k]; }
                tables(x, true);
                linearize();                      // scale[] is applied inside: the sums come back Jacobi-scaled
                if (!finite_ok) {                 // evaluation failure: the last good x stands
#pragma unroll
                    for (int k = 0; k < LSQ_NP; k++) x[k] = xp[k];
                    status = 3; break;
                }
                x_norm = 0.0;
#pragma unroll
                for (int k = 0; k < LSQ_NP; k++) x_norm += x[k] * x[k];
                x_norm = sqrt(x_norm);
                { const double t3 = 2.0 * rho - 1.0; radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3)); } decrease = 2.0; last_ok = true;
            } else {
                if (stop) break;
                tables(x, false);                 // R, t back to the accepted point (dR never left it)
                radius /= decrease; decrease *= 2.0; last_ok = false;
            }
        }
    }
    if (trace) {
#pragma unroll
        for (int k = 0; k < LSQ_NP; k++) trace[k] = x[k];
        trace[6] = (double)iteration; trace[7] = (double)status; trace[8] = cost0; trace[9] = x_cost;
    }
    double Rm[9]; so3exp(x, Rm); make_E_dev(Rm, inward, E);             // :156  make_spherical_essential_matrix(so3exp(r1)); t1 is dropped
}

// one wave, no barrier, no LDS (all 64 lanes of the wave must call it; E identical in every lane, in/out)
__device__ __forceinline__ void wave_sampson_lsq(const int* list, int cnt, const double* pu, const double* pv, bool inward, double* E, double* trace = nullptr) {
    sampson_lsq6(LsqWave{}, list, cnt, pu, pv, inward, E, trace);
}
// the whole workgroup (uniform control flow).  red: LDS double[28 * blockDim/64]; bc: LDS double[28]
__device__ __forceinline__ void block_sampson_lsq(const int* list, int cnt, const double* pu, const double* pv, bool inward, double* E, double* red, double* bc,
                                                  double* trace = nullptr) {
    sampson_lsq6(LsqBlock{red, bc}, list, cnt, pu, pv, inward, E, trace);
}

}  // namespace ssfm
