// spherical_sfm_amd -- batched spherical relative-pose RANSAC.
//
// Replaces the per-pair body of estimate_pairwise (reference examples/spherical_sfm_tools.cpp:332-420), i.e.
//   LocallyOptimizedMSAC<Matrix3d, ..., SphericalEstimator>::EstimateModel   include/RansacLib/ransac.h:128-275
//   SphericalEstimator::MinimalSolver -> spherical_solver_action_matrix        src/spherical_solvers.cpp:102-311
//   SphericalEstimator::EvaluateModelOnPoint (Sampson)                         src/spherical_estimator.cpp:67-78
//   SphericalEstimator::LeastSquares (final_least_squares_ = true)             src/spherical_estimator.cpp:110-157
//   SphericalEstimator::Decompose / decompose_spherical_essential_matrix       src/spherical_utils.cpp:16-66
// with ONE launch for thousands of pairs: a workgroup per image pair keeps the pair's rays in LDS; every lane draws
// 3-point samples (counter-based RNG), solves the minimal problem (QR nullspace -> 6x10 cubic constraints -> 4x4 action
// matrix -> eigen-solutions) and MSAC-scores its up-to-4 models against all rays (LDS broadcast reads); a block arg-min
// picks the pair's best model.  A second kernel refines it on its inliers (3-dof LM on the Sampson residuals, Ceres
// rules) and decomposes E into R.  The reference's sequential, adaptively terminated sampling (std::mt19937) is replaced
// by a fixed hypothesis budget evaluated in parallel, so parity is on the final R / inlier set, not on the sample trace
// (SURVEY 7 "RANSAC determinism").  Complex eigen-pairs of the action matrix are skipped: the reference scores the real
// part of such eigenvectors, which is never a valid model.
#include <algorithm>
#include <cstdio>
#include "ba_handle.h"
#include "dual.h"

namespace ssfm {

__device__ __forceinline__ double sampson_err(const double* E, const double* u, const double* v) {
    const double e0 = E[0] * u[0] + E[1] * u[1] + E[2] * u[2], e1 = E[3] * u[0] + E[4] * u[1] + E[5] * u[2], e2 = E[6] * u[0] + E[7] * u[1] + E[8] * u[2];
    const double f0 = E[0] * v[0] + E[3] * v[1] + E[6] * v[2], f1 = E[1] * v[0] + E[4] * v[1] + E[7] * v[2];
    const double d = v[0] * e0 + v[1] * e1 + v[2] * e2;
    return (d * d) / (e0 * e0 + e1 * e1 + f0 * f0 + f1 * f1);
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

// polynomial helpers in (x,y,z): Lin[3], Quad[6] = xx xy xz yy yz zz, Cub[10] = x3 x2y xy2 y3 x2z xyz y2z xz2 yz2 z3
__device__ __forceinline__ void qmul_acc(double* q, const double* a, const double* b) {
    q[0] += a[0] * b[0]; q[1] += a[0] * b[1] + a[1] * b[0]; q[2] += a[0] * b[2] + a[2] * b[0];
    q[3] += a[1] * b[1]; q[4] += a[1] * b[2] + a[2] * b[1]; q[5] += a[2] * b[2];
}
__device__ __forceinline__ void cub_acc(double* r, const double* a, const double* b, double s) {
    r[0] += s * (a[0] * b[0]); r[1] += s * (a[0] * b[1] + a[1] * b[0]); r[2] += s * (a[1] * b[1] + a[3] * b[0]); r[3] += s * (a[3] * b[1]);
    r[4] += s * (a[0] * b[2] + a[2] * b[0]); r[5] += s * (a[1] * b[2] + a[2] * b[1] + a[4] * b[0]); r[6] += s * (a[3] * b[2] + a[4] * b[1]);
    r[7] += s * (a[2] * b[2] + a[5] * b[0]); r[8] += s * (a[4] * b[2] + a[5] * b[1]); r[9] += s * (a[5] * b[2]);
}

// complex helpers for Ferrari's method (principal branches, as std::sqrt / std::pow(z, 1/3) of the reference's SolveQuartic)
struct cplx { double r, i; };
__device__ __forceinline__ cplx cmk(double r, double i = 0.0) { return cplx{r, i}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cplx{a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cplx{a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return cplx{a.r * s, a.i * s}; }
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) { const double d = b.r * b.r + b.i * b.i; return cplx{(a.r * b.r + a.i * b.i) / d, (a.i * b.r - a.r * b.i) / d}; }
__device__ __forceinline__ cplx csqrt_p(cplx a) {
    const double m = hypot(a.r, a.i);
    if (m == 0.0) return cplx{0.0, 0.0};
    const double sr = sqrt(0.5 * (m + fabs(a.r)));
    if (a.r >= 0.0) return cplx{sr, a.i / (2.0 * sr)};
    return cplx{fabs(a.i) / (2.0 * sr), (a.i >= 0.0) ? sr : -sr};
}
__device__ __forceinline__ cplx ccbrt_p(cplx a) {                   // exp(log(a) / 3), arg in (-pi, pi]
    const double m = hypot(a.r, a.i);
    if (m == 0.0) return cplx{0.0, 0.0};
    const double rho = cbrt(m), th = atan2(a.i, a.r) / 3.0;
    return cplx{rho * cos(th), rho * sin(th)};
}

// Minimal solver for one 3-point sample.  Es: up to 4 real solutions (row-major 3x3, unit Frobenius norm); returns count.
// POLY = false: action-matrix variant (src/spherical_solvers.cpp:102-311); POLY = true: quartic variant (:313-660), whose
// constraint matrix is the same six rows (times 1/2) with the monomials ordered [x^3 x^2y xy^2 x^2z xyz xz^2 | y^3 y^2z yz^2 z^3].
template <bool POLY>
__device__ int spherical_minimal_solver(const double* u3, const double* v3, double* Es) {
    // A^T (6x3), Householder QR without pivoting; B = last three columns of Q  (src/spherical_solvers.cpp:119-125)
    double At[6][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double* u = u3 + 3 * i; const double* v = v3 + 3 * i;
        At[0][i] = u[0] * v[0] - u[1] * v[1]; At[1][i] = u[0] * v[1] + u[1] * v[0]; At[2][i] = u[2] * v[0];
        At[3][i] = u[2] * v[1]; At[4][i] = u[0] * v[2]; At[5][i] = u[1] * v[2];
    }
    double hv[3][6], tau[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double alpha = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) if (i >= k) alpha += At[i][k] * At[i][k];
        alpha = sqrt(alpha);
        const double x0 = At[k][k], beta = (x0 >= 0) ? -alpha : alpha;
        double vn = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) { hv[k][i] = (i < k) ? 0.0 : ((i == k) ? x0 - beta : At[i][k]); vn += hv[k][i] * hv[k][i]; }
        tau[k] = (vn > 0) ? 2.0 / vn : 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) if (j >= k) {
            double d = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) d += hv[k][i] * At[i][j];
            d *= tau[k];
#pragma unroll
            for (int i = 0; i < 6; i++) At[i][j] -= d * hv[k][i];
        }
    }
    double B[6][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double q[6] = {0, 0, 0, 0, 0, 0}; q[3 + c] = 1.0;
#pragma unroll
        for (int k = 2; k >= 0; k--) {
            double d = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) d += hv[k][i] * q[i];
            d *= tau[k];
#pragma unroll
            for (int i = 0; i < 6; i++) q[i] -= d * hv[k][i];
        }
#pragma unroll
        for (int i = 0; i < 6; i++) B[i][c] = q[i];
    }
    // E(x,y,z) = [[p0,p1,p2],[p1,-p0,p3],[p4,p5,0]], p_k = B[k] . (x,y,z);  T = 2 E E^T E - tr(E E^T) E
    double Em[3][3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        Em[0][0][c] = B[0][c]; Em[0][1][c] = B[1][c]; Em[0][2][c] = B[2][c];
        Em[1][0][c] = B[1][c]; Em[1][1][c] = -B[0][c]; Em[1][2][c] = B[3][c];
        Em[2][0][c] = B[4][c]; Em[2][1][c] = B[5][c]; Em[2][2][c] = 0.0;
    }
    double EEt[3][3][6];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
#pragma unroll
            for (int m = 0; m < 6; m++) EEt[i][j][m] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) qmul_acc(EEt[i][j], Em[i][k], Em[j][k]);
        }
    double tr[6];
#pragma unroll
    for (int m = 0; m < 6; m++) tr[m] = EEt[0][0][m] + EEt[1][1][m] + EEt[2][2][m];
    // rows: -T01, T20, T00, T21, T12, T22  (the reference's C matrix, src/spherical_solvers.cpp:262-277)
    double C[6][10];
    const int ri[6] = {0, 2, 0, 2, 1, 2}, rj[6] = {1, 0, 0, 1, 2, 2};
    const double rs[6] = {-1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
#pragma unroll
    for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int m = 0; m < 10; m++) C[r][m] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) cub_acc(C[r], EEt[ri[r]][k], Em[k][rj[r]], 2.0 * rs[r]);
        cub_acc(C[r], tr, Em[ri[r]][rj[r]], -rs[r]);
    }
    if (POLY) {                                         // monomial order of the quartic variant; the factor 1/2 is exact
        const int perm[10] = {0, 1, 2, 4, 5, 7, 3, 6, 8, 9};
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double t[10];
#pragma unroll
            for (int m = 0; m < 10; m++) t[m] = 0.5 * C[r][perm[m]];
#pragma unroll
            for (int m = 0; m < 10; m++) C[r][m] = t[m];
        }
    }
    // G = C[:, :6]^-1 C[:, 6:]  by Gaussian elimination with partial pivoting (static indices: predicated row swaps)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int p = k; double best = fabs(C[k][k]);
#pragma unroll
        for (int i = 0; i < 6; i++) if (i > k) { const double a = fabs(C[i][k]); if (a > best) { best = a; p = i; } }
        if (best == 0.0) return 0;
#pragma unroll
        for (int i = 0; i < 6; i++) if (i > k && i == p) {
#pragma unroll
            for (int m = 0; m < 10; m++) { const double t = C[k][m]; C[k][m] = C[i][m]; C[i][m] = t; }
        }
        const double inv = 1.0 / C[k][k];
#pragma unroll
        for (int i = 0; i < 6; i++) if (i != k) {
            const double f = C[i][k] * inv;
#pragma unroll
            for (int m = 0; m < 10; m++) if (m >= k) C[i][m] -= f * C[k][m];
        }
#pragma unroll
        for (int m = 0; m < 10; m++) if (m >= k) C[k][m] *= inv;
    }
    if (POLY) {
        // rows 4, 5: xy + G4.[y^3 y^2 y 1] = 0, x + G5.[y^3 y^2 y 1] = 0 (z = 1)  =>  quartic in y  (src/spherical_solvers.cpp:623-627)
        const double* G4 = &C[4][6]; const double* G5 = &C[5][6];
        const double qa = -G5[0], qb = G4[0] - G5[1], qc = G4[1] - G5[2], qd = G4[2] - G5[3], qe = G4[3];
        if (qa == 0.0 || !isfinite(qa + qb + qc + qd + qe)) return 0;
        // Ferrari (src/spherical_solvers.cpp:15-69)
        const double a2 = qa * qa, b2 = qb * qb, a3 = a2 * qa, b3 = b2 * qb, a4 = a3 * qa, b4 = b3 * qb;
        const double alpha = -3.0 * b2 / (8.0 * a2) + qc / qa;
        const double beta = b3 / (8.0 * a3) - qb * qc / (2.0 * a2) + qd / qa;
        const double gamma = -3.0 * b4 / (256.0 * a4) + b2 * qc / (16.0 * a3) - qb * qd / (4.0 * a2) + qe / qa;
        const double P = -alpha * alpha / 12.0 - gamma;
        const double Q = -alpha * alpha * alpha / 108.0 + alpha * gamma / 3.0 - beta * beta / 8.0;
        const cplx Rr = cadd(cmk(-Q / 2.0), csqrt_p(cmk(Q * Q / 4.0 + P * P * P / 27.0)));
        const cplx U = ccbrt_p(Rr);
        cplx y;
        if (fabs(U.r) < 1e-8) y = csub(cmk(-5.0 * alpha / 6.0), ccbrt_p(cmk(Q)));
        else y = cadd(csub(cmk(-5.0 * alpha / 6.0), cdiv(cmk(P), cscale(U, 3.0))), U);
        const cplx w = csqrt_p(cadd(cmk(alpha), cscale(y, 2.0)));
        const cplx base = cadd(cmk(3.0 * alpha), cscale(y, 2.0));
        const cplx bw = cdiv(cmk(2.0 * beta), w);
        const cplx s1 = csqrt_p(cscale(cadd(base, bw), -1.0)), s2 = csqrt_p(cscale(csub(base, bw), -1.0));
        const double sh = -qb / (4.0 * qa);
        const cplx roots[4] = {cadd(cmk(sh), cscale(cadd(w, s1), 0.5)), cadd(cmk(sh), cscale(csub(w, s1), 0.5)),
                               cadd(cmk(sh), cscale(cadd(cscale(w, -1.0), s2), 0.5)), cadd(cmk(sh), cscale(csub(cscale(w, -1.0), s2), 0.5))};
        const double scale = 1.0 + fabs(qb / qa) + sqrt(fabs(qc / qa)) + cbrt(fabs(qd / qa)) + sqrt(sqrt(fabs(qe / qa)));
        int count = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (!isfinite(roots[s].r) || !(fabs(roots[s].i) <= 1e-7 * scale)) continue;     // complex pair: not a model
            double yv = roots[s].r;
#pragma unroll
            for (int it = 0; it < 2; it++) {                   // Newton polish on the real axis
                const double pv = (((qa * yv + qb) * yv + qc) * yv + qd) * yv + qe, dp = ((4 * qa * yv + 3 * qb) * yv + 2 * qc) * yv + qd;
                if (dp != 0.0) yv -= pv / dp;
            }
            const double xv = -(((G5[0] * yv + G5[1]) * yv + G5[2]) * yv + G5[3]);
            double ps[6];
#pragma unroll
            for (int k = 0; k < 6; k++) ps[k] = B[k][0] * xv + B[k][1] * yv + B[k][2];
            double* E = Es + 9 * count;
            E[0] = ps[0]; E[1] = ps[1]; E[2] = ps[2]; E[3] = ps[1]; E[4] = -ps[0]; E[5] = ps[3]; E[6] = ps[4]; E[7] = ps[5]; E[8] = 0.0;
            double n2 = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) n2 += E[k] * E[k];
            if (!(n2 > 0.0) || !isfinite(n2)) continue;
            const double inv = 1.0 / sqrt(n2);
#pragma unroll
            for (int k = 0; k < 9; k++) E[k] *= inv;
            count++;
        }
        return count;
    }
    // action matrix of multiplication by x on [y^2, x, y, 1]
    double M[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) { M[0][k] = -C[2][6 + k]; M[1][k] = -C[4][6 + k]; M[2][k] = -C[5][6 + k]; M[3][k] = 0.0; }
    M[3][1] = 1.0;
    // characteristic polynomial (Faddeev-LeVerrier) and its roots (Durand-Kerner on the monic quartic)
    double cc[4];
    {
        double Bk[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) Bk[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 1; k <= 4; k++) {
            double AB[4][4]; double trc = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) { double s = 0;
#pragma unroll
                    for (int t = 0; t < 4; t++) s += M[i][t] * Bk[t][j];
                    AB[i][j] = s; if (i == j) trc += s; }
            cc[k - 1] = -trc / k;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) Bk[i][j] = AB[i][j] + ((i == j) ? cc[k - 1] : 0.0);
        }
    }
    const double c3 = cc[0], c2 = cc[1], c1 = cc[2], c0 = cc[3];
    const double scale = 1.0 + fabs(c3) + sqrt(fabs(c2)) + cbrt(fabs(c1)) + sqrt(sqrt(fabs(c0)));
    double zr[4] = {0.4 * scale, -0.9 * scale, -0.4 * scale, 0.9 * scale}, zi[4] = {0.9 * scale, 0.4 * scale, -0.9 * scale, -0.4 * scale};
    for (int it = 0; it < 100; it++) {
        double change = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // p(z) by Horner in complex arithmetic
            double pr = zr[i] + c3, pi = zi[i];
            double tr_ = pr * zr[i] - pi * zi[i] + c2, ti = pr * zi[i] + pi * zr[i]; pr = tr_; pi = ti;
            tr_ = pr * zr[i] - pi * zi[i] + c1; ti = pr * zi[i] + pi * zr[i]; pr = tr_; pi = ti;
            tr_ = pr * zr[i] - pi * zi[i] + c0; ti = pr * zi[i] + pi * zr[i]; pr = tr_; pi = ti;
            double dr = 1.0, di = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) if (j != i) { const double ar = zr[i] - zr[j], ai = zi[i] - zi[j]; const double nr = dr * ar - di * ai, ni = dr * ai + di * ar; dr = nr; di = ni; }
            double dn = dr * dr + di * di; if (dn == 0.0) dn = 1e-300;
            const double qr_ = (pr * dr + pi * di) / dn, qi = (pi * dr - pr * di) / dn;
            zr[i] -= qr_; zi[i] -= qi; change = fmax(change, fabs(qr_) + fabs(qi));
        }
        if (change < 1e-15 * scale) break;
    }
    int count = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (fabs(zi[s]) > 1e-9 * scale) continue;             // complex pair: not a model
        double l = zr[s];
        // two Newton steps on the real axis polish the root
#pragma unroll
        for (int it = 0; it < 2; it++) { const double p = (((l + c3) * l + c2) * l + c1) * l + c0, dp = ((4 * l + 3 * c3) * l + 2 * c2) * l + c1; if (dp != 0.0) l -= p / dp; }
        // eigenvector (v0, v1, v2, 1): v1 = l; rows 1,2 of (M - l I) v = 0 give v0, v2
        const double a11 = M[1][0], a12 = M[1][2], b1 = -((M[1][1] - l) * l + M[1][3]);
        const double a21 = M[2][0], a22 = M[2][2] - l, b2 = -(M[2][1] * l + M[2][3]);
        const double det = a11 * a22 - a12 * a21;
        if (det == 0.0) continue;
        const double v2 = (a11 * b2 - b1 * a21) / det;
        const double bx = l, by = v2;
        double ps[6];
#pragma unroll
        for (int k = 0; k < 6; k++) ps[k] = B[k][0] * bx + B[k][1] * by + B[k][2];
        double* E = Es + 9 * count;
        E[0] = ps[0]; E[1] = ps[1]; E[2] = ps[2]; E[3] = ps[1]; E[4] = -ps[0]; E[5] = ps[3]; E[6] = ps[4]; E[7] = ps[5]; E[8] = 0.0;
        double n2 = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) n2 += E[k] * E[k];
        if (!(n2 > 0.0) || !isfinite(n2)) continue;
        const double inv = 1.0 / sqrt(n2);
#pragma unroll
        for (int k = 0; k < 9; k++) E[k] *= inv;
        count++;
    }
    return count;
}

// probe for parity tests: one lane per given sample
template <bool POLY>
__global__ void k_solver_probe(int S, const int* __restrict__ sample, const double* __restrict__ u, const double* __restrict__ v,
                               double* __restrict__ Es, int* __restrict__ counts) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double u3[9], v3[9];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { u3[3 * i + k] = u[3 * sample[3 * s + i] + k]; v3[3 * i + k] = v[3 * sample[3 * s + i] + k]; }
    double E[36];
    const int c = spherical_minimal_solver<POLY>(u3, v3, E);
    counts[s] = c;
    for (int k = 0; k < 36; k++) Es[36 * (size_t)s + k] = (k < 9 * c) ? E[k] : 0.0;
}

// ---- kernel 1: hypotheses + MSAC scores + per-pair arg-min ----------------------------------------------
template <bool POLY>
__global__ void __launch_bounds__(256)
k_ransac_hypotheses(const int* __restrict__ pair_ptr, const double* __restrict__ u, const double* __restrict__ v, double sq_thresh,
                    int num_hyp, unsigned long long seed, const int* __restrict__ pair_id, double* __restrict__ bestE,
                    double* __restrict__ bestScore) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sScore[256]; __shared__ int sIdx[256];
    const int pair = blockIdx.x;
    const int stream_id = pair_id ? pair_id[pair] : pair;      // a sharded batch keeps the random stream of the pair's global index
    const int r0 = pair_ptr[pair], n = pair_ptr[pair + 1] - r0;
    double* su = lds; double* sv = lds + (size_t)3 * n;
    for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) { su[i] = u[(size_t)3 * r0 + i]; sv[i] = v[(size_t)3 * r0 + i]; }
    __syncthreads();
    double myBest = 1.79e308; double myE[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n >= 3) {
        for (int h = threadIdx.x; h < num_hyp; h += blockDim.x) {
            // three distinct indices from a counter-based generator (sampling without replacement, sampling.h:77-97)
            int idx[3]; unsigned long long ctr = splitmix64(seed ^ ((unsigned long long)stream_id << 32) ^ (unsigned long long)h);
            for (int i = 0; i < 3; i++) {
                bool dup = true;
                while (dup) { ctr = splitmix64(ctr); idx[i] = (int)((ctr >> 11) % (unsigned long long)n); dup = false; for (int j = 0; j < i; j++) if (idx[j] == idx[i]) dup = true; }
            }
            double u3[9], v3[9];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) { u3[3 * i + k] = su[3 * idx[i] + k]; v3[3 * i + k] = sv[3 * idx[i] + k]; }
            double Es[36];
            const int cnt = spherical_minimal_solver<POLY>(u3, v3, Es);
            for (int m = 0; m < cnt; m++) {
                const double* E = Es + 9 * m;
                double sc = 0.0;
                for (int i = 0; i < n; i++) sc += fmin(sampson_err(E, su + 3 * i, sv + 3 * i), sq_thresh);      // MSAC, ransac.h:295-310
                if (sc < myBest) { myBest = sc; for (int k = 0; k < 9; k++) myE[k] = E[k]; }
            }
        }
    }
    sScore[threadIdx.x] = myBest; sIdx[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { if (sScore[threadIdx.x + s] < sScore[threadIdx.x]) { sScore[threadIdx.x] = sScore[threadIdx.x + s]; sIdx[threadIdx.x] = sIdx[threadIdx.x + s]; } }
        __syncthreads();
    }
    if (threadIdx.x == sIdx[0]) { for (int k = 0; k < 9; k++) bestE[9 * (size_t)pair + k] = myE[k]; bestScore[pair] = myBest; }
}

// ---- kernel 2: final least squares on the inliers + decomposition --------------------------------------
template <typename T>
__device__ __forceinline__ void sampson_residual_r(const T* r1, double tz, const double* u, const double* v, T* res) {
    // src/spherical_estimator.cpp:23-65 with ri = 0, ti = tj = (0,0,tz): R = Rj, t = -Rj ti + tj
    T R[9]; aa_to_matrix_t(r1, R);                      // row-major
    const T t[3] = {R[2] * (-tz), R[5] * (-tz), R[8] * (-tz) + tz};
    T E[9];
#pragma unroll
    for (int j = 0; j < 3; j++) { E[j] = t[1] * R[6 + j] - t[2] * R[3 + j]; E[3 + j] = t[2] * R[j] - t[0] * R[6 + j]; E[6 + j] = t[0] * R[3 + j] - t[1] * R[j]; }
    const T e0 = E[0] * u[0] + E[1] * u[1] + E[2] * u[2], e1 = E[3] * u[0] + E[4] * u[1] + E[5] * u[2], e2 = E[6] * u[0] + E[7] * u[1] + E[8] * u[2];
    const T f0 = E[0] * v[0] + E[3] * v[1] + E[6] * v[2], f1 = E[1] * v[0] + E[4] * v[1] + E[7] * v[2];
    const T d = e0 * v[0] + e1 * v[1] + e2 * v[2];
    *res = (d * d) / (e0 * e0 + e1 * e1 + f0 * f0 + f1 * f1);
}
__device__ void make_E_dev(const double* R, bool inward, double* E) {                // src/spherical_utils.cpp:9-14
    double t[3] = {R[2], R[5], R[8] - 1.0};
    if (inward) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
    for (int j = 0; j < 3; j++) { E[j] = t[1] * R[6 + j] - t[2] * R[3 + j]; E[3 + j] = t[2] * R[j] - t[0] * R[6 + j]; E[6 + j] = t[0] * R[3 + j] - t[1] * R[j]; }
}
__device__ double det3_dev(const double* M) { return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]); }
__device__ void decompose_E_dev(const double* E, bool inward, double* r) {           // src/spherical_utils.cpp:16-66
    double a[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[3 * i + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        if (a[1] * a[1] + a[2] * a[2] + a[5] * a[5] < 1e-300) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (a[3 * p + q] == 0.0) continue;
            const double th = (a[3 * q + q] - a[3 * p + p]) / (2 * a[3 * p + q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0)), c = 1 / sqrt(t * t + 1), s = t * c;
            for (int k = 0; k < 3; k++) { const double x = a[3 * k + p], y = a[3 * k + q]; a[3 * k + p] = c * x - s * y; a[3 * k + q] = s * x + c * y; }
            for (int k = 0; k < 3; k++) { const double x = a[3 * p + k], y = a[3 * q + k]; a[3 * p + k] = c * x - s * y; a[3 * q + k] = s * x + c * y; }
            for (int k = 0; k < 3; k++) { const double x = V[3 * k + p], y = V[3 * k + q]; V[3 * k + p] = c * x - s * y; V[3 * k + q] = s * x + c * y; }
        }
    }
    // order eigenvalues descending
    double d[3] = {a[0], a[4], a[8]}; int o[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2 - i; j++) if (d[o[j]] < d[o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    double Vs[9]; for (int k = 0; k < 3; k++) for (int i = 0; i < 3; i++) Vs[3 * i + k] = V[3 * i + o[k]];
    double uu[3][3];
    for (int k = 0; k < 2; k++) { for (int i = 0; i < 3; i++) uu[k][i] = E[3 * i] * Vs[k] + E[3 * i + 1] * Vs[3 + k] + E[3 * i + 2] * Vs[6 + k];
                                  const double n = sqrt(uu[k][0] * uu[k][0] + uu[k][1] * uu[k][1] + uu[k][2] * uu[k][2]); for (int i = 0; i < 3; i++) uu[k][i] /= n; }
    const double dd = uu[0][0] * uu[1][0] + uu[0][1] * uu[1][1] + uu[0][2] * uu[1][2]; for (int i = 0; i < 3; i++) uu[1][i] -= dd * uu[0][i];
    const double nn = sqrt(uu[1][0] * uu[1][0] + uu[1][1] * uu[1][1] + uu[1][2] * uu[1][2]); for (int i = 0; i < 3; i++) uu[1][i] /= nn;
    uu[2][0] = uu[0][1] * uu[1][2] - uu[0][2] * uu[1][1]; uu[2][1] = uu[0][2] * uu[1][0] - uu[0][0] * uu[1][2]; uu[2][2] = uu[0][0] * uu[1][1] - uu[0][1] * uu[1][0];
    double U[9]; for (int k = 0; k < 3; k++) for (int i = 0; i < 3; i++) U[3 * i + k] = uu[k][i];
    if (det3_dev(U) < 0) for (int i = 0; i < 9; i++) U[i] = -U[i];
    if (det3_dev(Vs) < 0) for (int i = 0; i < 9; i++) Vs[i] = -Vs[i];
    const double D[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, DT[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double UD[9], R1[9], R2[9];
    mat3_mul(U, D, UD); mat3_mul_bt(UD, Vs, R1); mat3_mul(U, DT, UD); mat3_mul_bt(UD, Vs, R2);
    const double tu[3] = {U[2], U[5], U[8]};
    double t1[3] = {R1[2], R1[5], R1[8] - 1}, t2[3] = {R2[2], R2[5], R2[8] - 1};
    if (inward) for (int k = 0; k < 3; k++) { t1[k] = -t1[k]; t2[k] = -t2[k]; }
    const double s1 = fabs(dot3(t1, tu) / norm3(t1)), s2 = fabs(dot3(t2, tu) / norm3(t2));
    if (s1 > s2) so3ln(R1, r); else so3ln(R2, r);
}

__global__ void __launch_bounds__(256)
k_ransac_refine(const int* __restrict__ pair_ptr, const double* __restrict__ u, const double* __restrict__ v, double sq_thresh, int inward,
                int min_num_inliers, int do_lsq, double* __restrict__ bestE, double* __restrict__ bestScore, double* __restrict__ outR,
                unsigned char* __restrict__ inlier_mask, int* __restrict__ num_inliers) {
    __shared__ double red[10 * 4];
    __shared__ double sh[16];
    __shared__ int shi[2];
    const int pair = blockIdx.x;
    const int r0 = pair_ptr[pair], n = pair_ptr[pair + 1] - r0;
    const double* pu = u + (size_t)3 * r0; const double* pv = v + (size_t)3 * r0;
    double E[9]; for (int k = 0; k < 9; k++) E[k] = bestE[9 * (size_t)pair + k];
    const double tz = inward ? 1.0 : -1.0;
    const bool have = bestScore[pair] < 1e308 && n >= 3;
    if (have && do_lsq) {
        // ---- LeastSquares on the inliers of the best model (Ceres LM rules: lm.hpp of the oracle / TrustRegionMinimizer)
        if (threadIdx.x == 0) { double r[3]; decompose_E_dev(E, inward != 0, r); sh[0] = r[0]; sh[1] = r[1]; sh[2] = r[2]; }
        __syncthreads();
        double x[3] = {sh[0], sh[1], sh[2]};
        __syncthreads();
        double radius = 1e4, decrease = 2.0, scale[3] = {1, 1, 1}, x_cost = 0, x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        double A[6], g[3];
        // linearise at x over the inliers of the best minimal model (a fixed set, ransac.h:257-259); every lane ends up with
        // the same sums, so the control flow below is uniform across the workgroup
        auto linearize = [&]() {
            double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // JtJ (00 01 02 11 12 22), Jtr (3), cost
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                if (!(sampson_err(E, pu + 3 * i, pv + 3 * i) < sq_thresh)) continue;
                typedef Dual<3> D3; D3 r1[3] = {D3(x[0], 0), D3(x[1], 1), D3(x[2], 2)}, res;
                sampson_residual_r<D3>(r1, tz, pu + 3 * i, pv + 3 * i, &res);
                const double j0 = res.v[0] * scale[0], j1 = res.v[1] * scale[1], j2 = res.v[2] * scale[2];
                acc[0] += j0 * j0; acc[1] += j0 * j1; acc[2] += j0 * j2; acc[3] += j1 * j1; acc[4] += j1 * j2; acc[5] += j2 * j2;
                acc[6] += j0 * res.a; acc[7] += j1 * res.a; acc[8] += j2 * res.a; acc[9] += 0.5 * res.a * res.a;
            }
            block_sum<10>(acc, red);
            if (threadIdx.x == 0) for (int k = 0; k < 10; k++) sh[k] = acc[k];
            __syncthreads();
            for (int k = 0; k < 6; k++) A[k] = sh[k];
            g[0] = sh[6]; g[1] = sh[7]; g[2] = sh[8]; x_cost = sh[9];
            __syncthreads();
        };
        linearize();
        // Jacobi scaling from the iteration-0 Jacobian
        scale[0] = 1.0 / (1.0 + sqrt(A[0])); scale[1] = 1.0 / (1.0 + sqrt(A[3])); scale[2] = 1.0 / (1.0 + sqrt(A[5]));
        A[0] *= scale[0] * scale[0]; A[1] *= scale[0] * scale[1]; A[2] *= scale[0] * scale[2]; A[3] *= scale[1] * scale[1]; A[4] *= scale[1] * scale[2]; A[5] *= scale[2] * scale[2];
        g[0] *= scale[0]; g[1] *= scale[1]; g[2] *= scale[2];
        int iteration = 0, invalid = 0; bool last_ok = true;
        while (true) {
            if (iteration >= 200) break;                                              // src/spherical_estimator.cpp:148
            const double gmax = fmax(fabs(g[0] / scale[0]), fmax(fabs(g[1] / scale[1]), fabs(g[2] / scale[2])));
            if (last_ok && gmax <= 1e-10) break;
            if (radius <= 1e-32) break;
            iteration++;
            double Ad[6] = {A[0], A[1], A[2], A[3], A[4], A[5]};
            Ad[0] += fmin(fmax(A[0], 1e-6), 1e32) / radius; Ad[3] += fmin(fmax(A[3], 1e-6), 1e32) / radius; Ad[5] += fmin(fmax(A[5], 1e-6), 1e32) / radius;
            double Ai[6]; sym3_inverse(Ad, Ai);
            const double st[3] = {-(Ai[0] * g[0] + Ai[1] * g[1] + Ai[2] * g[2]), -(Ai[1] * g[0] + Ai[3] * g[1] + Ai[4] * g[2]), -(Ai[2] * g[0] + Ai[4] * g[1] + Ai[5] * g[2])};
            const double sAs = A[0] * st[0] * st[0] + A[3] * st[1] * st[1] + A[5] * st[2] * st[2] + 2 * (A[1] * st[0] * st[1] + A[2] * st[0] * st[2] + A[4] * st[1] * st[2]);
            const double model = -((g[0] * st[0] + g[1] * st[1] + g[2] * st[2]) + 0.5 * sAs);   // -(Js)^T (r + Js/2)
            if (!(model > 0.0) || !isfinite(model)) {
                if (++invalid >= 10) break;                                           // max_num_consecutive_invalid_steps, :149
                radius /= decrease; decrease *= 2.0; last_ok = false; continue;
            }
            invalid = 0;
            const double xc[3] = {x[0] + st[0] * scale[0], x[1] + st[1] * scale[1], x[2] + st[2] * scale[2]};
            double c[1] = {0.0};
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                if (!(sampson_err(E, pu + 3 * i, pv + 3 * i) < sq_thresh)) continue;
                double r; sampson_residual_r<double>(xc, tz, pu + 3 * i, pv + 3 * i, &r); c[0] += 0.5 * r * r;
            }
            block_sum<1>(c, red);
            if (threadIdx.x == 0) sh[10] = c[0];
            __syncthreads();
            double cand = sh[10];
            __syncthreads();
            if (!isfinite(cand)) cand = 1.79e308;
            const double step_norm = sqrt((xc[0] - x[0]) * (xc[0] - x[0]) + (xc[1] - x[1]) * (xc[1] - x[1]) + (xc[2] - x[2]) * (xc[2] - x[2]));
            if (step_norm <= 1e-8 * (x_norm + 1e-8)) break;
            const double change = x_cost - cand;
            if (fabs(change) <= 1e-6 * x_cost) break;
            const double rho = (cand >= 1.79e308) ? -1.79e308 : change / model;
            if (rho > 1e-3) {
                x[0] = xc[0]; x[1] = xc[1]; x[2] = xc[2]; x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
                linearize();
                radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3))); decrease = 2.0; last_ok = true;
            } else { radius /= decrease; decrease *= 2.0; last_ok = false; }
        }
        // refined model replaces the best one only if it scores better (ransac.h:262-270)
        double Rm[9]; so3exp(x, Rm); double E2[9]; make_E_dev(Rm, inward != 0, E2);
        double c[1] = {0.0};
        for (int i = threadIdx.x; i < n; i += blockDim.x) c[0] += fmin(sampson_err(E2, pu + 3 * i, pv + 3 * i), sq_thresh);
        block_sum<1>(c, red);
        if (threadIdx.x == 0) sh[11] = c[0];
        __syncthreads();
        if (sh[11] < bestScore[pair]) { for (int k = 0; k < 9; k++) E[k] = E2[k]; if (threadIdx.x == 0) { bestScore[pair] = sh[11]; for (int k = 0; k < 9; k++) bestE[9 * (size_t)pair + k] = E2[k]; } }
        __syncthreads();
    }
    // inlier mask (examples/spherical_sfm_tools.cpp:388-392) and rotation (:410-419)
    double cnt[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const bool in = have && sampson_err(E, pu + 3 * i, pv + 3 * i) < sq_thresh;
        inlier_mask[r0 + i] = in ? 1 : 0; cnt[0] += in ? 1.0 : 0.0;
    }
    block_sum<1>(cnt, red);
    if (threadIdx.x == 0) {
        const int nin = (int)cnt[0]; num_inliers[pair] = nin;
        double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (have && nin > min_num_inliers) { double r[3]; decompose_E_dev(E, inward != 0, r); so3exp(r, Rm); }
        for (int k = 0; k < 9; k++) outR[9 * (size_t)pair + k] = Rm[k];
    }
    (void)shi;
}

// ---- focal-length search around the pose graph ------------------------------------------------
// One workgroup per trial focal: the loop_constraint_cost_fn of examples/spherical_sfm_tools.cpp:1138-1157 --
//   transform_image_matches (:1118-1132): E_new = T E T, T = diag(f/f0, f/f0, 1); decompose -> r_new -> R = so3exp(r_new)
//   initialize_rotations_sequential (:794-813): chain over the matches (k-1, k); a camera without such a match keeps the identity
//   get_cost (src/uncalibrated_pose_graph.cpp:116-145): 1/2 sum SoftLOne_0.03(|s log(R1 R0^T R^T)|^2), s = 1 / max |log R_rel|
// Lanes share the edges (decomposition, residuals), lane 0 walks the rotation chain.  Per-trial scratch in global memory.
__global__ void __launch_bounds__(256)
k_focal_trials(int n, int E, const int* __restrict__ e0, const int* __restrict__ e1, const int* __restrict__ chain_edge,
               const double* __restrict__ Es /*[E*9] row-major*/, int inward, double focal_guess, const double* __restrict__ focals,
               double* __restrict__ rnew_all /*[trials*E*3]*/, double* __restrict__ x_all /*[trials*n*3]*/,
               double* __restrict__ rot_all /*[trials*n*9] row-major*/, double* __restrict__ costs) {
    __shared__ double red[4]; __shared__ double s_scale;
    const int trial = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double f = focals[trial], tf = f / focal_guess;
    double* rnew = rnew_all + (size_t)trial * E * 3; double* x = x_all + (size_t)trial * n * 3; double* rot = rot_all + (size_t)trial * n * 9;
    double mx = 0.0;
    for (int e = tid; e < E; e += nt) {
        double En[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) En[3 * i + j] = Es[9 * (size_t)e + 3 * i + j] * ((i < 2) ? tf : 1.0) * ((j < 2) ? tf : 1.0);
        double r[3]; decompose_E_dev(En, inward != 0, r);
        // the reference stores so3exp(r_new) and get_cost takes so3ln of it again
        double Rm[9], rr[3]; so3exp(r, Rm); so3ln(Rm, rr);
        rnew[3 * e] = rr[0]; rnew[3 * e + 1] = rr[1]; rnew[3 * e + 2] = rr[2];
        mx = fmax(mx, norm3(rr));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) { double m = 0; for (int w = 0; w < (nt >> 6); w++) m = fmax(m, red[w]); s_scale = 1.0 / m; }
    // chain (lane 0; global writes of this block are visible to it after the barrier above)
    if (tid == 0) {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; k++) rot[k] = R[k];
        for (int idx = 1; idx < n; idx++) {
            const int e = chain_edge[idx];
            double* dst = rot + 9 * (size_t)idx;
            if (e >= 0) {
                double Rm[9], Rn[9]; so3exp(rnew + 3 * e, Rm); mat3_mul(Rm, R, Rn);
                for (int k = 0; k < 9; k++) { R[k] = Rn[k]; dst[k] = Rn[k]; }
            } else { for (int k = 0; k < 9; k++) dst[k] = (k % 4 == 0) ? 1.0 : 0.0; }
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) so3ln(rot + 9 * (size_t)i, x + 3 * i);
    __syncthreads();
    const double scale = s_scale;
    double c = 0.0;
    for (int e = tid; e < E; e += nt) {
        double Rm[9], R0[9], R1[9], A[9], C[9], res[3];
        angle_axis_to_matrix(rnew + 3 * e, Rm); angle_axis_to_matrix(x + 3 * e0[e], R0); angle_axis_to_matrix(x + 3 * e1[e], R1);
        mat3_mul_bt(R1, R0, A); mat3_mul_bt(A, Rm, C);
        matrix_to_angle_axis(C, res);
        const double s2 = scale * scale * (res[0] * res[0] + res[1] * res[1] + res[2] * res[2]);
        double rho0, rho1; robust_loss(2, 0.03, s2, rho0, rho1);
        c += 0.5 * rho0;
    }
    c = wave_sum(c);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int w = 0; w < (nt >> 6); w++) t += red[w]; costs[trial] = t; }
}

}  // namespace ssfm
using namespace ssfm;

static void rm_to_cm(const double* rm, double* cm) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cm[i + 3 * j] = rm[3 * i + j]; }

extern "C" void ssfm_ransac_default_options(ssfm_ransac_options* o) {
    o->num_hypotheses = 1024;          // fixed budget per pair (the reference runs 100..10000 adaptive iterations, ransac.h:49-54)
    o->seed = 0;                       // RansacOptions::random_seed_
    o->min_num_inliers = 0;            // estimate_pairwise's acceptance test (spherical_sfm_tools.cpp:410)
    o->final_least_squares = 1;        // spherical_sfm_tools.cpp:318
    o->inward = 0;
    o->use_poly_solver = 0;            // estimate_pairwise passes use_poly_solver = false (spherical_sfm_tools.cpp:378)
}

static int ransac_batch_impl(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v, double sq_thresh,
                             const ssfm_ransac_options& O, const int32_t* pair_id, double* E_out, double* R_out, uint8_t* inlier_mask,
                             int32_t* num_inliers, double* scores) {
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int total = pair_ptr[num_pairs];
    int max_n = 0; for (int p = 0; p < num_pairs; p++) max_n = std::max(max_n, pair_ptr[p + 1] - pair_ptr[p]);
    const size_t lds = (size_t)6 * max_n * sizeof(double);
    if (lds > 150 * 1024) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch: more than 3200 correspondences in one pair");
    DevBuf<int> dptr, dnin, dpid; DevBuf<double> du, dv, dE, dS, dR; DevBuf<unsigned char> dmask;
    std::vector<int> ptr(pair_ptr, pair_ptr + num_pairs + 1);
    SSFM_HIP_CHECK(ctx, upload(dptr, ptr, st));
    if (pair_id) { std::vector<int> ids(pair_id, pair_id + num_pairs); SSFM_HIP_CHECK(ctx, upload(dpid, ids, st)); }
    SSFM_HIP_CHECK(ctx, du.alloc((size_t)3 * total)); SSFM_HIP_CHECK(ctx, dv.alloc((size_t)3 * total));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(du.p, u, (size_t)3 * total * sizeof(double), hipMemcpyHostToDevice, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(dv.p, v, (size_t)3 * total * sizeof(double), hipMemcpyHostToDevice, st));
    SSFM_HIP_CHECK(ctx, dE.alloc((size_t)9 * num_pairs)); SSFM_HIP_CHECK(ctx, dS.alloc(num_pairs)); SSFM_HIP_CHECK(ctx, dR.alloc((size_t)9 * num_pairs));
    SSFM_HIP_CHECK(ctx, dmask.alloc(total)); SSFM_HIP_CHECK(ctx, dnin.alloc(num_pairs));
    if (O.use_poly_solver) {
        if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ransac_hypotheses<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_ransac_hypotheses<true>, dim3(num_pairs), dim3(256), lds, st, dptr.p, du.p, dv.p, sq_thresh, O.num_hypotheses, (unsigned long long)O.seed, pair_id ? dpid.p : nullptr, dE.p, dS.p);
    } else {
        if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ransac_hypotheses<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_ransac_hypotheses<false>, dim3(num_pairs), dim3(256), lds, st, dptr.p, du.p, dv.p, sq_thresh, O.num_hypotheses, (unsigned long long)O.seed, pair_id ? dpid.p : nullptr, dE.p, dS.p);
    }
    hipLaunchKernelGGL(k_ransac_refine, dim3(num_pairs), dim3(256), 0, st, dptr.p, du.p, dv.p, sq_thresh, O.inward, O.min_num_inliers, O.final_least_squares,
                       dE.p, dS.p, dR.p, dmask.p, dnin.p);
    std::vector<double> hE((size_t)9 * num_pairs), hR((size_t)9 * num_pairs), hS(num_pairs);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hE.data(), dE.p, hE.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hR.data(), dR.p, hR.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hS.data(), dS.p, hS.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    if (inlier_mask) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(inlier_mask, dmask.p, total, hipMemcpyDeviceToHost, st));
    if (num_inliers) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(num_inliers, dnin.p, num_pairs * sizeof(int), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (int p = 0; p < num_pairs; p++) { if (E_out) rm_to_cm(&hE[9 * (size_t)p], E_out + 9 * (size_t)p); if (R_out) rm_to_cm(&hR[9 * (size_t)p], R_out + 9 * (size_t)p); if (scores) scores[p] = hS[p]; }
    dptr.free(); dnin.free(); dpid.free(); du.free(); dv.free(); dE.free(); dS.free(); dR.free(); dmask.free();
    return SSFM_OK;
}

extern "C" int ssfm_ransac_batch(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v, double sq_thresh,
                                 const ssfm_ransac_options* opt, double* E_out, double* R_out, uint8_t* inlier_mask, int32_t* num_inliers,
                                 double* scores) {
    if (!ctx || !pair_ptr || !u || !v || num_pairs <= 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch: bad arguments");
    ssfm_ransac_options O; if (opt) O = *opt; else ssfm_ransac_default_options(&O);
    return ransac_batch_impl(ctx, num_pairs, pair_ptr, u, v, sq_thresh, O, nullptr, E_out, R_out, inlier_mask, num_inliers, scores);
}

// Multi-GPU estimate_pairwise (BASELINE configs[3]): image pairs are independent, so rank r of the context's
// communicator takes pairs r, r + nranks, ... (round robin keeps neighbouring-frame pairs, which have the most correspondences,
// spread over the ranks), runs them as one local batch with the random streams of their global indices, and one sum all-reduce
// of a zero-filled result table hands every rank every pair: [E 9 | R 9 | score | num_inliers] per pair + the inlier masks packed
// 32 per double (exact: one rank contributes each word).  Same results as ssfm_ransac_batch on one GPU, bit for bit.
extern "C" int ssfm_ransac_batch_sharded(ssfm_ctx* ctx, int32_t num_pairs, const int32_t* pair_ptr, const double* u, const double* v,
                                         double sq_thresh, const ssfm_ransac_options* opt, double* E_out, double* R_out,
                                         uint8_t* inlier_mask, int32_t* num_inliers, double* scores) {
    if (!ctx || !pair_ptr || !u || !v || num_pairs <= 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ransac_batch_sharded: bad arguments");
    ssfm_ransac_options O; if (opt) O = *opt; else ssfm_ransac_default_options(&O);
    const int nr = ctx->collective ? ctx->nranks : 1, rk = ctx->collective ? ctx->rank : 0;
    if (nr == 1 && !ctx->collective) return ransac_batch_impl(ctx, num_pairs, pair_ptr, u, v, sq_thresh, O, nullptr, E_out, R_out, inlier_mask, num_inliers, scores);
    const int total = pair_ptr[num_pairs];
    // local batch
    std::vector<int> ids, lptr(1, 0);
    for (int p = rk; p < num_pairs; p += nr) { ids.push_back(p); lptr.push_back(lptr.back() + pair_ptr[p + 1] - pair_ptr[p]); }
    const int nl = (int)ids.size(), ltotal = lptr.back();
    std::vector<double> lu((size_t)3 * ltotal), lv((size_t)3 * ltotal), lE((size_t)9 * nl), lR((size_t)9 * nl), lS(nl);
    std::vector<uint8_t> lmask(ltotal); std::vector<int> lnin(nl);
    for (int i = 0; i < nl; i++) {
        const size_t src = (size_t)3 * pair_ptr[ids[i]], cnt = (size_t)3 * (lptr[i + 1] - lptr[i]);
        if (cnt) { std::memcpy(&lu[(size_t)3 * lptr[i]], u + src, cnt * sizeof(double)); std::memcpy(&lv[(size_t)3 * lptr[i]], v + src, cnt * sizeof(double)); }
    }
    if (nl > 0) {
        const int rc = ransac_batch_impl(ctx, nl, lptr.data(), lu.data(), lv.data(), sq_thresh, O, ids.data(), lE.data(), lR.data(), lmask.data(), lnin.data(), lS.data());
        if (rc) return rc;
    }
    // result table; every mask word belongs to exactly one pair's rank only if words do not straddle pairs: pack per pair
    std::vector<size_t> wptr(num_pairs + 1, 0);
    for (int p = 0; p < num_pairs; p++) wptr[p + 1] = wptr[p] + (size_t)(pair_ptr[p + 1] - pair_ptr[p] + 31) / 32;
    const size_t per = 20, n_tab = per * num_pairs + wptr[num_pairs];
    std::vector<double> tab(n_tab, 0.0);
    for (int i = 0; i < nl; i++) {
        const int p = ids[i]; double* t = &tab[per * (size_t)p];
        std::memcpy(t, &lE[9 * (size_t)i], 9 * sizeof(double)); std::memcpy(t + 9, &lR[9 * (size_t)i], 9 * sizeof(double));
        t[18] = lS[i]; t[19] = (double)lnin[i];
        double* w = &tab[per * (size_t)num_pairs + wptr[p]];
        const int n = lptr[i + 1] - lptr[i];
        for (int k = 0; k < n; k += 32) { uint32_t bits = 0; for (int q = 0; q < 32 && k + q < n; q++) bits |= (uint32_t)(lmask[lptr[i] + k + q] != 0) << q; w[k / 32] = (double)bits; }
    }
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf<double> dtab;
    SSFM_HIP_CHECK(ctx, upload(dtab, tab, ctx->stream));
    { const int rc = ctx_allreduce(ctx, dtab.p, n_tab, ncclSum); if (rc) { dtab.free(); return rc; } }
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(tab.data(), dtab.p, n_tab * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    dtab.free();
    for (int p = 0; p < num_pairs; p++) {
        const double* t = &tab[per * (size_t)p];
        if (E_out) std::memcpy(E_out + 9 * (size_t)p, t, 9 * sizeof(double));
        if (R_out) std::memcpy(R_out + 9 * (size_t)p, t + 9, 9 * sizeof(double));
        if (scores) scores[p] = t[18];
        if (num_inliers) num_inliers[p] = (int32_t)t[19];
        if (inlier_mask) {
            const double* w = &tab[per * (size_t)num_pairs + wptr[p]];
            const int n = pair_ptr[p + 1] - pair_ptr[p];
            for (int k = 0; k < n; k++) inlier_mask[pair_ptr[p] + k] = (uint8_t)(((uint32_t)w[k / 32] >> (k % 32)) & 1u);
        }
    }
    (void)total;
    return SSFM_OK;
}

// parity probe: the minimal solver on given 3-point samples.  Es: [S*36] (4 column-major 3x3 per sample), counts: [S]
static int solver_probe(ssfm_ctx* ctx, bool poly, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples, double* Es, int32_t* counts) {
    if (!ctx || S <= 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_spherical_solver_probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<double> du, dv, dE; DevBuf<int> ds, dc;
    std::vector<double> hu(u, u + (size_t)3 * n), hv(v, v + (size_t)3 * n); std::vector<int> hs(samples, samples + (size_t)3 * S);
    SSFM_HIP_CHECK(ctx, upload(du, hu, st)); SSFM_HIP_CHECK(ctx, upload(dv, hv, st)); SSFM_HIP_CHECK(ctx, upload(ds, hs, st));
    SSFM_HIP_CHECK(ctx, dE.alloc((size_t)36 * S)); SSFM_HIP_CHECK(ctx, dc.alloc(S));
    if (poly) hipLaunchKernelGGL(k_solver_probe<true>, dim3((S + 63) / 64), dim3(64), 0, st, S, ds.p, du.p, dv.p, dE.p, dc.p);
    else hipLaunchKernelGGL(k_solver_probe<false>, dim3((S + 63) / 64), dim3(64), 0, st, S, ds.p, du.p, dv.p, dE.p, dc.p);
    std::vector<double> hE((size_t)36 * S);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hE.data(), dE.p, hE.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(counts, dc.p, S * sizeof(int), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (int s = 0; s < S; s++) for (int m = 0; m < 4; m++) rm_to_cm(&hE[36 * (size_t)s + 9 * m], Es + 36 * (size_t)s + 9 * m);
    du.free(); dv.free(); dE.free(); ds.free(); dc.free();
    return SSFM_OK;
}
extern "C" int ssfm_spherical_solver_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples,
                                           double* Es, int32_t* counts) { return solver_probe(ctx, false, n, u, v, S, samples, Es, counts); }
extern "C" int ssfm_spherical_solver_poly_probe(ssfm_ctx* ctx, int32_t n, const double* u, const double* v, int32_t S, const int32_t* samples,
                                                double* Es, int32_t* counts) { return solver_probe(ctx, true, n, u, v, S, samples, Es, counts); }

// find_best_focal_length_random without its random draw (examples/spherical_sfm_tools.cpp:1418-1496): the caller supplies the
// trial focals; costs[t] = loop_constraint_cost_fn(focals[t]); best_trial = first minimum; rotations_best = the sequential
// initialisation at that focal (column-major 3x3 per camera), ready for ssfm_posegraph_focal_solve (run_optimization, :1160-1188).
extern "C" int ssfm_focal_search(ssfm_ctx* ctx, int32_t n, int32_t E, const int32_t* index0, const int32_t* index1, const double* rel_rotations,
                                 int32_t inward, double focal_guess, int32_t num_trials, const double* focals, double* costs,
                                 int32_t* best_trial, double* rotations_best, double* rel_rotations_best) {
    if (!ctx || n <= 0 || E <= 0 || num_trials <= 0 || !index0 || !index1 || !rel_rotations || !focals)
        return fail(ctx, SSFM_ERR_INVALID, "ssfm_focal_search: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // Es[i] = make_spherical_essential_matrix(R_i, inward) (:1429-1433), row-major for the device
    std::vector<double> Es((size_t)9 * E); std::vector<int> e0(index0, index0 + E), e1(index1, index1 + E), chain(n, -1);
    for (int e = 0; e < E; e++) {
        double Rm[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rm[3 * i + j] = rel_rotations[9 * (size_t)e + i + 3 * j];
        double t[3] = {Rm[2], Rm[5], Rm[8] - 1.0}; if (inward) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
        double* Em = &Es[9 * (size_t)e];
        for (int j = 0; j < 3; j++) { Em[j] = t[1] * Rm[6 + j] - t[2] * Rm[3 + j]; Em[3 + j] = t[2] * Rm[j] - t[0] * Rm[6 + j]; Em[6 + j] = t[0] * Rm[3 + j] - t[1] * Rm[j]; }
        if (index1[e] >= 1 && index1[e] < n && index0[e] == index1[e] - 1 && chain[index1[e]] < 0) chain[index1[e]] = e;      // first match (k-1, k), :804-810
    }
    DevBuf<double> dEs, dF, dR, dX, dRot, dC; DevBuf<int> de0, de1, dch;
    std::vector<double> fv(focals, focals + num_trials);
    SSFM_HIP_CHECK(ctx, upload(dEs, Es, st)); SSFM_HIP_CHECK(ctx, upload(dF, fv, st)); SSFM_HIP_CHECK(ctx, upload(de0, e0, st));
    SSFM_HIP_CHECK(ctx, upload(de1, e1, st)); SSFM_HIP_CHECK(ctx, upload(dch, chain, st));
    SSFM_HIP_CHECK(ctx, dR.alloc((size_t)num_trials * E * 3)); SSFM_HIP_CHECK(ctx, dX.alloc((size_t)num_trials * n * 3));
    SSFM_HIP_CHECK(ctx, dRot.alloc((size_t)num_trials * n * 9)); SSFM_HIP_CHECK(ctx, dC.alloc(num_trials));
    hipLaunchKernelGGL(k_focal_trials, dim3(num_trials), dim3(256), 0, st, n, E, de0.p, de1.p, dch.p, dEs.p, inward, focal_guess, dF.p, dR.p, dX.p, dRot.p, dC.p);
    std::vector<double> hc(num_trials);
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), dC.p, hc.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    int best = 0; for (int t = 1; t < num_trials; t++) if (hc[t] < hc[best]) best = t;                 // :1467-1474 (strict <, first minimum)
    if (costs) std::memcpy(costs, hc.data(), hc.size() * sizeof(double));
    if (best_trial) *best_trial = best;
    if (rotations_best) {
        std::vector<double> hr((size_t)n * 9);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), dRot.p + (size_t)best * n * 9, hr.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        for (int i = 0; i < n; i++) rm_to_cm(&hr[9 * (size_t)i], rotations_best + 9 * (size_t)i);
    }
    if (rel_rotations_best) {                              // the matches as transform_image_matches leaves them at the best focal
        std::vector<double> hr((size_t)E * 3);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), dR.p + (size_t)best * E * 3, hr.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        for (int e = 0; e < E; e++) { double Rm[9]; so3exp(&hr[3 * (size_t)e], Rm); rm_to_cm(Rm, rel_rotations_best + 9 * (size_t)e); }
    }
    dEs.free(); dF.free(); dR.free(); dX.free(); dRot.free(); dC.free(); de0.free(); de1.free(); dch.free();
    return SSFM_OK;
}
