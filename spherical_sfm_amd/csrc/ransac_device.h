// spherical_sfm_amd -- device functions of the spherical relative-pose path, shared by the
// fixed-budget batch kernels (ransac.hip) and the reference-trace LO-MSAC kernel (lomsac.hip):
//   EvaluateModelOnPoint (Sampson)                        src/spherical_estimator.cpp:67-78
//   spherical_solver_action_matrix / _polynomial          src/spherical_solvers.cpp:102-311, 313-660 (SolveQuartic :15-69)
//   SampsonError                                          src/spherical_estimator.cpp:23-65
//   make / decompose_spherical_essential_matrix           src/spherical_utils.cpp:9-66
// 3x3 matrices are row-major here.
#pragma once
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

// Nullspace basis of the N x 6 constraint matrix (src/spherical_solvers.cpp:113-125): rows of A (:119), Householder QR of A^T with
// column pivoting (Eigen colPivHouseholderQr: the remaining column of largest norm is eliminated next), B = columns 3..5 of Q.
// N is a compile-time capacity; columns >= n_valid are zero and turn into identity reflectors, so one instantiation with N = 9
// serves every non-minimal sample size (NonMinimalSolver draws 4..9 rays, ransac.h:369-373).
template <int N>
__device__ void spherical_nullspace(const double* uN, const double* vN, int n_valid, double (*B)[3]) {
    double At[6][N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const bool on = i < n_valid;
        const double* u = uN + 3 * (on ? i : 0); const double* v = vN + 3 * (on ? i : 0);
        const double m = on ? 1.0 : 0.0;
        At[0][i] = m * (u[0] * v[0] - u[1] * v[1]); At[1][i] = m * (u[0] * v[1] + u[1] * v[0]); At[2][i] = m * (u[2] * v[0]);
        At[3][i] = m * (u[2] * v[1]); At[4][i] = m * (u[0] * v[2]); At[5][i] = m * (u[1] * v[2]);
    }
    constexpr int STEPS = (N < 6) ? N : 6;
    double hv[STEPS][6], tau[STEPS];
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
        // pivot: first column j >= k of largest remaining norm
        double best = -1.0; int piv = k;
#pragma unroll
        for (int j = 0; j < N; j++) if (j >= k) {
            double sn = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) if (i >= k) sn += At[i][j] * At[i][j];
            if (sn > best) { best = sn; piv = j; }
        }
#pragma unroll
        for (int j = 0; j < N; j++) if (j > k && j == piv) {
#pragma unroll
            for (int i = 0; i < 6; i++) { const double t = At[i][k]; At[i][k] = At[i][j]; At[i][j] = t; }
        }
        const double alpha = sqrt(best);
        const double x0 = At[k][k], beta = (x0 >= 0) ? -alpha : alpha;
        double vn = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) { hv[k][i] = (i < k || alpha == 0.0) ? 0.0 : ((i == k) ? x0 - beta : At[i][k]); vn += hv[k][i] * hv[k][i]; }
        tau[k] = (vn > 0) ? 2.0 / vn : 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) if (j >= k) {
            double d = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) d += hv[k][i] * At[i][j];
            d *= tau[k];
#pragma unroll
            for (int i = 0; i < 6; i++) At[i][j] -= d * hv[k][i];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double q[6] = {0, 0, 0, 0, 0, 0}; q[3 + c] = 1.0;
#pragma unroll
        for (int k = STEPS - 1; k >= 0; k--) {
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
}

__device__ __forceinline__ bool essential_from_basis(const double (*B)[3], double bx, double by, double* E) {   // src/spherical_solvers.cpp:296-305
    double ps[6];
#pragma unroll
    for (int k = 0; k < 6; k++) ps[k] = B[k][0] * bx + B[k][1] * by + B[k][2];
    E[0] = ps[0]; E[1] = ps[1]; E[2] = ps[2]; E[3] = ps[1]; E[4] = -ps[0]; E[5] = ps[3]; E[6] = ps[4]; E[7] = ps[5]; E[8] = 0.0;
    double n2 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) n2 += E[k] * E[k];
    const double nrm = sqrt(n2);
#pragma unroll
    for (int k = 0; k < 9; k++) E[k] /= nrm;
    return (n2 > 0.0) && isfinite(n2);
}

// The models of a sample from its nullspace basis.  Es: up to 4 solutions (row-major 3x3, unit Frobenius norm); returns the count.
// POLY = false: action-matrix variant (src/spherical_solvers.cpp:102-311); POLY = true: quartic variant (:313-660), whose
// constraint matrix is the same six rows (times 1/2) with the monomials ordered [x^3 x^2y xy^2 x^2z xyz xz^2 | y^3 y^2z yz^2 z^3].
// ALL = false: real solutions only, roots polished by two Newton steps (the fixed-budget batch: a complex eigen-pair is never a model).
// ALL = true : the reference's behaviour -- always four candidates, the REAL PARTS of complex eigenvectors / quartic roots included
//              (:296, :629-640), in root order, unpolished: what LocallyOptimizedMSAC scores, so the reference-trace kernel uses it.
template <bool POLY, bool ALL>
__device__ int spherical_models_from_basis(const double (*B)[3], double* Es) {
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
        if (!ALL && (qa == 0.0 || !isfinite(qa + qb + qc + qd + qe))) return 0;
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
            double yv = roots[s].r;
            if (!ALL) {
                if (!isfinite(roots[s].r) || !(fabs(roots[s].i) <= 1e-7 * scale)) continue;     // complex pair: not a model
#pragma unroll
                for (int it = 0; it < 2; it++) {                   // Newton polish on the real axis
                    const double pv = (((qa * yv + qb) * yv + qc) * yv + qd) * yv + qe, dp = ((4 * qa * yv + 3 * qb) * yv + 2 * qc) * yv + qd;
                    if (dp != 0.0) yv -= pv / dp;
                }
            }
            const double y2 = yv * yv, y3 = y2 * yv;
            const double xv = -G5[0] * y3 - G5[1] * y2 - G5[2] * yv - G5[3];
            const bool ok = essential_from_basis(B, xv, yv, Es + 9 * count);
            if (ALL || ok) count++;
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
    // simple roots converge quadratically: once the largest correction is below 1e-13 of the root bound, one more sweep leaves every root at
    // rounding level (a 1e-15 test is met by rounding noise only now and then -- 2 % of the samples ran all sweeps, and a wave waits for its slowest lane)
    bool last = false;
    for (int it = 0; it < (ALL ? 200 : 100); it++) {
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
            zr[i] -= qr_; zi[i] -= qi; change = fmax(change, ALL ? hypot(qr_, qi) : fabs(qr_) + fabs(qi));
        }
        if (last) break;
        if (change < 1e-13 * scale) last = true;
    }
    int count = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (ALL) {
            // eigenvector (v0, v1, v2, 1) of the possibly complex root: v1 = l, rows 1, 2 of (M - l I) v = 0 give v0, v2; real parts kept
            const cplx l = cmk(zr[s], zi[s]);
            const cplx b1 = cscale(cadd(cmul(csub(cmk(M[1][1]), l), l), cmk(M[1][3])), -1.0);
            const cplx a22 = csub(cmk(M[2][2]), l), b2 = cscale(cadd(cscale(l, M[2][1]), cmk(M[2][3])), -1.0);
            const cplx det = csub(cscale(a22, M[1][0]), cmk(M[1][2] * M[2][0]));
            cplx v2 = cmk(0.0);
            if (det.r != 0.0 || det.i != 0.0) v2 = cdiv(csub(cscale(b2, M[1][0]), cscale(b1, M[2][0])), det);
            essential_from_basis(B, l.r, v2.r, Es + 9 * count);
            count++;
            continue;
        }
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
        if (essential_from_basis(B, l, v2, Es + 9 * count)) count++;
    }
    return count;
}

// Minimal solver for one 3-point sample (real solutions only), the fixed-budget batch's hypothesis generator.
template <bool POLY>
__device__ int spherical_minimal_solver(const double* u3, const double* v3, double* Es) {
    double B[6][3];
    spherical_nullspace<3>(u3, v3, 3, B);
    return spherical_models_from_basis<POLY, false>(B, Es);
}

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


// ---- workgroup-cooperative pieces (every thread of the block calls them with the same arguments: uniform control flow) -----------

// GetInliers (include/RansacLib/ransac.h:311-336): indices i with Sampson(E, i) < thresh, ascending, into list; returns the count.
// s_cnt: LDS int[blockDim/64].
__device__ int block_inlier_list(const double* E, const double* pu, const double* pv, int n, double thresh, int* list, int* s_cnt) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    int total = 0;
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        const bool in = (i < n) && (sampson_err(E, pu + 3 * i, pv + 3 * i) < thresh);
        const unsigned long long b = __ballot(in);
        if (lane == 0) s_cnt[w] = __popcll(b);
        __syncthreads();
        int off = total, tile = 0;
        for (int k = 0; k < nw; k++) { const int c = s_cnt[k]; if (k < w) off += c; tile += c; }
        if (in) list[off + __popcll(b & ((1ull << lane) - 1ull))] = i;
        total += tile;
        __syncthreads();
    }
    return total;
}

// ScoreModel (ransac.h:295-310): sum_i min(Sampson(E, i), thresh); every thread returns the block-wide sum.
__device__ double block_msac_score(const double* E, const double* pu, const double* pv, int n, double thresh, double* red, double* s_bcast) {
    double c[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) c[0] += fmin(sampson_err(E, pu + 3 * i, pv + 3 * i), thresh);
    block_sum<1>(c, red);
    if (threadIdx.x == 0) *s_bcast = c[0];
    __syncthreads();
    const double r = *s_bcast;
    __syncthreads();
    return r;
}

// SphericalEstimator::LeastSquares lives in sampson_lsq.h (six free parameters [r1; t1], src/spherical_estimator.cpp:140-144).

// ---- std::mt19937 + std::uniform_int_distribution<int> of libstdc++ (GCC >= 11), as RansacLib draws them
// (include/RansacLib/sampling.h:46-135, utils.h:48-73).  State: 624 words in LDS, position in a register that every thread keeps.
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18); return y;
}
__device__ __forceinline__ unsigned mt_mix(unsigned a, unsigned b) { const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu); return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
// the 624-word state transition, cooperative (three dependent phases + the last word)
__device__ void mt_twist(unsigned* s) {
    const int T = blockDim.x, tid = threadIdx.x;
    const int lo[3] = {0, 227, 454}, hi[3] = {227, 454, 623};
    for (int ph = 0; ph < 3; ph++) {
        for (int base = lo[ph]; base < hi[ph]; base += T) {
            const int i = base + tid; unsigned nv = 0; const bool on = i < hi[ph];
            if (on) nv = s[(ph == 0) ? i + 397 : i - 227] ^ mt_mix(s[i], s[i + 1]);
            __syncthreads();
            if (on) s[i] = nv;
            __syncthreads();
        }
    }
    unsigned last = 0;
    if (tid == 0) last = s[396] ^ mt_mix(s[623], s[0]);
    __syncthreads();
    if (tid == 0) s[623] = last;
    __syncthreads();
}
__device__ __forceinline__ unsigned mt_next(unsigned* s, int& pos) {
    if (pos >= 624) { mt_twist(s); pos = 0; }
    return mt_temper(s[pos++]);
}
// Lemire's nearly divisionless method as libstdc++ instantiates it for a 32-bit engine (bits/uniform_int_dist.h: _S_nd<uint64_t>)
__device__ __forceinline__ bool lemire_accept(unsigned raw, unsigned range, unsigned* out) {
    const unsigned long long prod = (unsigned long long)raw * (unsigned long long)range;
    const unsigned low = (unsigned)prod;
    *out = (unsigned)(prod >> 32);
    if (low < range) { const unsigned thr = (0u - range) % range; if (low < thr) return false; }
    return true;
}
__device__ __forceinline__ int mt_uniform_int(unsigned* s, int& pos, int a, int b) {       // uniform_int_distribution<int>(a, b)(rng)
    const unsigned range = (unsigned)(b - a) + 1u;
    unsigned r;
    while (!lemire_accept(mt_next(s, pos), range, &r)) {}
    return a + (int)r;
}
// host side: state after seed(s) (the first draw twists it)
inline void mt_seed_host(unsigned seed, unsigned* s) { s[0] = seed; for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (unsigned)i; }

// utils::RandomShuffleAndResize (utils.h:48-73): Fisher-Yates over list[0..m) with the LO stream, only the first `keep` entries are
// used afterwards.  The draws of the discarded tail still advance the generator: with fast = true they are checked for Lemire
// rejections in parallel (a rejection, probability ~ m / 2^32, falls back to the sequential walk from that draw on).
// s_dr: LDS int[>= keep]; s_flag: LDS int.
__device__ void block_shuffle_resize(int* list, int m, int keep, unsigned* s, int& pos, bool fast, int* s_dr, int* s_flag) {
    if (m < 2) return;
    const int nd = m - 1;                               // draws i = 0 .. m-2, draw i is uniform in [i, m-1]
    const int head = (keep < nd) ? keep : nd;
    for (int i = 0; i < head; i++) { const int idx = mt_uniform_int(s, pos, i, m - 1); if (threadIdx.x == 0) s_dr[i] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < head; i++) { const int t = list[i]; list[i] = list[s_dr[i]]; list[s_dr[i]] = t; }
    __syncthreads();
    int i = head;
    while (i < nd) {
        if (!fast) { (void)mt_uniform_int(s, pos, i, m - 1); i++; continue; }
        if (pos >= 624) { mt_twist(s); pos = 0; }
        const int seg = min(nd - i, 624 - pos);
        // first draw of the segment whose word is rejected (none: seg)
        if (threadIdx.x == 0) *s_flag = seg;
        __syncthreads();
        int first = seg;
        for (int j = threadIdx.x; j < seg; j += blockDim.x) {
            unsigned r; if (!lemire_accept(mt_temper(s[pos + j]), (unsigned)(m - (i + j)), &r)) { first = j; break; }
        }
        if (first < seg) atomicMin(s_flag, first);
        __syncthreads();
        first = *s_flag;
        __syncthreads();
        pos += first; i += first;
        if (first < seg) { (void)mt_uniform_int(s, pos, i, m - 1); i++; }       // the rejected draw, sequentially (consumes >= 2 words)
    }
}

}  // namespace ssfm
