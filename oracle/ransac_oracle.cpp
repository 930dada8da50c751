// ORACLE (test infrastructure only) -- CPU restatement of the spherical relative-pose RANSAC path.
//
//  * EvaluateModelOnPoint (Sampson)           src/spherical_estimator.cpp:67-78
//  * spherical_solver_action_matrix           src/spherical_solvers.cpp:102-311
//      rows of A (:119); nullspace basis B = last three columns of Q from a column-pivoted Householder QR of A^T
//      (:124-125); the 6x10 matrix C (:127-277) holds the cubic forms  -T01, T20, T00, T21, T12, T22  of
//      T = 2 E E^T E - tr(E E^T) E  with  E = [[p0,p1,p2],[p1,-p0,p3],[p4,p5,0]],  p = B (x,y,z)^T, over the monomials
//      [x^3, x^2y, xy^2, y^3, x^2z, xyz, y^2z, xz^2, yz^2, z^3]  (identified symbolically; built here by polynomial
//      arithmetic from that definition, not from the reference's generated expressions);  G = C[:, :6]^-1 C[:, 6:] (:279);
//      4x4 action matrix for multiplication by x on the basis [y^2, x, y, 1] (:281-285); solutions = rows 1..3 of its
//      eigenvectors (:296), E normalised to unit Frobenius norm (:305).  The reference keeps the REAL PART of complex
//      eigenvectors as returned by Eigen::EigenSolver; such candidates are never valid models, and the oracle gives
//      them the deterministic stand-in Re(v) with v scaled so that its last entry is 1.
//  * NonMinimalSolver / LeastSquares / Decompose   src/spherical_estimator.cpp:86-164  (SampsonError :23-65)
//  * make / decompose_spherical_essential_matrix   src/spherical_utils.cpp:9-66
//  * LocallyOptimizedMSAC::EstimateModel, LocalOptimization, LeastSquaresFit, GetInliers   include/RansacLib/ransac.h:128-420
//  * UniformSampling (std::mt19937 + std::uniform_int_distribution, libstdc++)          include/RansacLib/sampling.h:46-135
//  * NumRequiredIterations, RandomShuffleAndResize                                      include/RansacLib/utils.h:48-140
//  * estimate_pairwise per-pair logic                examples/spherical_sfm_tools.cpp:309-431
// PARITY PARTLY PINNED (ssfm_oracle.h): LO-MSAC control flow, SolveQuartic, the constraint matrices and both solver back ends against the reference itself; LeastSquares / Decompose / the QR basis restated.  3x3 matrices cross the C API column-major.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
#include <array>
#include "lm.hpp"
#ifdef SSFM_ORACLE_REAL_RANSACLIB      // oracle/_ref build: the reference's own include/RansacLib drives the same estimators (lomsac_reference.hpp)
#include "lomsac_reference.hpp"
#define ORACLE_LOMSAC LoMsacReference
#else
#include "lomsac.hpp"
#define ORACLE_LOMSAC LoMsac
#endif
#include "rotation.hpp"
#include "ssfm_oracle.h"

namespace oracle {

struct Rays { int n; const double* u; const double* v; };   // [n*3] each

// E row-major inside this file
static inline double sampson(const double* E, const double* u, const double* v) {
    const double Eu[3] = {E[0] * u[0] + E[1] * u[1] + E[2] * u[2], E[3] * u[0] + E[4] * u[1] + E[5] * u[2], E[6] * u[0] + E[7] * u[1] + E[8] * u[2]};
    const double Etv[2] = {E[0] * v[0] + E[3] * v[1] + E[6] * v[2], E[1] * v[0] + E[4] * v[1] + E[7] * v[2]};
    const double d = v[0] * Eu[0] + v[1] * Eu[1] + v[2] * Eu[2];
    return (d * d) / (Eu[0] * Eu[0] + Eu[1] * Eu[1] + Etv[0] * Etv[0] + Etv[1] * Etv[1]);
}

// ---- small polynomial algebra in (x,y,z): linear (3), quadratic (6: xx xy xz yy yz zz), cubic (10, order above)
struct Lin { double c[3]; };
struct Quad { double c[6]; };
struct Cub { double c[10]; };
static inline Quad mul(const Lin& a, const Lin& b) {
    Quad q; q.c[0] = a.c[0] * b.c[0]; q.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0]; q.c[2] = a.c[0] * b.c[2] + a.c[2] * b.c[0];
    q.c[3] = a.c[1] * b.c[1]; q.c[4] = a.c[1] * b.c[2] + a.c[2] * b.c[1]; q.c[5] = a.c[2] * b.c[2]; return q; }
static inline Quad add(const Quad& a, const Quad& b) { Quad q; for (int i = 0; i < 6; i++) q.c[i] = a.c[i] + b.c[i]; return q; }
// cubic monomials: 0 x3, 1 x2y, 2 xy2, 3 y3, 4 x2z, 5 xyz, 6 y2z, 7 xz2, 8 yz2, 9 z3
static inline void acc(Cub& r, const Quad& q, const Lin& l, double s) {
    const double* a = q.c; const double* b = l.c;
    r.c[0] += s * (a[0] * b[0]);
    r.c[1] += s * (a[0] * b[1] + a[1] * b[0]);
    r.c[2] += s * (a[1] * b[1] + a[3] * b[0]);
    r.c[3] += s * (a[3] * b[1]);
    r.c[4] += s * (a[0] * b[2] + a[2] * b[0]);
    r.c[5] += s * (a[1] * b[2] + a[2] * b[1] + a[4] * b[0]);
    r.c[6] += s * (a[3] * b[2] + a[4] * b[1]);
    r.c[7] += s * (a[2] * b[2] + a[5] * b[0]);
    r.c[8] += s * (a[4] * b[2] + a[5] * b[1]);
    r.c[9] += s * (a[5] * b[2]);
}

// Householder QR with column pivoting of the 6 x N matrix A^T; returns Q (6x6, row-major)
static void qr_colpiv_Q(std::vector<double> At, int N, double Q[36]) {      // At row-major 6 x N, modified
    const int rows = 6, steps = std::min(rows, N);
    std::vector<std::vector<double>> vs; std::vector<double> taus;
    std::vector<double> colnorm(N);
    for (int j = 0; j < N; j++) { double s = 0; for (int i = 0; i < rows; i++) s += At[i * N + j] * At[i * N + j]; colnorm[j] = s; }
    for (int k = 0; k < steps; k++) {
        int piv = k; double best = -1;
        for (int j = k; j < N; j++) { double s = 0; for (int i = k; i < rows; i++) s += At[i * N + j] * At[i * N + j]; colnorm[j] = s; if (s > best) { best = s; piv = j; } }
        if (piv != k) for (int i = 0; i < rows; i++) std::swap(At[i * N + k], At[i * N + piv]);
        std::vector<double> v(rows, 0.0);
        double alpha = 0; for (int i = k; i < rows; i++) alpha += At[i * N + k] * At[i * N + k];
        alpha = std::sqrt(alpha);
        if (alpha == 0.0) { vs.push_back(v); taus.push_back(0.0); continue; }
        const double x0 = At[k * N + k];
        const double beta = (x0 >= 0) ? -alpha : alpha;
        for (int i = k; i < rows; i++) v[i] = At[i * N + k];
        v[k] = x0 - beta;
        double vn = 0; for (int i = k; i < rows; i++) vn += v[i] * v[i];
        const double tau = (vn > 0) ? 2.0 / vn : 0.0;
        for (int j = k; j < N; j++) { double d = 0; for (int i = k; i < rows; i++) d += v[i] * At[i * N + j]; d *= tau; for (int i = k; i < rows; i++) At[i * N + j] -= d * v[i]; }
        vs.push_back(v); taus.push_back(tau);
    }
    for (int i = 0; i < 36; i++) Q[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int k = (int)vs.size() - 1; k >= 0; k--) {       // Q = H0 H1 ... applied to I from the right-most
        const auto& v = vs[k]; const double tau = taus[k];
        for (int j = 0; j < 6; j++) { double d = 0; for (int i = 0; i < 6; i++) d += v[i] * Q[i * 6 + j]; d *= tau; for (int i = 0; i < 6; i++) Q[i * 6 + j] -= d * v[i]; }
    }
}

static bool lu_solve6(double A[36], double Bm[24]) {      // A 6x6 row-major, B 6x4; partial pivoting (Eigen .lu())
    for (int k = 0; k < 6; k++) {
        int p = k; for (int i = k + 1; i < 6; i++) if (std::fabs(A[i * 6 + k]) > std::fabs(A[p * 6 + k])) p = i;
        if (A[p * 6 + k] == 0.0) return false;
        if (p != k) { for (int j = 0; j < 6; j++) std::swap(A[k * 6 + j], A[p * 6 + j]); for (int j = 0; j < 4; j++) std::swap(Bm[k * 4 + j], Bm[p * 4 + j]); }
        for (int i = k + 1; i < 6; i++) {
            const double f = A[i * 6 + k] / A[k * 6 + k];
            for (int j = k; j < 6; j++) A[i * 6 + j] -= f * A[k * 6 + j];
            for (int j = 0; j < 4; j++) Bm[i * 4 + j] -= f * Bm[k * 4 + j];
        }
    }
    for (int k = 5; k >= 0; k--) for (int j = 0; j < 4; j++) {
        double s = Bm[k * 4 + j]; for (int i = k + 1; i < 6; i++) s -= A[k * 6 + i] * Bm[i * 4 + j];
        Bm[k * 4 + j] = s / A[k * 6 + k];
    }
    return true;
}

typedef std::complex<double> cd;
static double g_dk_tol = std::getenv("ORACLE_DK_TOL") ? std::atof(std::getenv("ORACLE_DK_TOL")) : 1e-13;
static long long g_dk_hist[201];
static long long g_lsq_calls = 0, g_lsq_iterations = 0, g_lsq_points = 0;      // diagnostics: oracle_lsq_counters      // iterations of the root finder per call (diagnostics: oracle_dk_histogram)
// eigenvalues of a real 4x4: shifted QR on the Hessenberg form would be the textbook way; for a 4x4 the characteristic
// polynomial (Faddeev-LeVerrier) + Durand-Kerner/Newton polishing is exact enough and short.
static void eig4(const double M[16], cd lam[4]) {
    double c[5]; // lambda^4 + c3 l^3 + c2 l^2 + c1 l + c0
    double Mk[16], I[16]; for (int i = 0; i < 16; i++) { I[i] = (i % 5 == 0); Mk[i] = M[i]; }
    double cc[4];
    double Bk[16]; std::memcpy(Bk, I, sizeof(I));   // B1 = I
    for (int k = 1; k <= 4; k++) {
        double AB[16]; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int t = 0; t < 4; t++) s += M[i * 4 + t] * Bk[t * 4 + j]; AB[i * 4 + j] = s; }
        double tr = AB[0] + AB[5] + AB[10] + AB[15];
        cc[k - 1] = -tr / k;
        for (int i = 0; i < 16; i++) Bk[i] = AB[i] + cc[k - 1] * I[i];
    }
    c[4] = 1; c[3] = cc[0]; c[2] = cc[1]; c[1] = cc[2]; c[0] = cc[3];
    (void)Mk;
    cd z[4] = {cd(0.4, 0.9), cd(-0.9, 0.4), cd(-0.4, -0.9), cd(0.9, -0.4)};
    double scale = 1.0 + std::fabs(c[3]) + std::sqrt(std::fabs(c[2])) + std::cbrt(std::fabs(c[1])) + std::sqrt(std::sqrt(std::fabs(c[0])));
    for (auto& r : z) r *= scale;
    auto P = [&](cd x) { return (((x + c[3]) * x + c[2]) * x + c[1]) * x + c[0]; };
    // Simple roots converge quadratically: once the largest correction is below 1e-13 (relative to the root bound) ONE more sweep puts every root
    // at rounding level.  (A test at 1e-15 is met by rounding noise only now and then: 2 % of the action matrices of noisy minimal samples ran
    // all 200 sweeps for the same roots.)
    int used = 200; bool last = false;
    for (int it = 0; it < 200; it++) {
        double change = 0;
        for (int i = 0; i < 4; i++) {
            cd den = 1; for (int j = 0; j < 4; j++) if (j != i) den *= (z[i] - z[j]);
            if (std::abs(den) == 0) den = 1e-300;
            const cd dz = P(z[i]) / den; z[i] -= dz; change = std::max(change, std::abs(dz));
        }
        if (last) { used = it + 1; break; }
        if (change < g_dk_tol * scale) last = true;
    }
    g_dk_hist[std::min(used, 200)]++;
    for (int i = 0; i < 4; i++) lam[i] = z[i];
}

// Common front end of both minimal solvers (src/spherical_solvers.cpp:113-125 / 324-337 and the generated constraint matrix):
// nullspace basis B (6x3) and the six cubic constraints -T01, T20, T00, T21, T12, T22 of T = 2 E E^T E - tr(E E^T) E over the
// monomials [x^3, x^2y, xy^2, y^3, x^2z, xyz, y^2z, xz^2, yz^2, z^3]  (identified symbolically from the generated code; the
// polynomial variant holds the same rows times 1/2 in a different monomial order).
// the six cubic constraints from a GIVEN nullspace basis (everything of src/spherical_solvers.cpp:127-277 / 339-621 is a function of B alone)
static void constraints_from_B(const double B[6][3], Cub rows[6]) {
    Lin p[6]; for (int k = 0; k < 6; k++) for (int j = 0; j < 3; j++) p[k].c[j] = B[k][j];
    Lin zero = {{0, 0, 0}}, np0 = {{-p[0].c[0], -p[0].c[1], -p[0].c[2]}};
    const Lin* Em[3][3] = {{&p[0], &p[1], &p[2]}, {&p[1], &np0, &p[3]}, {&p[4], &p[5], &zero}};
    Quad EEt[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Quad q = mul(*Em[i][0], *Em[j][0]); q = add(q, mul(*Em[i][1], *Em[j][1])); q = add(q, mul(*Em[i][2], *Em[j][2])); EEt[i][j] = q; }
    Quad tr = add(add(EEt[0][0], EEt[1][1]), EEt[2][2]);
    auto T = [&](int i, int j, double s) { Cub c; for (double& x : c.c) x = 0; for (int k = 0; k < 3; k++) acc(c, EEt[i][k], *Em[k][j], 2.0 * s); acc(c, tr, *Em[i][j], -s); return c; };
    rows[0] = T(0, 1, -1.0); rows[1] = T(2, 0, 1.0); rows[2] = T(0, 0, 1.0); rows[3] = T(2, 1, 1.0); rows[4] = T(1, 2, 1.0); rows[5] = T(2, 2, 1.0);
}
static bool solver_front(const Rays& R, const int* sample, int N, double B[6][3], Cub rows[6]) {
    if (N < 3) return false;
    std::vector<double> At((size_t)6 * N);
    for (int i = 0; i < N; i++) {
        const double* u = R.u + 3 * sample[i]; const double* v = R.v + 3 * sample[i];
        const double row[6] = {u[0] * v[0] - u[1] * v[1], u[0] * v[1] + u[1] * v[0], u[2] * v[0], u[2] * v[1], u[0] * v[2], u[1] * v[2]};
        for (int k = 0; k < 6; k++) At[(size_t)k * N + i] = row[k];
    }
    double Q[36]; qr_colpiv_Q(At, N, Q);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 3; j++) B[i][j] = Q[i * 6 + 3 + j];
    constraints_from_B(B, rows);
    return true;
}
static void essential_from_b(const double B[6][3], const double b[3], double* E) {      // src/spherical_solvers.cpp:296-305 / 645-654
    double ps[6]; for (int k = 0; k < 6; k++) ps[k] = B[k][0] * b[0] + B[k][1] * b[1] + B[k][2] * b[2];
    E[0] = ps[0]; E[1] = ps[1]; E[2] = ps[2]; E[3] = ps[1]; E[4] = -ps[0]; E[5] = ps[3]; E[6] = ps[4]; E[7] = ps[5]; E[8] = 0.0;
    double nrm = 0; for (int k = 0; k < 9; k++) nrm += E[k] * E[k]; nrm = std::sqrt(nrm);
    for (int k = 0; k < 9; k++) E[k] /= nrm;
}

// src/spherical_solvers.cpp:102-311.  sample: indices into rays.  Es: 4 x 9 row-major.  Returns number of models.
static int action_matrix_back(const double B[6][3], const Cub rows[6], double Es[36]);
static int solver_action_matrix(const Rays& R, const int* sample, int N, double Es[36]) {
    double B[6][3]; Cub rows[6];
    if (!solver_front(R, sample, N, B, rows)) return 0;
    return action_matrix_back(B, rows, Es);
}
static int action_matrix_back(const double B[6][3], const Cub rows[6], double Es[36]) {
    double C1[36], C2[24];
    for (int r = 0; r < 6; r++) { for (int k = 0; k < 6; k++) C1[r * 6 + k] = rows[r].c[k]; for (int k = 0; k < 4; k++) C2[r * 4 + k] = rows[r].c[6 + k]; }
    if (!lu_solve6(C1, C2)) return 0;
    double M[16] = {0};
    for (int k = 0; k < 4; k++) { M[0 * 4 + k] = -C2[2 * 4 + k]; M[1 * 4 + k] = -C2[4 * 4 + k]; M[2 * 4 + k] = -C2[5 * 4 + k]; }
    M[3 * 4 + 1] = 1.0;
    cd lam[4]; eig4(M, lam);
    for (int s = 0; s < 4; s++) {
        // eigenvector with last entry 1: rows 1..3 of (M - lam I) v = 0 give v1 = lam (row 3) and a 2x2 system for v0, v2
        const cd l = lam[s];
        const cd v1 = l;                                   // row 3: v1 - l v3 = 0, v3 = 1
        // rows 1 and 2: M10 v0 + (M11 - l) v1 + M12 v2 + M13 = 0 ; M20 v0 + M21 v1 + (M22 - l) v2 + M23 = 0
        const cd a11 = M[4], a12 = M[6], b1 = -((M[5] - l) * v1 + M[7]);
        const cd a21 = M[8], a22 = M[10] - l, b2 = -(M[9] * v1 + M[11]);
        const cd det = a11 * a22 - a12 * a21;
        cd v0, v2;
        if (std::abs(det) > 0) { v0 = (b1 * a22 - a12 * b2) / det; v2 = (a11 * b2 - b1 * a21) / det; } else { v0 = 0; v2 = 0; }
        (void)v0;
        const double b[3] = {v1.real(), v2.real(), 1.0};
        essential_from_b(B, b, Es + 9 * s);
    }
    return 4;
}

// SolveQuartic (Ferrari, "from Theia library"), src/spherical_solvers.cpp:15-69
static void solve_quartic(double a, double b, double c, double d, double e, cd roots[4]) {
    const double a2 = a * a, b2 = b * b, a3 = a2 * a, b3 = b2 * b, a4 = a3 * a, b4 = b3 * b;
    const double alpha = -3.0 * b2 / (8.0 * a2) + c / a;
    const double beta = b3 / (8.0 * a3) - b * c / (2.0 * a2) + d / a;
    const double gamma = -3.0 * b4 / (256.0 * a4) + b2 * c / (16.0 * a3) - b * d / (4.0 * a2) + e / a;
    const double alpha2 = alpha * alpha, alpha3 = alpha2 * alpha;
    const cd P(-alpha2 / 12.0 - gamma, 0);
    const cd Q(-alpha3 / 108.0 + alpha * gamma / 3.0 - std::pow(beta, 2.0) / 8.0, 0);
    const cd Rr = -Q / 2.0 + std::sqrt(std::pow(Q, 2.0) / 4.0 + std::pow(P, 3.0) / 27.0);
    const cd U = std::pow(Rr, (1.0 / 3.0));
    cd y;
    if (std::abs(U.real()) < 1e-8) y = -5.0 * alpha / 6.0 - std::pow(Q, (1.0 / 3.0));
    else y = -5.0 * alpha / 6.0 - P / (3.0 * U) + U;
    const cd w = std::sqrt(alpha + 2.0 * y);
    roots[0] = -b / (4.0 * a) + 0.5 * (w + std::sqrt(-(3.0 * alpha + 2.0 * y + 2.0 * beta / w)));
    roots[1] = -b / (4.0 * a) + 0.5 * (w - std::sqrt(-(3.0 * alpha + 2.0 * y + 2.0 * beta / w)));
    roots[2] = -b / (4.0 * a) + 0.5 * (-w + std::sqrt(-(3.0 * alpha + 2.0 * y - 2.0 * beta / w)));
    roots[3] = -b / (4.0 * a) + 0.5 * (-w - std::sqrt(-(3.0 * alpha + 2.0 * y - 2.0 * beta / w)));
}

// src/spherical_solvers.cpp:313-660: same constraints (times 1/2), monomials [x^3 x^2y xy^2 x^2z xyz xz^2 | y^3 y^2z yz^2 z^3];
// rows 4, 5 of G = C[:, :6]^-1 C[:, 6:] give xy and x as cubics in y (z = 1) => quartic in y; the REAL PARTS of all four roots
// are used (SolveQuarticReals without tolerance, :629-631).  imag_out (optional): imaginary parts, for the tests.
static const int kPolyPerm[10] = {0, 1, 2, 4, 5, 7, 3, 6, 8, 9};
static int polynomial_back(const double B[6][3], const Cub rows[6], double Es[36], double* imag_out, double* abcde_out);
static int solver_polynomial(const Rays& R, const int* sample, int N, double Es[36], double* imag_out = nullptr) {
    double B[6][3]; Cub rows[6];
    if (!solver_front(R, sample, N, B, rows)) return 0;
    return polynomial_back(B, rows, Es, imag_out, nullptr);
}
static int polynomial_back(const double B[6][3], const Cub rows[6], double Es[36], double* imag_out, double* abcde_out) {
    const int* perm = kPolyPerm;
    double C1[36], C2[24];
    for (int r = 0; r < 6; r++) { for (int k = 0; k < 6; k++) C1[r * 6 + k] = 0.5 * rows[r].c[perm[k]]; for (int k = 0; k < 4; k++) C2[r * 4 + k] = 0.5 * rows[r].c[perm[6 + k]]; }
    if (!lu_solve6(C1, C2)) return 0;
    const double* G4 = C2 + 16; const double* G5 = C2 + 20;
    const double a = -G5[0], b = G4[0] - G5[1], c = G4[1] - G5[2], d = G4[2] - G5[3], e = G4[3];
    if (abcde_out) { abcde_out[0] = a; abcde_out[1] = b; abcde_out[2] = c; abcde_out[3] = d; abcde_out[4] = e; }
    cd roots[4]; solve_quartic(a, b, c, d, e, roots);
    for (int s = 0; s < 4; s++) {
        const double y = roots[s].real(), y2 = y * y, y3 = y2 * y;
        const double x = -G5[0] * y3 - G5[1] * y2 - G5[2] * y - G5[3];
        const double bs[3] = {x, y, 1.0};
        essential_from_b(B, bs, Es + 9 * s);
        if (imag_out) imag_out[s] = roots[s].imag();
    }
    return 4;
}

// ---- src/spherical_utils.cpp:9-66 (row-major 3x3 here)
static void make_E(const double* Rm, bool inward, double* E) {
    double t[3] = {Rm[2], Rm[5], Rm[8] - 1.0};
    if (inward) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
    const double S[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) E[3 * i + j] = S[3 * i] * Rm[j] + S[3 * i + 1] * Rm[3 + j] + S[3 * i + 2] * Rm[6 + j];
}
static void sym_eig3(const double A[9], double w[3], double V[9]) {       // Jacobi, eigenvalues descending
    double a[9]; std::memcpy(a, A, 72); for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5]; if (off < 1e-300) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (a[3 * p + q] == 0.0) continue;
            const double th = (a[3 * q + q] - a[3 * p + p]) / (2 * a[3 * p + q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0)), c = 1 / std::sqrt(t * t + 1), s = t * c;
            for (int k = 0; k < 3; k++) { const double akp = a[3 * k + p], akq = a[3 * k + q]; a[3 * k + p] = c * akp - s * akq; a[3 * k + q] = s * akp + c * akq; }
            for (int k = 0; k < 3; k++) { const double apk = a[3 * p + k], aqk = a[3 * q + k]; a[3 * p + k] = c * apk - s * aqk; a[3 * q + k] = s * apk + c * aqk; }
            for (int k = 0; k < 3; k++) { const double vkp = V[3 * k + p], vkq = V[3 * k + q]; V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq; }
        }
    }
    int idx[3] = {0, 1, 2}; double d[3] = {a[0], a[4], a[8]};
    std::sort(idx, idx + 3, [&](int x, int y) { return d[x] > d[y]; });
    double Vs[9]; for (int k = 0; k < 3; k++) { w[k] = d[idx[k]]; for (int i = 0; i < 3; i++) Vs[3 * i + k] = V[3 * i + idx[k]]; }
    std::memcpy(V, Vs, 72);
}
static double det3(const double* M) { return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]); }
static void svd3(const double* E, double U[9], double V[9]) {
    double EtE[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) EtE[3 * i + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
    double w[3]; sym_eig3(EtE, w, V);
    double u[3][3];
    for (int k = 0; k < 2; k++) { for (int i = 0; i < 3; i++) u[k][i] = E[3 * i] * V[k] + E[3 * i + 1] * V[3 + k] + E[3 * i + 2] * V[6 + k];
                                  const double n = std::sqrt(u[k][0] * u[k][0] + u[k][1] * u[k][1] + u[k][2] * u[k][2]); for (int i = 0; i < 3; i++) u[k][i] /= n; }
    // re-orthogonalise u1 against u0, third = u0 x u1
    double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2]; for (int i = 0; i < 3; i++) u[1][i] -= d * u[0][i];
    double n = std::sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]); for (int i = 0; i < 3; i++) u[1][i] /= n;
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    for (int k = 0; k < 3; k++) for (int i = 0; i < 3; i++) U[3 * i + k] = u[k][i];
}
static void rm_so3ln(const double* Rm, double* r) { double cm[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cm[i + 3 * j] = Rm[3 * i + j]; so3ln(cm, r); }
static void rm_so3exp(const double* r, double* Rm) { double cm[9]; so3exp(r, cm); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rm[3 * i + j] = cm[i + 3 * j]; }
static void decompose_E(const double* E, bool inward, double r[3], double t[3]) {
    double U[9], V[9]; svd3(E, U, V);
    if (det3(U) < 0) for (double& x : U) x = -x;
    if (det3(V) < 0) for (double& x : V) x = -x;
    const double D[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, DT[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    auto mul3 = [](const double* A, const double* B, double* C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j]; };
    double VT[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) VT[3 * i + j] = V[3 * j + i];
    double UD[9], R1[9], R2[9]; mul3(U, D, UD); mul3(UD, VT, R1); mul3(U, DT, UD); mul3(UD, VT, R2);
    const double tu[3] = {U[2], U[5], U[8]};
    double t1[3] = {R1[2], R1[5], R1[8] - 1}, t2[3] = {R2[2], R2[5], R2[8] - 1};
    if (inward) for (int k = 0; k < 3; k++) { t1[k] = -t1[k]; t2[k] = -t2[k]; }
    const double n1 = std::sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]), n2 = std::sqrt(t2[0] * t2[0] + t2[1] * t2[1] + t2[2] * t2[2]);
    const double s1 = std::fabs((t1[0] * tu[0] + t1[1] * tu[1] + t1[2] * tu[2]) / n1), s2 = std::fabs((t2[0] * tu[0] + t2[1] * tu[1] + t2[2] * tu[2]) / n2);
    if (s1 > s2) { rm_so3ln(R1, r); std::memcpy(t, t1, 24); } else { rm_so3ln(R2, r); std::memcpy(t, t2, 24); }
}

// ---- LeastSquares (src/spherical_estimator.cpp:110-157): Ceres on SampsonError residuals (:23-65).
// Parameter blocks of the problem are r0, t0, r1, t1, u[i], v[i]; the source sets u[i], v[i] (:140-141), r0 (:143) and t0 (:144)
// constant and NOTHING ELSE: t1 (declared :118-119, initialised to t0) stays a free 3-vector.  Ceres therefore minimises over the six
// parameters x = [r1; t1] (reduced program order = order of first appearance: r1 before t1), and the caller throws t1 away afterwards
// (:156 rebuilds E from so3exp(r1) alone).  Rounds 1-2 of this oracle fitted r1 only, following SURVEY a12's sentence instead of the
// source; the converged rotations of the two problems differ by 3e-5..7e-5 rad at 1/1000 noise (tests/test_ransac_lsq6_*).  The
// 3-parameter fit is kept below as `r_only` for exactly that negative test.
template <typename T>
static void sampson_residual(const T rj[3], const T tj[3], bool inward, const double* u, const double* v, T* res) {
    // :33-44 with ri = 0 (Ri = I: AngleAxisToRotationMatrix takes its first-order branch, I + [0]x), ti = (0, 0, tz):
    // R = Rj Ri^T = Rj,  t = Rj (-Ri^T ti) + tj = -Rj ti + tj
    T Rj[9]; AngleAxisToRotationMatrix(rj, Rj);   // column-major
    const double tz = inward ? 1.0 : -1.0;
    T t[3] = {Rj[6] * (-tz) + tj[0], Rj[7] * (-tz) + tj[1], Rj[8] * (-tz) + tj[2]};
    // E = [t]x R, R(i,j) = Rj[i + 3j]   (:46-51)
    T E[9];
    for (int j = 0; j < 3; j++) {
        const T c0 = Rj[0 + 3 * j], c1 = Rj[1 + 3 * j], c2 = Rj[2 + 3 * j];
        E[0 * 3 + j] = t[1] * c2 - t[2] * c1;      // row 0 of [t]x = (0, -t2, t1)
        E[1 * 3 + j] = t[2] * c0 - t[0] * c2;      // row 1 = (t2, 0, -t0)
        E[2 * 3 + j] = t[0] * c1 - t[1] * c0;      // row 2 = (-t1, t0, 0)
    }
    const T Eu[3] = {E[0] * u[0] + E[1] * u[1] + E[2] * u[2], E[3] * u[0] + E[4] * u[1] + E[5] * u[2], E[6] * u[0] + E[7] * u[1] + E[8] * u[2]};
    const T Etv0 = E[0] * v[0] + E[3] * v[1] + E[6] * v[2], Etv1 = E[1] * v[0] + E[4] * v[1] + E[7] * v[2];
    const T d = Eu[0] * v[0] + Eu[1] * v[1] + Eu[2] * v[2];
    *res = (d * d) / (Eu[0] * Eu[0] + Eu[1] * Eu[1] + Etv0 * Etv0 + Etv1 * Etv1);      // :59-60
}
// NP = 6: x = [r1; t1] as the reference runs it.  NP = 3: x = r1 with t1 pinned to t0 (NOT the reference; negative test only).
template <int NP>
struct SampsonLSQ : LMProblem {
    const Rays& R; const std::vector<int>& idx; bool inward;
    std::vector<double> res, J;   // per residual: value, NP partials
    SampsonLSQ(const Rays& r, const std::vector<int>& i, bool in) : R(r), idx(i), inward(in), res(i.size()), J(NP * i.size()) {}
    int num_parameters() const override { return NP; }
    template <typename T> void eval(const T* x, int k, T* r) const {
        if (NP == 6) sampson_residual<T>(x, x + 3, inward, R.u + 3 * k, R.v + 3 * k, r);
        else { const T t1[3] = {T(0.0), T(0.0), T(inward ? 1.0 : -1.0)}; sampson_residual<T>(x, t1, inward, R.u + 3 * k, R.v + 3 * k, r); }
    }
    bool cost_only(const double* x, double* cost) override {
        double c = 0; for (int k : idx) { double r; eval<double>(x, k, &r); c += 0.5 * r * r; }
        *cost = c; return std::isfinite(c);
    }
    bool linearize(const double* x, double* cost, double* g) override {
        typedef Jet<NP> JN; double c = 0; for (int k = 0; k < NP; k++) g[k] = 0;
        for (size_t q = 0; q < idx.size(); q++) {
            JN xs[NP], r; for (int k = 0; k < NP; k++) xs[k] = JN(x[k], k);
            eval<JN>(xs, idx[q], &r);
            res[q] = r.a; for (int k = 0; k < NP; k++) { J[NP * q + k] = r.v[k]; g[k] += r.v[k] * r.a; }
            c += 0.5 * r.a * r.a;
        }
        *cost = c; return std::isfinite(c);
    }
    void squared_column_norms(const double* s, double* out) override {
        for (int k = 0; k < NP; k++) out[k] = 0;
        for (size_t q = 0; q < idx.size(); q++) for (int k = 0; k < NP; k++) out[k] += J[NP * q + k] * J[NP * q + k];
        if (s) for (int k = 0; k < NP; k++) out[k] *= s[k] * s[k];
    }
    // DENSE_NORMAL_CHOLESKY (:147): lhs = Js^T Js + D^2, LL^T; a non-positive pivot is a failed linear solve (Eigen::LLT NumericalIssue)
    bool solve(const double* s, const double* D, double* y) override {
        double A[NP * NP] = {0}, b[NP] = {0};
        for (size_t q = 0; q < idx.size(); q++) for (int a = 0; a < NP; a++) { const double ja = J[NP * q + a] * s[a]; b[a] += ja * res[q]; for (int c = 0; c < NP; c++) A[NP * a + c] += ja * J[NP * q + c] * s[c]; }
        for (int a = 0; a < NP; a++) A[(NP + 1) * a] += D[a] * D[a];
        double L[NP * NP] = {0};
        for (int j = 0; j < NP; j++) { double d = A[(NP + 1) * j]; for (int k = 0; k < j; k++) d -= L[NP * j + k] * L[NP * j + k]; if (!(d > 0)) return false; L[(NP + 1) * j] = std::sqrt(d);
            for (int i = j + 1; i < NP; i++) { double v = A[NP * i + j]; for (int k = 0; k < j; k++) v -= L[NP * i + k] * L[NP * j + k]; L[NP * i + j] = v / L[(NP + 1) * j]; } }
        double z[NP]; for (int i = 0; i < NP; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[NP * i + k] * z[k]; z[i] = v / L[(NP + 1) * i]; }
        for (int i = NP - 1; i >= 0; i--) { double v = z[i]; for (int k = i + 1; k < NP; k++) v -= L[NP * k + i] * y[k]; y[i] = v / L[(NP + 1) * i]; }
        return true;
    }
    double model_cost_change(const double* s, const double* step) override {
        double a = 0; for (size_t q = 0; q < idx.size(); q++) { double m = 0; for (int k = 0; k < NP; k++) m += J[NP * q + k] * s[k] * step[k]; a += m * (res[q] + 0.5 * m); } return -a;
    }
    void plus(const double* x, const double* d, double* o) override { for (int k = 0; k < NP; k++) o[k] = x[k] + d[k]; }
};
static bool g_lsq_r_only = false;          // negative test only: the 3-parameter fit of rounds 1-2
// diagnostics (tests log WHERE a device trace leaves the oracle's): every LeastSquares call of a run, in order
struct LsqRecord { std::vector<int> sample; double E_in[9], E_out[9], x[6]; int iterations, termination; };
static bool g_lsq_log_on = false; static std::vector<LsqRecord> g_lsq_log;
struct NonMinRecord { std::vector<int> sample; double E_out[9]; int ok; int after_lsq_calls; };      // NonMinimalSolver calls of the same run
static std::vector<NonMinRecord> g_nonmin_log;
static void least_squares(const Rays& R, bool inward, const std::vector<int>& sample, double* E, LMSummary* out_sm = nullptr, double* out_x = nullptr) {
    LsqRecord rec; if (g_lsq_log_on) { rec.sample = sample; std::memcpy(rec.E_in, E, 72); }
    double x[6], t[3]; decompose_E(E, inward, x, t);                                     // :115-117 r1 = r
    x[3] = 0; x[4] = 0; x[5] = inward ? 1.0 : -1.0;                                      // :118-119 t1 = (0,0,-1) / (0,0,1)
    LMOptions o; o.max_num_iterations = 200; o.max_num_consecutive_invalid_steps = 10;   // :146-150
    LMSummary sm;
    if (g_lsq_r_only) { SampsonLSQ<3> P(R, sample, inward); sm = lm_minimize(P, o, x); }
    else { SampsonLSQ<6> P(R, sample, inward); sm = lm_minimize(P, o, x); }
    g_lsq_calls++; g_lsq_iterations += sm.iterations; g_lsq_points += (long long)sample.size();
    if (out_sm) *out_sm = sm;
    if (out_x) std::memcpy(out_x, x, 48);
    double Rm[9]; rm_so3exp(x, Rm); make_E(Rm, inward, E);                               // :156 t1 is discarded
    if (g_lsq_log_on) { std::memcpy(rec.E_out, E, 72); std::memcpy(rec.x, x, 48); rec.iterations = sm.iterations; rec.termination = sm.termination; g_lsq_log.push_back(rec); }
}

typedef std::array<double, 9> EMat;      // row-major
struct SphericalSolver {                 // SphericalEstimator, include/sphericalsfm/spherical_estimator.h:8-35
    const Rays& R; bool inward;
    int min_sample_size() const { return 3; }
    int non_minimal_sample_size() const { return 4; }
    int num_data() const { return R.n; }
    int MinimalSolver(const std::vector<int>& sample, std::vector<EMat>* Es) const {
        double buf[36]; const int k = solver_action_matrix(R, sample.data(), (int)sample.size(), buf);
        Es->resize(k); for (int i = 0; i < k; i++) std::memcpy((*Es)[i].data(), buf + 9 * i, 72);
        return k;
    }
    int NonMinimalSolver(const std::vector<int>& sample, EMat* E) const {            // src/spherical_estimator.cpp:86-108
        double buf[36]; if (solver_action_matrix(R, sample.data(), (int)sample.size(), buf) == 0) return 0;
        double bs = INFINITY; int bi = 0;
        for (int i = 0; i < 4; i++) { double sc = 0; for (int j : sample) sc += sampson(buf + 9 * i, R.u + 3 * j, R.v + 3 * j); if (sc < bs) { bs = sc; bi = i; } }
        std::memcpy(E->data(), buf + 9 * bi, 72);
        if (g_lsq_log_on) { NonMinRecord r; r.sample = sample; std::memcpy(r.E_out, E->data(), 72); r.ok = 1; r.after_lsq_calls = (int)g_lsq_log.size(); g_nonmin_log.push_back(r); }
        return 1;
    }
    double EvaluateModelOnPoint(const EMat& E, int i) const { return sampson(E.data(), R.u + 3 * i, R.v + 3 * i); }
    void LeastSquares(const std::vector<int>& sample, EMat* E) const { least_squares(R, inward, sample, E->data()); }
};

static void cm_to_rm(const double* cm, double* rm) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rm[3 * i + j] = cm[i + 3 * j]; }
static void rm_to_cm(const double* rm, double* cm) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cm[i + 3 * j] = rm[3 * i + j]; }

}  // namespace oracle
using namespace oracle;

extern "C" double oracle_sampson(const double E_cm[9], const double u[3], const double v[3]) { double E[9]; cm_to_rm(E_cm, E); return sampson(E, u, v); }

extern "C" int oracle_spherical_solver(int32_t n, const double* u, const double* v, int32_t ns, const int32_t* sample, double Es_cm[36]) {
    Rays R{n, u, v}; double Es[36];
    const int k = solver_action_matrix(R, sample, ns, Es);
    for (int i = 0; i < k; i++) rm_to_cm(Es + 9 * i, Es_cm + 9 * i);
    return k;
}
extern "C" int oracle_spherical_solver_poly(int32_t n, const double* u, const double* v, int32_t ns, const int32_t* sample, double Es_cm[36], double imag[4]) {
    Rays R{n, u, v}; double Es[36];
    const int k = solver_polynomial(R, sample, ns, Es, imag);
    for (int i = 0; i < k; i++) rm_to_cm(Es + 9 * i, Es_cm + 9 * i);
    return k;
}
extern "C" void oracle_make_spherical_essential_matrix(const double R_cm[9], int32_t inward, double E_cm[9]) { double Rm[9], E[9]; cm_to_rm(R_cm, Rm); make_E(Rm, inward != 0, E); rm_to_cm(E, E_cm); }
extern "C" void oracle_decompose_spherical_essential_matrix(const double E_cm[9], int32_t inward, double r[3], double t[3]) { double E[9]; cm_to_rm(E_cm, E); decompose_E(E, inward != 0, r, t); }
extern "C" void oracle_sampson_least_squares(int32_t n, const double* u, const double* v, int32_t ns, const int32_t* sample, int32_t inward, double E_cm[9]) {
    Rays R{n, u, v}; std::vector<int> s(sample, sample + ns); double E[9]; cm_to_rm(E_cm, E); least_squares(R, inward != 0, s, E); rm_to_cm(E, E_cm);
}
// the same fit with its trace exposed: x_out = [r1; t1] at the end, stats = {iterations, termination, successful, unsuccessful},
// costs = {initial, final}.  r_only != 0 runs the 3-parameter fit of rounds 1-2 (NOT the reference; the negative test of tests/).
extern "C" void oracle_sampson_least_squares_ex(int32_t n, const double* u, const double* v, int32_t ns, const int32_t* sample, int32_t inward, int32_t r_only,
                                                double E_cm[9], double x_out[6], int32_t stats[4], double costs[2]) {
    Rays R{n, u, v}; std::vector<int> s(sample, sample + ns); double E[9]; cm_to_rm(E_cm, E);
    LMSummary sm; g_lsq_r_only = r_only != 0;
    least_squares(R, inward != 0, s, E, &sm, x_out);
    g_lsq_r_only = false;
    rm_to_cm(E, E_cm);
    if (stats) { stats[0] = sm.iterations; stats[1] = sm.termination; stats[2] = sm.num_successful_steps; stats[3] = sm.num_unsuccessful_steps; }
    if (costs) { costs[0] = sm.initial_cost; costs[1] = sm.final_cost; }
}
// per-pair logic of estimate_pairwise (examples/spherical_sfm_tools.cpp:314-318,378-419): options, EstimateModel, inlier mask, R
extern "C" int oracle_ransac_pair(int32_t n, const double* u, const double* v, int32_t inward, double sq_thresh, uint32_t min_it, uint32_t max_it,
                                  uint32_t seed, int32_t min_num_inliers, double E_cm[9], double R_cm[9], uint8_t* inlier_mask, uint32_t* iterations,
                                  double* best_score) {
    Rays R{n, u, v};
    MSACOptions o; o.sq_thresh = sq_thresh; o.num_lo_steps = 0; o.num_lsq_it = 0; o.final_lsq = true; o.min_it = min_it; o.max_it = max_it; o.seed = seed;
    SphericalSolver solver{R, inward != 0};
    ORACLE_LOMSAC<SphericalSolver, EMat> M(solver, o);
    EMat Em{}; MSACStats st;
    M.estimate(&Em, &st);
    double E[9]; std::memcpy(E, Em.data(), 72);
    const uint32_t it = st.iterations; const double bs = st.best_score;
    int nin = 0;
    for (int i = 0; i < n; i++) { const bool in = sampson(E, u + 3 * i, v + 3 * i) < sq_thresh; if (inlier_mask) inlier_mask[i] = in; nin += in; }
    if (iterations) *iterations = it; if (best_score) *best_score = bs;
    rm_to_cm(E, E_cm);
    double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (nin > min_num_inliers) { double r[3], t[3]; decompose_E(E, inward != 0, r, t); rm_so3exp(r, Rm); }
    rm_to_cm(Rm, R_cm);
    return nin;
}

// LocallyOptimizedMSAC with every LORansacOptions field exposed (tests of the reference-trace GPU mode): lo_steps / lsq_iterations as in
// LORansacOptions (ransac.h:62-88), final_lsq = final_least_squares_.  stats: [2] = num_iterations, number_lo_iterations.
extern "C" int oracle_lomsac_pair(int32_t n, const double* u, const double* v, int32_t inward, int32_t use_poly, double sq_thresh, uint32_t min_it, uint32_t max_it,
                                  double success_prob, uint32_t seed, int32_t num_lo_steps, int32_t num_lsq_it, double thresh_mult, int32_t min_sample_mult,
                                  int32_t non_min_mult, uint32_t lo_start, int32_t final_lsq, int32_t min_num_inliers, double E_cm[9], double R_cm[9],
                                  uint8_t* inlier_mask, uint32_t* stats, double* best_score) {
    Rays R{n, u, v};
    MSACOptions o; o.sq_thresh = sq_thresh; o.num_lo_steps = num_lo_steps; o.num_lsq_it = num_lsq_it; o.final_lsq = final_lsq != 0; o.min_it = min_it; o.max_it = max_it;
    o.seed = seed; o.prob = success_prob; o.thresh_mult = thresh_mult; o.min_sample_mult = min_sample_mult; o.non_min_mult = non_min_mult; o.lo_start = lo_start;
    struct PolySolver : SphericalSolver {
        bool poly;
        PolySolver(const Rays& r, bool in, bool p) : SphericalSolver{r, in}, poly(p) {}
        int MinimalSolver(const std::vector<int>& sample, std::vector<EMat>* Es) const {          // src/spherical_estimator.cpp:80-84
            if (!poly) return SphericalSolver::MinimalSolver(sample, Es);
            double buf[36]; const int k = solver_polynomial(R, sample.data(), (int)sample.size(), buf);
            Es->resize(k); for (int i = 0; i < k; i++) std::memcpy((*Es)[i].data(), buf + 9 * i, 72);
            return k;
        }
    } solver(R, inward != 0, use_poly != 0);
    ORACLE_LOMSAC<PolySolver, EMat> M(solver, o);
    EMat Em{}; MSACStats st;
    M.estimate(&Em, &st);
    double E[9]; std::memcpy(E, Em.data(), 72);
    int nin = 0;
    const bool have = st.best_score < std::numeric_limits<double>::max();
    for (int i = 0; i < n; i++) { const bool in = have && sampson(E, u + 3 * i, v + 3 * i) < sq_thresh; if (inlier_mask) inlier_mask[i] = in; nin += in; }
    if (stats) { stats[0] = st.iterations; stats[1] = (uint32_t)st.lo_count; }
    if (best_score) *best_score = st.best_score;
    rm_to_cm(E, E_cm);
    double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (have && nin > min_num_inliers) { double r[3], t[3]; decompose_E(E, inward != 0, r, t); rm_so3exp(r, Rm); }
    rm_to_cm(Rm, R_cm);
    return nin;
}
// Reference-pin hooks (tests/test_reference_pins_cpu.py).  B: 6x3 row-major nullspace basis; variant 0 = action matrix
// (C as src/spherical_solvers.cpp:272-277 lays it out), 1 = polynomial (:339-621: rows halved, its monomial order);
// C_out 6x10 row-major; Es_out 4x9 COLUMN-major like every 3x3 of this API; abcde_out (variant 1): the quartic of :623-627.
extern "C" int oracle_solver_from_basis(const double B_rm[18], int32_t variant, double C_out[60], double Es_out[36], double imag_out[4], double abcde_out[5]) {
    double B[6][3]; for (int i = 0; i < 6; i++) for (int j = 0; j < 3; j++) B[i][j] = B_rm[3 * i + j];
    Cub rows[6]; constraints_from_B(B, rows);
    for (int r = 0; r < 6; r++) for (int k = 0; k < 10; k++) C_out[r * 10 + k] = variant ? 0.5 * rows[r].c[kPolyPerm[k]] : rows[r].c[k];
    double Es[36]; for (double& x : Es) x = 0;
    const int k = variant ? polynomial_back(B, rows, Es, imag_out, abcde_out) : action_matrix_back(B, rows, Es);
    for (int s = 0; s < 4; s++) rm_to_cm(Es + 9 * s, Es_out + 9 * s);
    return k;
}
extern "C" void oracle_solve_quartic(double a, double b, double c, double d, double e, double re_im[8]) {      // src/spherical_solvers.cpp:15-69
    cd roots[4]; solve_quartic(a, b, c, d, e, roots);
    for (int i = 0; i < 4; i++) { re_im[2 * i] = roots[i].real(); re_im[2 * i + 1] = roots[i].imag(); }
}
// which LO-MSAC this library was built around: 0 = the restatement (lomsac.hpp), 1 = the reference's include/RansacLib (oracle/_ref only)
extern "C" int32_t oracle_lomsac_is_reference(void) {
#ifdef SSFM_ORACLE_REAL_RANSACLIB
    return 1;
#else
    return 0;
#endif
}
// SphericalEstimator::NonMinimalSolver (src/spherical_estimator.cpp:86-108)
extern "C" int oracle_nonminimal_solver(int32_t n, const double* u, const double* v, int32_t ns, const int32_t* sample, double E_cm[9]) {
    Rays R{n, u, v}; SphericalSolver S{R, false};
    std::vector<int> s(sample, sample + ns); EMat E{};
    const int ok = S.NonMinimalSolver(s, &E);
    rm_to_cm(E.data(), E_cm);
    return ok;
}
// nraw raw words of std::mt19937(seed), then std::uniform_int_distribution<int>(lo[i], hi[i]) draws of the same engine (libstdc++)
extern "C" void oracle_mt19937_draws(uint32_t seed, int32_t n, const int32_t* lo, const int32_t* hi, int32_t* out, int32_t nraw, uint32_t* raw) {
    std::mt19937 rng; rng.seed(seed);
    for (int i = 0; i < nraw; i++) raw[i] = (uint32_t)rng();
    for (int i = 0; i < n; i++) { std::uniform_int_distribution<int> d(lo[i], hi[i]); out[i] = d(rng); }
}

// LeastSquares call log (single-threaded use): enable != 0 clears and starts recording
extern "C" void oracle_lsq_log_enable(int32_t enable) { g_lsq_log_on = enable != 0; if (enable) { g_lsq_log.clear(); g_nonmin_log.clear(); } }
extern "C" int32_t oracle_nonmin_log_count() { return (int32_t)g_nonmin_log.size(); }
extern "C" int32_t oracle_nonmin_log_get(int32_t i, int32_t* sample9, double E_out_cm[9], int32_t* after_lsq_calls) {
    if (i < 0 || i >= (int)g_nonmin_log.size()) return -1;
    const NonMinRecord& r = g_nonmin_log[i];
    for (int k = 0; k < (int)r.sample.size() && k < 9; k++) sample9[k] = r.sample[k];
    rm_to_cm(r.E_out, E_out_cm); *after_lsq_calls = r.after_lsq_calls;
    return (int32_t)r.sample.size();
}
extern "C" int32_t oracle_lsq_log_count() { return (int32_t)g_lsq_log.size(); }
extern "C" int32_t oracle_lsq_log_get(int32_t i, int32_t cap, int32_t* sample, double E_in_cm[9], double E_out_cm[9], double x[6], int32_t* iterations) {
    if (i < 0 || i >= (int)g_lsq_log.size()) return -1;
    const LsqRecord& r = g_lsq_log[i];
    for (int k = 0; k < (int)r.sample.size() && k < cap; k++) sample[k] = r.sample[k];
    rm_to_cm(r.E_in, E_in_cm); rm_to_cm(r.E_out, E_out_cm); std::memcpy(x, r.x, 48); *iterations = r.iterations;
    return (int32_t)r.sample.size();
}

extern "C" void oracle_dk_histogram(int64_t* out201, int32_t reset) { for (int i = 0; i <= 200; i++) { out201[i] = g_dk_hist[i]; if (reset) g_dk_hist[i] = 0; } }

extern "C" void oracle_lsq_counters(int64_t* out3, int32_t reset) { out3[0] = g_lsq_calls; out3[1] = g_lsq_iterations; out3[2] = g_lsq_points; if (reset) g_lsq_calls = g_lsq_iterations = g_lsq_points = 0; }
