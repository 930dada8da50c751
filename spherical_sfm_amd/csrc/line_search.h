// spherical_sfm_amd -- the projected Armijo line search of Ceres' trust-region loop (host side of the pose-graph solver).
//
// optimize_rotations_and_focal_length bounds its focal multiplier (reference src/uncalibrated_pose_graph.cpp:181-182), which makes the
// problem "constrained" for Ceres 2.2: TrustRegionMinimizer then sends every valid step through DoLineSearch before evaluating the
// candidate -- an ARMIJO search from step size 1 along delta on f(a) = cost(Plus(x, a delta)) (Plus projects onto the box), cubic
// interpolation through values and directional derivatives, sufficient decrease 1e-4, contraction to [1e-3, 0.6] of the current step,
// at most 20 iterations, give up below |a delta|_inf = 1e-9; delta *= a on success, unchanged otherwise.
// Everything here is a handful of flops on the host; the function evaluations are kernel launches (rotavg_solver.hip: k_rot_ls_eval).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>

namespace ssfm {

struct LsSample { double a = 0, f = 0, df = 0; bool has_f = false, has_df = false; };

struct LsPoly {                      // c[0] x^deg + ... + c[deg]
    double c[6]; int deg;
    double at(double x) const { double v = 0; for (int i = 0; i <= deg; i++) v = v * x + c[i]; return v; }
};

// polynomial through the valid values / slopes of up to three samples: dense solve with complete pivoting
inline LsPoly ls_interpolate(const LsSample* s, int ns) {
    int m = 0; for (int i = 0; i < ns; i++) m += (s[i].has_f ? 1 : 0) + (s[i].has_df ? 1 : 0);
    LsPoly P; P.deg = m - 1; for (double& v : P.c) v = 0;
    double A[6][7]; int r = 0;
    for (int i = 0; i < ns; i++) {
        if (s[i].has_f) { for (int j = 0; j < m; j++) A[r][j] = std::pow(s[i].a, P.deg - j); A[r][m] = s[i].f; r++; }
        if (s[i].has_df) { for (int j = 0; j < m; j++) A[r][j] = (j < P.deg) ? (P.deg - j) * std::pow(s[i].a, P.deg - j - 1) : 0.0; A[r][m] = s[i].df; r++; }
    }
    int perm[6]; for (int i = 0; i < m; i++) perm[i] = i;
    int rank = m;
    for (int k = 0; k < m; k++) {
        int bi = k, bj = k; double big = 0;
        for (int i = k; i < m; i++) for (int j = k; j < m; j++) if (std::fabs(A[i][j]) > big) { big = std::fabs(A[i][j]); bi = i; bj = j; }
        if (big == 0.0) { rank = k; break; }
        if (bi != k) for (int j = 0; j <= m; j++) std::swap(A[k][j], A[bi][j]);
        if (bj != k) { for (int i = 0; i < m; i++) std::swap(A[i][k], A[i][bj]); std::swap(perm[k], perm[bj]); }
        for (int i = k + 1; i < m; i++) { const double q = A[i][k] / A[k][k]; for (int j = k; j <= m; j++) A[i][j] -= q * A[k][j]; }
    }
    double y[6] = {0, 0, 0, 0, 0, 0};
    for (int k = rank - 1; k >= 0; k--) { double v = A[k][m]; for (int j = k + 1; j < rank; j++) v -= A[k][j] * y[j]; y[k] = v / A[k][k]; }
    for (int k = 0; k < m; k++) P.c[perm[k]] = y[k];
    return P;
}

// real parts of the roots of c[0] x^deg + ... (closed forms up to degree 2, simultaneous iteration above); returns their number
inline int ls_root_real_parts(const double* cin, int deg, double* out) {
    while (deg > 0 && cin[0] == 0.0) { cin++; deg--; }
    if (deg <= 0) return 0;
    if (deg == 1) { out[0] = -cin[1] / cin[0]; return 1; }
    if (deg == 2) {
        const double a = cin[0], b = cin[1], c = cin[2], disc = b * b - 4 * a * c, sq = std::sqrt(std::fabs(disc));
        if (disc >= 0) { if (b >= 0) { out[0] = (-b - sq) / (2 * a); out[1] = (2 * c) / (-b - sq); } else { out[0] = (2 * c) / (-b + sq); out[1] = (-b + sq) / (2 * a); } }
        else out[0] = out[1] = -b / (2 * a);
        return 2;
    }
    typedef std::complex<double> cd;
    double m[6]; for (int i = 0; i <= deg; i++) m[i] = cin[i] / cin[0];
    double bound = 1.0; for (int i = 1; i <= deg; i++) bound = std::max(bound, 1.0 + std::pow(std::fabs(m[i]), 1.0 / i));
    cd z[5]; const cd seed(0.4, 0.9); for (int i = 0; i < deg; i++) z[i] = bound * std::pow(seed, i + 1) / std::pow(std::abs(seed), i + 1);
    for (int it = 0; it < 500; it++) {
        double moved = 0;
        for (int i = 0; i < deg; i++) {
            cd p = 0; for (int k = 0; k <= deg; k++) p = p * z[i] + m[k];
            cd q = 1; for (int j = 0; j < deg; j++) if (j != i) q *= (z[i] - z[j]);
            if (std::abs(q) == 0) q = 1e-300;
            const cd dz = p / q; z[i] -= dz; moved = std::max(moved, std::abs(dz));
        }
        if (moved < 1e-16 * bound) break;
    }
    for (int i = 0; i < deg; i++) out[i] = z[i].real();
    return deg;
}

// step size minimising the interpolant on [lo, hi]: mid point, both ends, the critical points, then the sample abscissae inside
inline double ls_next_step(const LsSample& zero, const LsSample& prev, const LsSample& cur, double lo, double hi) {
    if (!cur.has_f) return std::min(std::max(0.5 * cur.a, lo), hi);
    LsSample s[3] = {zero, cur, prev}; const int ns = prev.has_f ? 3 : 2;
    const LsPoly P = ls_interpolate(s, ns);
    double best_a = 0.5 * (lo + hi), best = P.at(best_a);
    auto consider = [&](double a) { const double v = P.at(a); if (v < best) { best = v; best_a = a; } };
    consider(lo); consider(hi);
    if (P.deg >= 2) {
        double d[6]; for (int i = 0; i < P.deg; i++) d[i] = (P.deg - i) * P.c[i];
        double roots[5]; const int nr = ls_root_real_parts(d, P.deg - 1, roots);
        for (int i = 0; i < nr; i++) if (roots[i] >= lo && roots[i] <= hi) consider(roots[i]);
    }
    for (int i = 0; i < ns; i++) if (s[i].a >= lo && s[i].a <= hi) consider(s[i].a);
    return best_a;
}

// Eval: LsSample operator()(double a).  True + *a_out when a step size with sufficient decrease was found.
template <class Eval>
inline bool ls_armijo(Eval&& eval, double f0, double df0, double delta_max_norm, double* a_out) {
    LsSample zero; zero.a = 0; zero.f = f0; zero.df = df0; zero.has_f = zero.has_df = true;
    LsSample prev, cur = eval(1.0);
    for (int tries = 0; !cur.has_f || cur.f > f0 + 1e-4 * df0 * cur.a;) {
        if (++tries >= 20) return false;
        const double a = ls_next_step(zero, prev, cur, 1e-3 * cur.a, 0.6 * cur.a);
        if (a * delta_max_norm < 1e-9) return false;
        prev = cur; cur = eval(a);
    }
    *a_out = cur.a;
    return true;
}

}  // namespace ssfm
