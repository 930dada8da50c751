// ORACLE (test infrastructure only) -- the projected Armijo line search Ceres 2.2.0 runs inside its trust-region loop for
// bounds-constrained problems.
//
// Un-vendored third-party behaviour (Ceres Solver 2.2.0, docker/Dockerfile:50-56), restated from its published source:
//   TrustRegionMinimizer::Minimize  -- `if (options_.is_constrained && options_.max_num_line_search_step_size_iterations > 0)
//                                       DoLineSearch(x_, gradient_, x_cost_, &delta_);`  between ComputeTrustRegionStep and
//                                       ComputeCandidatePointAndEvaluateCost  (trust_region_minimizer.cc)
//   TrustRegionMinimizer::DoLineSearch -- ARMIJO search from step size 1 along delta with the solver's line-search defaults:
//                                       CUBIC interpolation, sufficient_function_decrease 1e-4, max_step_contraction 1e-3,
//                                       min_step_contraction 0.6, min_line_search_step_size 1e-9, at most 20 iterations;
//                                       on success delta *= optimal step size, otherwise delta is kept
//   LineSearchFunction::Evaluate    -- f(a) = cost(Plus(x, a delta)) with Plus projecting onto the bounds; f'(a) = delta . gradient there
//   ArmijoLineSearch::DoSearch, LineSearch::InterpolatingPolynomialMinimizingStepSize           (line_search.cc)
//   FindInterpolatingPolynomial (full-pivoting LU of the Vandermonde-type system), MinimizePolynomial (mid point, both ends, real parts
//   of all roots of the derivative), FindPolynomialRoots (closed forms up to degree 2, companion-matrix eigenvalues above)  (polynomial.cc)
// The reference reaches this code in optimize_rotations_and_focal_length only: it is the one solve with bounds
// (src/uncalibrated_pose_graph.cpp:181-182).  PARITY UNPINNED for this file (Ceres path; ssfm_oracle.h).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>
#include <vector>

namespace oracle {

struct FunctionSample { double x = 0, value = 0, gradient = 0; bool value_is_valid = false, gradient_is_valid = false; };

inline double evaluate_polynomial(const std::vector<double>& p, double x) { double v = 0; for (double c : p) v = v * x + c; return v; }   // highest degree first

// coefficients (highest degree first) of the polynomial through the valid values / gradients of the samples
inline std::vector<double> find_interpolating_polynomial(const std::vector<FunctionSample>& s) {
    int nc = 0; for (const auto& q : s) nc += (q.value_is_valid ? 1 : 0) + (q.gradient_is_valid ? 1 : 0);
    const int degree = nc - 1;
    std::vector<double> A((size_t)nc * nc, 0.0), b(nc, 0.0);
    int row = 0;
    for (const auto& q : s) {
        if (q.value_is_valid) { for (int j = 0; j <= degree; j++) A[(size_t)row * nc + j] = std::pow(q.x, degree - j); b[row++] = q.value; }
        if (q.gradient_is_valid) { for (int j = 0; j < degree; j++) A[(size_t)row * nc + j] = (degree - j) * std::pow(q.x, degree - j - 1); b[row++] = q.gradient; }
    }
    // Gaussian elimination with full pivoting (Eigen::FullPivLU, threshold 0)
    std::vector<int> colperm(nc); for (int i = 0; i < nc; i++) colperm[i] = i;
    int rank = nc;
    for (int k = 0; k < nc; k++) {
        int pr = k, pc = k; double best = 0;
        for (int i = k; i < nc; i++) for (int j = k; j < nc; j++) if (std::fabs(A[(size_t)i * nc + j]) > best) { best = std::fabs(A[(size_t)i * nc + j]); pr = i; pc = j; }
        if (best == 0.0) { rank = k; break; }
        if (pr != k) { for (int j = 0; j < nc; j++) std::swap(A[(size_t)k * nc + j], A[(size_t)pr * nc + j]); std::swap(b[k], b[pr]); }
        if (pc != k) { for (int i = 0; i < nc; i++) std::swap(A[(size_t)i * nc + k], A[(size_t)i * nc + pc]); std::swap(colperm[k], colperm[pc]); }
        for (int i = k + 1; i < nc; i++) {
            const double f = A[(size_t)i * nc + k] / A[(size_t)k * nc + k];
            for (int j = k; j < nc; j++) A[(size_t)i * nc + j] -= f * A[(size_t)k * nc + j];
            b[i] -= f * b[k];
        }
    }
    std::vector<double> y(nc, 0.0), x(nc, 0.0);
    for (int k = rank - 1; k >= 0; k--) { double v = b[k]; for (int j = k + 1; j < rank; j++) v -= A[(size_t)k * nc + j] * y[j]; y[k] = v / A[(size_t)k * nc + k]; }
    for (int k = 0; k < nc; k++) x[colperm[k]] = y[k];
    return x;
}

// real parts of the roots (polynomial.cc: FindPolynomialRoots)
inline std::vector<double> polynomial_root_real_parts(std::vector<double> p) {
    size_t lead = 0; while (lead + 1 < p.size() && p[lead] == 0.0) lead++;            // RemoveLeadingZeros
    p.erase(p.begin(), p.begin() + lead);
    const int degree = (int)p.size() - 1;
    std::vector<double> re;
    if (degree <= 0) return re;
    if (degree == 1) { re.push_back(-p[1] / p[0]); return re; }
    if (degree == 2) {
        const double a = p[0], b = p[1], c = p[2], D = b * b - 4 * a * c, sD = std::sqrt(std::fabs(D));
        if (D >= 0) { if (b >= 0) { re.push_back((-b - sD) / (2.0 * a)); re.push_back((2.0 * c) / (-b - sD)); } else { re.push_back((2.0 * c) / (-b + sD)); re.push_back((-b + sD) / (2.0 * a)); } }
        else { re.push_back(-b / (2.0 * a)); re.push_back(-b / (2.0 * a)); }
        return re;
    }
    // degree >= 3: Ceres takes the eigenvalues of the balanced companion matrix; here Durand-Kerner on the monic polynomial (same roots)
    typedef std::complex<double> cd;
    std::vector<double> m(p.size()); for (size_t i = 0; i < p.size(); i++) m[i] = p[i] / p[0];
    double scale = 1.0; for (int i = 1; i <= degree; i++) scale = std::max(scale, 1.0 + std::pow(std::fabs(m[i]), 1.0 / i));
    std::vector<cd> z(degree); for (int i = 0; i < degree; i++) z[i] = scale * std::pow(cd(0.4, 0.9), i + 1) / std::pow(std::abs(cd(0.4, 0.9)), i + 1);
    auto P = [&](cd x) { cd v = 0; for (double c : m) v = v * x + c; return v; };
    for (int it = 0; it < 500; it++) {
        double change = 0;
        for (int i = 0; i < degree; i++) {
            cd den = 1; for (int j = 0; j < degree; j++) if (j != i) den *= (z[i] - z[j]);
            if (std::abs(den) == 0) den = 1e-300;
            const cd dz = P(z[i]) / den; z[i] -= dz; change = std::max(change, std::abs(dz));
        }
        if (change < 1e-16 * scale) break;
    }
    for (const cd& r : z) re.push_back(r.real());
    return re;
}

inline void minimize_polynomial(const std::vector<double>& p, double x_min, double x_max, double* opt_x, double* opt_v) {
    *opt_x = (x_min + x_max) / 2.0; *opt_v = evaluate_polynomial(p, *opt_x);
    const double vmin = evaluate_polynomial(p, x_min); if (vmin < *opt_v) { *opt_v = vmin; *opt_x = x_min; }
    const double vmax = evaluate_polynomial(p, x_max); if (vmax < *opt_v) { *opt_v = vmax; *opt_x = x_max; }
    if (p.size() <= 2) return;
    const int degree = (int)p.size() - 1;
    std::vector<double> d(degree); for (int i = 0; i < degree; i++) d[i] = (degree - i) * p[i];
    for (double r : polynomial_root_real_parts(d)) {
        if (r < x_min || r > x_max) continue;
        const double v = evaluate_polynomial(p, r);
        if (v < *opt_v) { *opt_v = v; *opt_x = r; }
    }
}

inline double interpolating_step_size(const FunctionSample& lower, const FunctionSample& previous, const FunctionSample& current, double min_step, double max_step) {
    if (!current.value_is_valid) return std::min(std::max(current.x * 0.5, min_step), max_step);
    std::vector<FunctionSample> s; s.push_back(lower); s.push_back(current);       // CUBIC: values and gradients
    if (previous.value_is_valid) s.push_back(previous);
    const std::vector<double> p = find_interpolating_polynomial(s);
    double x, v; minimize_polynomial(p, min_step, max_step, &x, &v);
    for (const auto& q : s) { if (q.x < min_step || q.x > max_step) continue; const double pv = evaluate_polynomial(p, q.x); if (pv < v) { x = q.x; v = pv; } }
    return x;
}

// ArmijoLineSearch::DoSearch from step size 1.  eval(a) fills a sample.  Returns true and *step on success.
inline bool armijo_line_search(const std::function<FunctionSample(double)>& eval, double initial_cost, double initial_gradient, double direction_max_norm, double* step) {
    const double sufficient_decrease = 1e-4, max_contraction = 1e-3, min_contraction = 0.6, min_step_size = 1e-9; const int max_iterations = 20;
    FunctionSample lower; lower.x = 0; lower.value = initial_cost; lower.gradient = initial_gradient; lower.value_is_valid = lower.gradient_is_valid = true;
    FunctionSample previous, current = eval(1.0);
    int it = 0;
    while (!current.value_is_valid || current.value > initial_cost + sufficient_decrease * initial_gradient * current.x) {
        if (++it >= max_iterations) return false;
        const double a = interpolating_step_size(lower, previous, current, max_contraction * current.x, min_contraction * current.x);
        if (a * direction_max_norm < min_step_size) return false;
        previous = current; current = eval(a);
    }
    *step = current.x;
    return true;
}

}  // namespace oracle
