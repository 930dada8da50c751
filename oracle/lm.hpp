// ORACLE (test infrastructure only) -- Levenberg-Marquardt trust-region loop.
//
// Restates the control flow of Ceres 2.2.0's TrustRegionMinimizer + LevenbergMarquardtStrategy
// (un-vendored third-party dependency of the reference; pinned in docker/Dockerfile:50-56), which
// is what every `ceres::Solve` on the hot path runs: src/sfm.cpp:273-276 (max_num_iterations 2000,
// max_num_consecutive_invalid_steps 100), src/rotation_averaging.cpp:75-80 and
// src/uncalibrated_pose_graph.cpp:187-191 (defaults: 50 iterations, 5 invalid steps),
// src/spherical_estimator.cpp:146-154 (200 iterations, 10 invalid steps).
//
// PARITY UNPINNED for this file: the reference holds no golden vectors for any of these solves and Ceres cannot be run here
// and Ceres cannot be built here, so this loop is anchored on Ceres' published algorithm:
//   * cost = 1/2 sum rho(|r|^2); robustified residual/Jacobian = sqrt(rho') * (r, J) because
//     rho'' <= 0 for Cauchy / SoftLOne (Corrector degenerates to scaling);
//   * Jacobi scaling: s_j = 1/(1 + |J_col_j|) from the iteration-0 Jacobian, kept for the solve;
//   * LM diagonal: D = sqrt(clamp(diag(J's^T J's), 1e-6, 1e32) / radius), refreshed only after an
//     accepted step; step solves (J's^T J's + D^2) y = J's^T r, trust_region_step = -y;
//   * model_cost_change = -(J's y')^T (r + J's y'/2) with y' = -y; step valid iff > 0;
//   * delta = s o step; candidate = Plus(x, delta) (box projection for bounded blocks);
//   * parameter tolerance |delta| <= 1e-8 (|x| + 1e-8) -> CONVERGENCE (candidate NOT taken);
//   * function tolerance |cost - candidate| <= 1e-6 cost -> CONVERGENCE (candidate NOT taken);
//   * accept iff (cost - candidate)/model_cost_change > 1e-3; then
//       radius /= max(1/3, 1 - (2 rho - 1)^3), decrease_factor = 2;
//     else radius /= decrease_factor, decrease_factor *= 2;
//   * after an accepted step: gradient max-norm |x - Plus(x, -g)|_inf <= 1e-10 -> CONVERGENCE;
//   * iteration >= max_num_iterations -> NO_CONVERGENCE; radius <= 1e-32 -> CONVERGENCE;
//   * invalid steps (solver failure or model_cost_change <= 0) count consecutively -> FAILURE;
//   * bounds-constrained problems only (Program::IsBoundsConstrained -- the reference has one: the focal multiplier of
//     src/uncalibrated_pose_graph.cpp:181-182): IterationZero projects the start point, and every valid step goes through the
//     projected Armijo line search of line_search.hpp before the candidate is evaluated (delta *= step size on success;
//     model_cost_change stays that of the full step).
// Not restated (uncertain for 2.2.0, no effect on the tests here): newer Ceres only tests the parameter tolerance after at least
// one successful step.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>
#include "line_search.hpp"

namespace oracle {

enum Termination { CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2 };

struct LMOptions {
    int max_num_iterations = 50;
    int max_num_consecutive_invalid_steps = 5;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    double min_relative_decrease = 1e-3;
    bool jacobi_scaling = true;
    int verbose = 0;
};

struct LMSummary {
    int termination = NO_CONVERGENCE;
    int iterations = 0;          // index of the last iteration (Ceres: iterations.size()-1)
    int num_successful_steps = 0;
    int num_unsuccessful_steps = 0;
    int num_linear_solves = 0;
    int num_line_search_contractions = 0;   // iterations whose step the line search shortened
    double initial_cost = 0, final_cost = 0;
};

// What the loop needs from a least-squares problem.  "Scaled" = after Jacobi column scaling.
struct LMProblem {
    virtual ~LMProblem() {}
    virtual int num_parameters() const = 0;
    // robustified cost at x, nothing stored
    virtual bool cost_only(const double* x, double* cost) = 0;
    // linearise at x: stores robustified residuals + UNSCALED Jacobian, returns cost and gradient J^T r
    virtual bool linearize(const double* x, double* cost, double* gradient) = 0;
    // squared column norms of the stored Jacobian times scale^2 (scale may be null = 1)
    virtual void squared_column_norms(const double* scale, double* out) = 0;
    // solve (Js^T Js + diag(D)^2) y = Js^T r  with Js = J diag(scale); false = solver failure
    virtual bool solve(const double* scale, const double* D, double* y) = 0;
    // returns -(Js step)^T (r + Js step / 2)
    virtual double model_cost_change(const double* scale, const double* step) = 0;
    // x_plus = x + delta, projected on bounds where present
    virtual void plus(const double* x, const double* delta, double* x_plus) = 0;
    // bounds-constrained problems: cost and gradient J^T r at x WITHOUT touching the stored linearisation (line-search evaluations)
    virtual bool is_constrained() const { return false; }
    virtual bool cost_and_gradient(const double* /*x*/, double* /*cost*/, double* /*gradient*/) { return false; }
};

inline double vec_norm(const std::vector<double>& v) {
    double s = 0; for (double x : v) s += x * x; return std::sqrt(s);
}

inline LMSummary lm_minimize(LMProblem& prob, const LMOptions& opt, double* parameters) {
    const int n = prob.num_parameters();
    LMSummary sum;
    std::vector<double> x(parameters, parameters + n), cand(n), grad(n), scale(n, 1.0), diag(n), D(n),
        step(n), delta(n), tmp(n);
    double radius = opt.initial_trust_region_radius;
    double decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int num_consecutive_invalid = 0;

    // bounds: project the start point (Ceres IterationZero for constrained problems)
    std::fill(delta.begin(), delta.end(), 0.0);
    prob.plus(x.data(), delta.data(), cand.data());
    x = cand;
    double x_norm = vec_norm(x);

    double x_cost = 0;
    auto gradient_norms = [&](double& gmax) {
        for (int i = 0; i < n; i++) tmp[i] = -grad[i];
        prob.plus(x.data(), tmp.data(), cand.data());
        gmax = 0;
        for (int i = 0; i < n; i++) gmax = std::fmax(gmax, std::fabs(x[i] - cand[i]));
    };
    if (!prob.linearize(x.data(), &x_cost, grad.data())) { sum.termination = FAILURE; return sum; }
    if (opt.jacobi_scaling) {
        prob.squared_column_norms(nullptr, scale.data());
        for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + std::sqrt(scale[i]));
    }
    double gmax; gradient_norms(gmax);
    sum.initial_cost = x_cost;
    double minimum_cost = x_cost;
    for (int i = 0; i < n; i++) parameters[i] = x[i];
    sum.num_successful_steps = 1;   // Ceres counts iteration 0 as successful
    int iteration = 0;
    bool last_step_successful = true;
    if (opt.verbose) std::printf("[oracle lm] iter %4d cost %.12e |g|inf %.3e radius %.3e\n", 0, x_cost, gmax, radius);

    while (true) {
        // ---- FinalizeIterationAndCheckIfMinimizerCanContinue
        if (iteration >= opt.max_num_iterations) { sum.termination = NO_CONVERGENCE; break; }
        if (last_step_successful && gmax <= opt.gradient_tolerance) { sum.termination = CONVERGENCE; break; }
        if (radius <= opt.min_trust_region_radius) { sum.termination = CONVERGENCE; break; }
        iteration++;
        // ---- ComputeTrustRegionStep
        if (!reuse_diagonal) {
            prob.squared_column_norms(scale.data(), diag.data());
            for (int i = 0; i < n; i++) diag[i] = std::fmin(std::fmax(diag[i], opt.min_lm_diagonal), opt.max_lm_diagonal);
        }
        for (int i = 0; i < n; i++) D[i] = std::sqrt(diag[i] / radius);
        bool ok = prob.solve(scale.data(), D.data(), step.data());
        sum.num_linear_solves++;
        reuse_diagonal = true;
        bool valid = false;
        double model_cost_change = 0;
        if (ok) {
            bool finite = true;
            for (int i = 0; i < n; i++) { step[i] = -step[i]; if (!std::isfinite(step[i])) finite = false; }
            if (finite) {
                model_cost_change = prob.model_cost_change(scale.data(), step.data());
                valid = model_cost_change > 0.0;
            }
        }
        if (!valid) {
            // ---- HandleInvalidStep
            if (++num_consecutive_invalid >= opt.max_num_consecutive_invalid_steps) { sum.termination = FAILURE; break; }
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            last_step_successful = false; sum.num_unsuccessful_steps++;
            if (opt.verbose) std::printf("[oracle lm] iter %4d invalid step, radius %.3e\n", iteration, radius);
            continue;
        }
        num_consecutive_invalid = 0;
        for (int i = 0; i < n; i++) delta[i] = step[i] * scale[i];
        // ---- DoLineSearch (is_constrained && max_num_line_search_step_size_iterations > 0)
        if (prob.is_constrained()) {
            std::vector<double> g2(n);
            auto eval = [&](double a) {
                FunctionSample fs; fs.x = a;
                for (int i = 0; i < n; i++) tmp[i] = a * delta[i];
                prob.plus(x.data(), tmp.data(), cand.data());
                double c;
                if (!prob.cost_and_gradient(cand.data(), &c, g2.data()) || !std::isfinite(c)) return fs;
                fs.value = c; fs.value_is_valid = true;
                double d = 0; for (int i = 0; i < n; i++) d += delta[i] * g2[i];
                if (std::isfinite(d)) { fs.gradient = d; fs.gradient_is_valid = true; }
                return fs;
            };
            double g0 = 0, dmax = 0; for (int i = 0; i < n; i++) { g0 += grad[i] * delta[i]; dmax = std::fmax(dmax, std::fabs(delta[i])); }
            double a = 1.0;
            if (getenv("ORACLE_LS_DEBUG")) { FunctionSample f1 = eval(1.0); std::printf("[ls] it %d cost %.12e g0 %.6e f(1) %.12e armijo_rhs %.12e\n", iteration, x_cost, g0, f1.value, x_cost + 1e-4 * g0); }
            if (armijo_line_search(eval, x_cost, g0, dmax, &a)) { for (int i = 0; i < n; i++) delta[i] *= a; if (a != 1.0) sum.num_line_search_contractions++; }
        }
        // ---- ComputeCandidatePointAndEvaluateCost
        prob.plus(x.data(), delta.data(), cand.data());
        double cand_cost;
        if (!prob.cost_only(cand.data(), &cand_cost) || !std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
        // ---- ParameterToleranceReached
        double step_norm = 0; for (int i = 0; i < n; i++) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
        step_norm = std::sqrt(step_norm);
        if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) { sum.termination = CONVERGENCE; break; }
        // ---- FunctionToleranceReached
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= opt.function_tolerance * x_cost) { sum.termination = CONVERGENCE; break; }
        // ---- IsStepSuccessful
        const double rel = (cand_cost >= std::numeric_limits<double>::max())
                               ? std::numeric_limits<double>::lowest() : cost_change / model_cost_change;
        if (rel > opt.min_relative_decrease) {
            x = cand; x_norm = vec_norm(x);
            if (!prob.linearize(x.data(), &x_cost, grad.data())) { sum.termination = FAILURE; break; }
            gradient_norms(gmax);
            { const double t = 2.0 * rel - 1.0;      // Ceres: pow(2 rho - 1, 3); here t*t*t on BOTH sides (oracle and device), because glibc's pow and the device's are not bit-identical
              radius = radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t); }
            radius = std::fmin(opt.max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = false;
            last_step_successful = true; sum.num_successful_steps++;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; for (int i = 0; i < n; i++) parameters[i] = x[i]; }
        } else {
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            last_step_successful = false; sum.num_unsuccessful_steps++;
        }
        if (opt.verbose)
            std::printf("[oracle lm] iter %4d cost %.12e change %.3e |g|inf %.3e |step| %.3e rho %.3e radius %.3e %s\n",
                        iteration, x_cost, cost_change, gmax, step_norm, rel, radius, last_step_successful ? "" : "(rejected)");
    }
    sum.iterations = iteration;
    sum.final_cost = minimum_cost;
    return sum;
}

}  // namespace oracle
