// ORACLE (test infrastructure only) -- CPU restatement of the SO(3) pose-graph solvers.
//
//  * RotationError               src/rotation_averaging.cpp:15-42
//  * optimize_rotations          src/rotation_averaging.cpp:44-91
//  * decompose_rotation          src/uncalibrated_pose_graph.cpp:16-31
//  * UncalibratedPoseGraphError  src/uncalibrated_pose_graph.cpp:33-79
//  * PoseGraphError              src/uncalibrated_pose_graph.cpp:81-114
//  * get_cost                    src/uncalibrated_pose_graph.cpp:116-145
//  * optimize_rotations_and_focal_length  src/uncalibrated_pose_graph.cpp:147-203
// Residuals are evaluated with dual numbers (width 6 or 7) like the reference's AutoDiffCostFunction;
// the solve is the restated Ceres loop of oracle/lm.hpp (defaults: 50 iterations) with an exact
// Cholesky of J^T J + D^2 in place of SPARSE_NORMAL_CHOLESKY.  PARITY UNPINNED for this file (Ceres path; ssfm_oracle.h).
#include <algorithm>
#include <cstring>
#include <vector>
#include "lm.hpp"
#include "rotation.hpp"
#include "skyline.hpp"
#include "ssfm_oracle.h"

namespace oracle {

static inline void softlone(double a, double s, double rho[3]) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
    rho[0] = 2.0 * b * (tmp - 1.0);
    rho[1] = std::fmax(std::numeric_limits<double>::min(), 1.0 / tmp);
    rho[2] = -(c * rho[1]) / (2.0 * sum);
}

// src/uncalibrated_pose_graph.cpp:16-31 (R column-major)
static void decompose_rotation(const double R[9], double& rx, double& ry, double& thetaxy, double& thetaz) {
    double Z[3] = {R[6], R[7], R[8]};
    const double zn = std::sqrt(Z[0] * Z[0] + Z[1] * Z[1] + Z[2] * Z[2]);
    Z[0] /= zn; Z[1] /= zn; Z[2] /= zn;
    // axis = (0,0,1) x Z
    double axis[3] = {-Z[1], Z[0], 0.0};
    const double an = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    const double rxy[3] = {axis[0] / an, axis[1] / an, axis[2] / an};
    thetaxy = std::acos(Z[2]);
    const double v[3] = {thetaxy * rxy[0], thetaxy * rxy[1], thetaxy * rxy[2]};
    double Rxy[9]; so3exp(v, Rxy);
    rx = rxy[0]; ry = rxy[1];
    double RxyT[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RxyT[i + 3 * j] = Rxy[j + 3 * i];
    double Rz[9]; mat3_mul(RxyT, R, Rz);
    double rz[3]; so3ln(Rz, rz);
    thetaz = rz[2];
}

struct EdgeConst {       // per-edge constants of the three functors
    double meas[9];      // RotationError: R ; PoseGraphError: unused
    double r[3];         // PoseGraphError: so3ln(R)
    double rx, ry, thetaxy, thetaz;   // UncalibratedPoseGraphError
};

// res = scale * log( R1 * R0^T * R^T ), shared tail of all three functors
template <typename T>
static inline void cycle_residual(const T r0[3], const T r1[3], const T R[9], double scale, T res[3]) {
    T R0[9], R1[9], A[9], C[9];
    AngleAxisToRotationMatrix(r0, R0);
    AngleAxisToRotationMatrix(r1, R1);
    mat3_mul_bt(R1, R0, A);
    mat3_mul_bt(A, R, C);
    RotationMatrixToAngleAxis(C, res);
    res[0] *= scale; res[1] *= scale; res[2] *= scale;
}

template <typename T>
static inline void edge_residual(int kind, const EdgeConst& e, double scale, const T r0[3], const T r1[3], const T& f, T res[3]) {
    T R[9];
    if (kind == 0) {
        for (int i = 0; i < 9; i++) R[i] = T(e.meas[i]);
    } else if (kind == 1) {
        T myr[3] = {T(e.r[0]), T(e.r[1]), T(e.r[2])};
        AngleAxisToRotationMatrix(myr, R);
    } else {
        const T fsq = f * f;
        const T num = 2.0 * f * std::sin(e.thetaxy);
        const T den = (1.0 + fsq) * std::cos(e.thetaxy) + (1.0 - fsq);
        const T thp = jatan2(num, den);
        const T rxy[3] = {thp * e.rx, thp * e.ry, T(0.0)};
        const T rz[3] = {T(0.0), T(0.0), T(e.thetaz)};
        T Rxy[9], Rz[9];
        AngleAxisToRotationMatrix(rxy, Rxy);
        AngleAxisToRotationMatrix(rz, Rz);
        mat3_mul(Rxy, Rz, R);
    }
    cycle_residual(r0, r1, R, scale, res);
}

struct EdgeLin { double r[3]; double J0[3][3]; double J1[3][3]; double Jf[3]; };

struct PoseGraphOracle : LMProblem {
    int n = 0, E = 0, kind = 0;
    double scale = 1.0, loss_a = 0.03;
    bool with_f = false; double f_lo = 0, f_hi = 0;
    std::vector<int> e0, e1;
    std::vector<EdgeConst> ec;
    std::vector<double> data0;          // [n*3]
    std::vector<int> node_idx;          // x-index or -1
    int f_idx = -1, nx = 0;
    std::vector<EdgeLin> lin;
    std::vector<int> sky_of_x; Skyline S; std::vector<double> rhs;

    int num_parameters() const override { return nx; }
    inline void node(const double* x, int i, double r[3]) const {
        int k = node_idx[i]; for (int d = 0; d < 3; d++) r[d] = k >= 0 ? x[k + d] : data0[i * 3 + d];
    }
    bool cost_only(const double* x, double* cost) override {
        double c = 0; const double f = with_f ? x[f_idx] : 1.0;
        for (int e = 0; e < E; e++) {
            double r0[3], r1[3], res[3]; node(x, e0[e], r0); node(x, e1[e], r1);
            edge_residual<double>(kind, ec[e], scale, r0, r1, f, res);
            double rho[3]; softlone(loss_a, res[0] * res[0] + res[1] * res[1] + res[2] * res[2], rho);
            c += 0.5 * rho[0];
        }
        *cost = c; return std::isfinite(c);
    }
    bool linearize(const double* x, double* cost, double* g) override { return eval(x, cost, g, true); }
    bool is_constrained() const override { return with_f; }                       // src/uncalibrated_pose_graph.cpp:181-182
    bool cost_and_gradient(const double* x, double* cost, double* g) override { return eval(x, cost, g, false); }
    bool eval(const double* x, double* cost, double* g, bool store) {
        typedef Jet<7> J;
        double c = 0; std::fill(g, g + nx, 0.0);
        for (int e = 0; e < E; e++) {
            double a0[3], a1[3]; node(x, e0[e], a0); node(x, e1[e], a1);
            J r0[3] = {J(a0[0], 0), J(a0[1], 1), J(a0[2], 2)}, r1[3] = {J(a1[0], 3), J(a1[1], 4), J(a1[2], 5)};
            J f(with_f ? x[f_idx] : 1.0, 6), res[3];
            edge_residual<J>(kind, ec[e], scale, r0, r1, f, res);
            double rho[3]; softlone(loss_a, res[0].a * res[0].a + res[1].a * res[1].a + res[2].a * res[2].a, rho);
            c += 0.5 * rho[0];
            const double sr = std::sqrt(rho[1]);
            EdgeLin tmp; EdgeLin& L = store ? lin[e] : tmp;
            for (int a = 0; a < 3; a++) {
                L.r[a] = sr * res[a].a;
                for (int k = 0; k < 3; k++) { L.J0[a][k] = sr * res[a].v[k]; L.J1[a][k] = sr * res[a].v[3 + k]; }
                L.Jf[a] = sr * res[a].v[6];
            }
            const int i0 = node_idx[e0[e]], i1 = node_idx[e1[e]];
            for (int a = 0; a < 3; a++) {
                for (int k = 0; k < 3; k++) { if (i0 >= 0) g[i0 + k] += L.J0[a][k] * L.r[a]; if (i1 >= 0) g[i1 + k] += L.J1[a][k] * L.r[a]; }
                if (with_f) g[f_idx] += L.Jf[a] * L.r[a];
            }
        }
        *cost = c; return std::isfinite(c);
    }
    void squared_column_norms(const double* s, double* out) override {
        std::fill(out, out + nx, 0.0);
        for (int e = 0; e < E; e++) {
            const EdgeLin& L = lin[e]; const int i0 = node_idx[e0[e]], i1 = node_idx[e1[e]];
            for (int a = 0; a < 3; a++) {
                for (int k = 0; k < 3; k++) { if (i0 >= 0) out[i0 + k] += L.J0[a][k] * L.J0[a][k]; if (i1 >= 0) out[i1 + k] += L.J1[a][k] * L.J1[a][k]; }
                if (with_f) out[f_idx] += L.Jf[a] * L.Jf[a];
            }
        }
        if (s) for (int i = 0; i < nx; i++) out[i] *= s[i] * s[i];
    }
    // gather the (scaled) row of edge e as (index, value) pairs
    inline int row(int e, int a, const double* s, int idx[7], double val[7]) const {
        const EdgeLin& L = lin[e]; int m = 0; const int i0 = node_idx[e0[e]], i1 = node_idx[e1[e]];
        if (i0 >= 0) for (int k = 0; k < 3; k++) { idx[m] = i0 + k; val[m++] = L.J0[a][k] * s[i0 + k]; }
        if (i1 >= 0) for (int k = 0; k < 3; k++) { idx[m] = i1 + k; val[m++] = L.J1[a][k] * s[i1 + k]; }
        if (with_f) { idx[m] = f_idx; val[m++] = L.Jf[a] * s[f_idx]; }
        return m;
    }
    bool solve(const double* s, const double* D, double* y) override {
        S.zero(); std::fill(rhs.begin(), rhs.end(), 0.0);
        for (int i = 0; i < nx; i++) S.at(sky_of_x[i], sky_of_x[i]) += D[i] * D[i];
        for (int e = 0; e < E; e++)
            for (int a = 0; a < 3; a++) {
                int idx[7]; double val[7]; const int m = row(e, a, s, idx, val);
                for (int u = 0; u < m; u++) {
                    const int su = sky_of_x[idx[u]];
                    rhs[su] += val[u] * lin[e].r[a];
                    for (int v = 0; v < m; v++) { const int sv = sky_of_x[idx[v]]; if (sv <= su) S.at(su, sv) += val[u] * val[v]; }
                }
            }
        if (!S.factor()) return false;
        S.solve(rhs.data());
        for (int i = 0; i < nx; i++) y[i] = rhs[sky_of_x[i]];
        return true;
    }
    double model_cost_change(const double* s, const double* step) override {
        double acc = 0;
        for (int e = 0; e < E; e++)
            for (int a = 0; a < 3; a++) {
                int idx[7]; double val[7]; const int m = row(e, a, s, idx, val);
                double mr = 0; for (int u = 0; u < m; u++) mr += val[u] * step[idx[u]];
                acc += mr * (lin[e].r[a] + 0.5 * mr);
            }
        return -acc;
    }
    void plus(const double* x, const double* d, double* out) override {
        for (int i = 0; i < nx; i++) out[i] = x[i] + d[i];
        if (with_f) out[f_idx] = std::fmin(std::fmax(out[f_idx], f_lo), f_hi);   // ParameterBlock::Plus box projection
    }

    void setup(int n_, const double* rotations, int E_, const int32_t* i0, const int32_t* i1, const double* rel, int kind_, bool hold_first) {
        n = n_; E = E_; kind = kind_;
        data0.resize((size_t)n * 3);
        for (int i = 0; i < n; i++) so3ln(&rotations[i * 9], &data0[i * 3]);
        double maxn = 0;
        e0.assign(i0, i0 + E); e1.assign(i1, i1 + E); ec.resize(E);
        for (int e = 0; e < E; e++) {
            double r[3]; so3ln(&rel[e * 9], r);
            maxn = std::fmax(maxn, std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
            std::memcpy(ec[e].meas, &rel[e * 9], 72); std::memcpy(ec[e].r, r, 24);
            if (kind == 2) { double Rr[9]; so3exp(r, Rr); decompose_rotation(Rr, ec[e].rx, ec[e].ry, ec[e].thetaxy, ec[e].thetaz); }
        }
        scale = 1.0 / maxn;
        std::vector<char> in_problem(n, 0);
        for (int e = 0; e < E; e++) in_problem[e0[e]] = in_problem[e1[e]] = 1;
        node_idx.assign(n, -1); nx = 0;
        for (int i = 0; i < n; i++) if (in_problem[i] && !(hold_first && i == 0)) { node_idx[i] = nx; nx += 3; }
        with_f = (kind == 2); if (with_f) f_idx = nx++;
        lin.resize(E);
        // elimination order over nodes
        std::vector<std::vector<int>> adj(n);
        for (int e = 0; e < E; e++) if (e0[e] != e1[e]) { adj[e0[e]].push_back(e1[e]); adj[e1[e]].push_back(e0[e]); }
        for (auto& v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
        std::vector<int> order = rcm_order(adj), nfirst(n, 0);
        sky_of_x.assign(nx, -1); int ns = 0;
        for (int k = 0; k < n; k++) { int i = order[k]; nfirst[i] = ns; if (node_idx[i] >= 0) for (int d = 0; d < 3; d++) sky_of_x[node_idx[i] + d] = ns++; }
        if (with_f) sky_of_x[f_idx] = ns++;
        std::vector<int> first(ns);
        for (int i = 0; i < n; i++) {
            if (node_idx[i] < 0) continue;
            int f = nfirst[i]; for (int j : adj[i]) f = std::min(f, nfirst[j]);
            for (int d = 0; d < 3; d++) first[sky_of_x[node_idx[i] + d]] = f;
        }
        if (with_f) first[ns - 1] = 0;
        S.init(first); rhs.assign(ns, 0.0);
    }
};

// test hook (tests/test_scipy_anchor_cpu.py solves to the exact minimum): 0 = the reference's Ceres defaults
static int g_pg_max_it = 0; static double g_pg_ftol = 1e-6, g_pg_gtol = 1e-10, g_pg_ptol = 1e-8; static int g_pg_last_contractions = 0;

static double run_pose_graph(int kind, int32_t n, double* rotations, int32_t E, const int32_t* i0, const int32_t* i1,
                             const double* rel, double* focal_length, double min_focal, double max_focal, oracle_summary* s) {
    PoseGraphOracle P;
    P.setup(n, rotations, E, i0, i1, rel, kind, true);
    std::vector<double> x(P.nx);
    for (int i = 0; i < n; i++) if (P.node_idx[i] >= 0) for (int d = 0; d < 3; d++) x[P.node_idx[i] + d] = P.data0[i * 3 + d];
    if (P.with_f) { x[P.f_idx] = 1.0; P.f_lo = min_focal / *focal_length; P.f_hi = max_focal / *focal_length; }
    LMOptions opt;   // Ceres defaults (src/rotation_averaging.cpp:75-78)
    if (g_pg_max_it > 0) { opt.max_num_iterations = g_pg_max_it; opt.function_tolerance = g_pg_ftol; opt.gradient_tolerance = g_pg_gtol; opt.parameter_tolerance = g_pg_ptol; }
    LMSummary r = lm_minimize(P, opt, x.data());
    g_pg_last_contractions = r.num_line_search_contractions;
    for (int i = 0; i < n; i++) {
        double v[3]; P.node(x.data(), i, v);
        so3exp(v, &rotations[i * 9]);      // src/rotation_averaging.cpp:88 -- every rotation is re-exponentiated
    }
    if (P.with_f) *focal_length *= x[P.f_idx];
    if (s) {
        std::memset(s, 0, sizeof(*s));
        s->termination = r.termination; s->iterations = r.iterations; s->num_successful_steps = r.num_successful_steps;
        s->num_unsuccessful_steps = r.num_unsuccessful_steps; s->num_linear_solves = r.num_linear_solves;
        s->initial_cost = r.initial_cost; s->final_cost = r.final_cost; s->num_residual_blocks = E; s->threads_used = 1;
    }
    return r.final_cost;
}

}  // namespace oracle
using namespace oracle;

extern "C" double oracle_optimize_rotations(int32_t n, double* rotations, int32_t E, const int32_t* i0, const int32_t* i1,
                                            const double* rel, oracle_summary* s) {
    return run_pose_graph(0, n, rotations, E, i0, i1, rel, nullptr, 0, 0, s);
}

extern "C" double oracle_optimize_rotations_and_focal_length(int32_t n, double* rotations, int32_t E, const int32_t* i0,
                                                             const int32_t* i1, const double* rel, double* focal_length,
                                                             double min_focal, double max_focal, oracle_summary* s) {
    return run_pose_graph(2, n, rotations, E, i0, i1, rel, focal_length, min_focal, max_focal, s);
}

extern "C" double oracle_get_cost(int32_t n, const double* rotations, int32_t E, const int32_t* i0, const int32_t* i1, const double* rel) {
    PoseGraphOracle P;
    P.setup(n, rotations, E, i0, i1, rel, 1, false);   // src/uncalibrated_pose_graph.cpp:131-143: PoseGraphError, nothing held
    std::vector<double> x(P.nx);
    for (int i = 0; i < n; i++) if (P.node_idx[i] >= 0) for (int d = 0; d < 3; d++) x[P.node_idx[i] + d] = P.data0[i * 3 + d];
    double c = 0; P.cost_only(x.data(), &c); return c;
}

extern "C" void oracle_rotation_edge(int32_t kind, const double r0[3], const double r1[3], double f, const double Rmeas[9],
                                     double scale, double res[3], double jac[21]) {
    typedef Jet<7> J;
    EdgeConst e; std::memcpy(e.meas, Rmeas, 72); so3ln(Rmeas, e.r);
    if (kind == 2) { double Rr[9]; so3exp(e.r, Rr); decompose_rotation(Rr, e.rx, e.ry, e.thetaxy, e.thetaz); }
    J a0[3] = {J(r0[0], 0), J(r0[1], 1), J(r0[2], 2)}, a1[3] = {J(r1[0], 3), J(r1[1], 4), J(r1[2], 5)}, ff(f, 6), out[3];
    edge_residual<J>(kind, e, scale, a0, a1, ff, out);
    for (int a = 0; a < 3; a++) { res[a] = out[a].a; for (int k = 0; k < 7; k++) jac[a * 7 + k] = out[a].v[k]; }
}

extern "C" void oracle_pose_graph_test_options(int32_t max_iterations, double function_tolerance, double gradient_tolerance, double parameter_tolerance) {
    g_pg_max_it = max_iterations; g_pg_ftol = function_tolerance; g_pg_gtol = gradient_tolerance; g_pg_ptol = parameter_tolerance;
}
extern "C" int32_t oracle_pose_graph_last_line_search_contractions() { return g_pg_last_contractions; }

// ---- line_search.hpp through C, for tests/test_line_search_cpu.py
extern "C" int32_t oracle_ls_polynomial(int32_t ns, const double* x, const double* f, const double* df, const uint8_t* valid /* [2*ns]: f, df per sample */, double* coeff) {
    std::vector<FunctionSample> s(ns);
    for (int i = 0; i < ns; i++) { s[i].x = x[i]; s[i].value = f[i]; s[i].gradient = df[i]; s[i].value_is_valid = valid[2 * i]; s[i].gradient_is_valid = valid[2 * i + 1]; }
    const std::vector<double> p = find_interpolating_polynomial(s);
    for (size_t i = 0; i < p.size(); i++) coeff[i] = p[i];
    return (int32_t)p.size();
}
extern "C" int32_t oracle_ls_roots(int32_t degree, const double* coeff, double* real_parts) {
    const std::vector<double> r = polynomial_root_real_parts(std::vector<double>(coeff, coeff + degree + 1));
    for (size_t i = 0; i < r.size(); i++) real_parts[i] = r[i];
    return (int32_t)r.size();
}
// s = [f(0), f'(0), a, f(a), f'(a)]: the next trial step on [lo a, hi a]
extern "C" double oracle_ls_step(const double* s, double lo, double hi) {
    FunctionSample z, c, prev; z.x = 0; z.value = s[0]; z.gradient = s[1]; z.value_is_valid = z.gradient_is_valid = true;
    c.x = s[2]; c.value = s[3]; c.gradient = s[4]; c.value_is_valid = c.gradient_is_valid = true;
    return interpolating_step_size(z, prev, c, lo * c.x, hi * c.x);
}
extern "C" int32_t oracle_ls_armijo_poly(int32_t degree, const double* coeff, double direction_norm, double* step, int32_t* evaluations) {
    std::vector<double> p(coeff, coeff + degree + 1), d(degree);
    for (int i = 0; i < degree; i++) d[i] = (degree - i) * p[i];
    int ne = 0;
    auto eval = [&](double a) { FunctionSample q; q.x = a; q.value = evaluate_polynomial(p, a); q.gradient = evaluate_polynomial(d, a); q.value_is_valid = q.gradient_is_valid = true; ne++; return q; };
    const bool ok = armijo_line_search(eval, evaluate_polynomial(p, 0.0), evaluate_polynomial(d, 0.0), direction_norm, step);
    *evaluations = ne;
    return ok ? 1 : 0;
}
