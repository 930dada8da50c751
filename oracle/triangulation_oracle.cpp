// ORACLE (test infrastructure only) -- CPU restatement of SfM::Retriangulate and its estimator.
//
//  * SfM::Retriangulate                      src/sfm.cpp:156-192  (per point: observations over cameras in index order;
//                                            point zeroed; < 3 observations -> stays zero; LO-MSAC with
//                                            squared_inlier_threshold_ = 4, final_least_squares_ = true, other options
//                                            default; < 3 inliers -> stays zero)
//  * TriangulationEstimator::EvaluateModelOnPoint   src/triangulation_estimator.cpp:46-54 (behind the camera -> DBL_MAX)
//  * MinimalSolver / NonMinimalSolver (DLT, SVD)    src/triangulation_estimator.cpp:56-86
//  * LeastSquares (TriangulationError, point-only)  src/triangulation_estimator.cpp:18-44, 88-127
//  * the LO-MSAC loop                               oracle/lomsac.hpp (include/RansacLib)
// PARITY PARTLY PINNED (ssfm_oracle.h): the per-point LO-MSAC control flow against the reference's own RansacLib template; the estimator (DLT by SVD, Ceres fit) restated.
#include <omp.h>
#include <array>
#include <cstring>
#include <vector>
#include "lm.hpp"
#ifdef SSFM_ORACLE_REAL_RANSACLIB      // oracle/_ref build: the reference's own include/RansacLib drives the same estimator (lomsac_reference.hpp)
#include "lomsac_reference.hpp"
#define ORACLE_LOMSAC LoMsacReference
#else
#include "lomsac.hpp"
#define ORACLE_LOMSAC LoMsac
#endif
#include "rotation.hpp"
#include "ssfm_oracle.h"

namespace oracle {

struct TriObs { double t[3], r[3], P[12], x[2], focal; };      // P = [so3exp(r) | t], row-major 3x4 (Pose::P, src/sfm_types.cpp:14-19)
typedef std::array<double, 3> Pt;

// smallest right singular vector of A (m x 4, row-major) by one-sided Jacobi (what JacobiSVD's V.col(3) is, up to sign)
static void smallest_right_singular_vector(std::vector<double> A, int m, double out[4]) {
    double V[16]; for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 3; p++) for (int q = p + 1; q < 4; q++) {
            double alpha = 0, beta = 0, gamma = 0;
            for (int i = 0; i < m; i++) { alpha += A[i * 4 + p] * A[i * 4 + p]; beta += A[i * 4 + q] * A[i * 4 + q]; gamma += A[i * 4 + p] * A[i * 4 + q]; }
            if (gamma == 0.0) continue;
            off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta)), c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
            for (int i = 0; i < m; i++) { const double ap = A[i * 4 + p], aq = A[i * 4 + q]; A[i * 4 + p] = c * ap - s * aq; A[i * 4 + q] = s * ap + c * aq; }
            for (int i = 0; i < 4; i++) { const double vp = V[i * 4 + p], vq = V[i * 4 + q]; V[i * 4 + p] = c * vp - s * vq; V[i * 4 + q] = s * vp + c * vq; }
        }
        if (off < 1e-15) break;
    }
    int best = 0; double bn = 1e300;
    for (int q = 0; q < 4; q++) { double nq = 0; for (int i = 0; i < m; i++) nq += A[i * 4 + q] * A[i * 4 + q]; if (nq < bn) { bn = nq; best = q; } }
    for (int i = 0; i < 4; i++) out[i] = V[i * 4 + best];
}

template <typename T>
static inline void tri_residual(const TriObs& o, const T X[3], T res[2]) {           // src/triangulation_estimator.cpp:20-43
    const T r[3] = {T(o.r[0]), T(o.r[1]), T(o.r[2])};
    T PX[3]; AngleAxisRotatePoint(r, X, PX);
    PX[0] = PX[0] + o.t[0]; PX[1] = PX[1] + o.t[1]; PX[2] = PX[2] + o.t[2];
    res[0] = o.focal * (PX[0] / PX[2]) - o.x[0];
    res[1] = o.focal * (PX[1] / PX[2]) - o.x[1];
}

struct TriLSQ : LMProblem {
    const std::vector<TriObs>& obs; const std::vector<int>& idx;
    std::vector<double> res, J;
    TriLSQ(const std::vector<TriObs>& o, const std::vector<int>& i) : obs(o), idx(i), res(2 * i.size()), J(6 * i.size()) {}
    int num_parameters() const override { return 3; }
    bool cost_only(const double* x, double* cost) override {
        double c = 0; for (int k : idx) { double r[2]; tri_residual<double>(obs[k], x, r); c += 0.5 * (r[0] * r[0] + r[1] * r[1]); }
        *cost = c; return std::isfinite(c);
    }
    bool linearize(const double* x, double* cost, double* g) override {
        typedef Jet<3> J3; double c = 0; g[0] = g[1] = g[2] = 0;
        for (size_t q = 0; q < idx.size(); q++) {
            J3 X[3] = {J3(x[0], 0), J3(x[1], 1), J3(x[2], 2)}, r[2];
            tri_residual<J3>(obs[idx[q]], X, r);
            for (int a = 0; a < 2; a++) { res[2 * q + a] = r[a].a; for (int k = 0; k < 3; k++) { J[6 * q + 3 * a + k] = r[a].v[k]; g[k] += r[a].v[k] * r[a].a; } c += 0.5 * r[a].a * r[a].a; }
        }
        *cost = c; return std::isfinite(c);
    }
    void squared_column_norms(const double* s, double* out) override {
        out[0] = out[1] = out[2] = 0; for (size_t q = 0; q < 2 * idx.size(); q++) for (int k = 0; k < 3; k++) out[k] += J[3 * q + k] * J[3 * q + k];
        if (s) for (int k = 0; k < 3; k++) out[k] *= s[k] * s[k];
    }
    bool solve(const double* s, const double* D, double* y) override {
        double A[9] = {0}, b[3] = {0};
        for (size_t q = 0; q < 2 * idx.size(); q++) for (int a = 0; a < 3; a++) { const double ja = J[3 * q + a] * s[a]; b[a] += ja * res[q]; for (int c = 0; c < 3; c++) A[3 * a + c] += ja * J[3 * q + c] * s[c]; }
        for (int a = 0; a < 3; a++) A[4 * a] += D[a] * D[a];
        double L[9] = {0};
        for (int j = 0; j < 3; j++) { double d = A[4 * j]; for (int k = 0; k < j; k++) d -= L[3 * j + k] * L[3 * j + k]; if (!(d > 0)) return false; L[4 * j] = std::sqrt(d);
            for (int i = j + 1; i < 3; i++) { double v = A[3 * i + j]; for (int k = 0; k < j; k++) v -= L[3 * i + k] * L[3 * j + k]; L[3 * i + j] = v / L[4 * j]; } }
        double z[3]; for (int i = 0; i < 3; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[3 * i + k] * z[k]; z[i] = v / L[4 * i]; }
        for (int i = 2; i >= 0; i--) { double v = z[i]; for (int k = i + 1; k < 3; k++) v -= L[3 * k + i] * y[k]; y[i] = v / L[4 * i]; }
        return true;
    }
    double model_cost_change(const double* s, const double* step) override {
        double a = 0; for (size_t q = 0; q < 2 * idx.size(); q++) { double m = 0; for (int k = 0; k < 3; k++) m += J[3 * q + k] * s[k] * step[k]; a += m * (res[q] + 0.5 * m); } return -a;
    }
    void plus(const double* x, const double* d, double* o) override { for (int k = 0; k < 3; k++) o[k] = x[k] + d[k]; }
};

struct TriSolver {                                     // TriangulationEstimator, include/sphericalsfm/triangulation_estimator.h:18-39
    const std::vector<TriObs>& obs;
    int min_sample_size() const { return 2; }
    int non_minimal_sample_size() const { return 2; }
    int num_data() const { return (int)obs.size(); }
    double EvaluateModelOnPoint(const Pt& X, int i) const {
        const TriObs& o = obs[i];
        const double px = o.P[0] * X[0] + o.P[1] * X[1] + o.P[2] * X[2] + o.t[0], py = o.P[4] * X[0] + o.P[5] * X[1] + o.P[6] * X[2] + o.t[1],
                     pz = o.P[8] * X[0] + o.P[9] * X[1] + o.P[10] * X[2] + o.t[2];
        if (pz < 0) return std::numeric_limits<double>::max();
        const double r0 = o.focal * px / pz - o.x[0], r1 = o.focal * py / pz - o.x[1];
        return r0 * r0 + r1 * r1;
    }
    int NonMinimalSolver(const std::vector<int>& sample, Pt* X) const {
        const int N = (int)sample.size();
        std::vector<double> A((size_t)2 * N * 4);
        for (int n = 0; n < N; n++) {
            const TriObs& o = obs[sample[n]];
            const double p0 = o.x[0] / o.focal, p1 = o.x[1] / o.focal;
            for (int k = 0; k < 4; k++) { A[(2 * n) * 4 + k] = o.P[8 + k] * p0 - o.P[k]; A[(2 * n + 1) * 4 + k] = o.P[8 + k] * p1 - o.P[4 + k]; }
        }
        double Xh[4]; smallest_right_singular_vector(A, 2 * N, Xh);
        (*X)[0] = Xh[0] / Xh[3]; (*X)[1] = Xh[1] / Xh[3]; (*X)[2] = Xh[2] / Xh[3];
        return 1;
    }
    int MinimalSolver(const std::vector<int>& sample, std::vector<Pt>* out) const { Pt X; if (!NonMinimalSolver(sample, &X)) return 0; out->assign(1, X); return 1; }
    void LeastSquares(const std::vector<int>& sample, Pt* X) const {
        TriLSQ P(obs, sample);
        LMOptions o; o.max_num_iterations = 200; o.max_num_consecutive_invalid_steps = 10;      // src/triangulation_estimator.cpp:119-123
        lm_minimize(P, o, X->data());
    }
};

}  // namespace oracle
using namespace oracle;

// observation lists of every point over the cameras, ascending; map semantics (last value of a repeated key), src/sfm.cpp:164-169
static std::vector<std::vector<int64_t>> point_lists(const oracle_ba_problem* p) {
    const int Np = p->num_points; const int64_t M = p->num_observations;
    std::vector<std::vector<int64_t>> per_pt(Np);
    for (int64_t i = 0; i < M; i++) if (p->obs_pt[i] >= 0 && p->obs_pt[i] < Np) per_pt[p->obs_pt[i]].push_back(i);
    for (auto& ids : per_pt) {
        std::stable_sort(ids.begin(), ids.end(), [&](int64_t a, int64_t b) { return p->obs_cam[a] < p->obs_cam[b]; });
        std::vector<int64_t> keep;
        for (size_t k = 0; k < ids.size(); k++) if (!(k + 1 < ids.size() && p->obs_cam[ids[k + 1]] == p->obs_cam[ids[k]])) keep.push_back(ids[k]);
        ids.swap(keep);
    }
    return per_pt;
}
static std::vector<TriObs> point_observations(const oracle_ba_problem* p, const std::vector<int64_t>& ids) {
    std::vector<TriObs> obs;
    for (int64_t id : ids) {
        const int c = p->obs_cam[id];
        TriObs o; std::memcpy(o.t, &p->cameras[(size_t)c * 6], 24); std::memcpy(o.r, &p->cameras[(size_t)c * 6 + 3], 24);
        double Rc[9]; so3exp(o.r, Rc);                                                         // column-major; Pose::Pose, src/sfm_types.cpp:14-19
        for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) o.P[4 * a + b] = Rc[a + 3 * b]; o.P[4 * a + 3] = o.t[a]; }
        o.x[0] = p->obs_xy[2 * id]; o.x[1] = p->obs_xy[2 * id + 1]; o.focal = *p->focal;
        obs.push_back(o);
    }
    return obs;
}

// points: in/out.  Every point index in [0, num_points) "exists"; observations of a point over the cameras, ascending.
// Optional trace outputs: stats_out [2 * num_points] = RansacStatistics::num_iterations, number_lo_iterations of every point's run;
// inlier_flags_out [num_observations] = 1 where the observation is in the final stats.inlier_indices of its point.
extern "C" int oracle_retriangulate_ex(oracle_ba_problem* p, int32_t num_threads, int32_t* num_inliers_out, uint32_t* stats_out, uint8_t* inlier_flags_out) {
    const int Np = p->num_points; const int64_t M = p->num_observations;
    const std::vector<std::vector<int64_t>> per_pt = point_lists(p);
    if (inlier_flags_out) std::memset(inlier_flags_out, 0, (size_t)M);
    omp_set_num_threads(std::max(1, std::min(num_threads > 0 ? num_threads : 1, omp_get_num_procs())));
#pragma omp parallel for schedule(dynamic, 64)
    for (int j = 0; j < Np; j++) {
        const std::vector<TriObs> obs = point_observations(p, per_pt[j]);
        double* X = &p->points[(size_t)j * 3]; X[0] = X[1] = X[2] = 0.0;                            // src/sfm.cpp:172
        if (num_inliers_out) num_inliers_out[j] = 0;
        if (stats_out) { stats_out[2 * j] = 0; stats_out[2 * j + 1] = 0; }
        if (obs.size() < 3) continue;
        MSACOptions o; o.sq_thresh = 4.0; o.final_lsq = true;                                      // src/sfm.cpp:175-177
        TriSolver solver{obs};
        ORACLE_LOMSAC<TriSolver, Pt> R(solver, o);
        Pt Xm{}; MSACStats st;
        const int nin = R.estimate(&Xm, &st);
        if (num_inliers_out) num_inliers_out[j] = nin;
        if (stats_out) { stats_out[2 * j] = st.iterations; stats_out[2 * j + 1] = (uint32_t)st.lo_count; }
        if (inlier_flags_out) for (int k : st.inliers) inlier_flags_out[per_pt[j][k]] = 1;
        if (nin < 3) continue;                                                                     // src/sfm.cpp:186
        X[0] = Xm[0]; X[1] = Xm[1]; X[2] = Xm[2];
    }
    return 0;
}
extern "C" int oracle_retriangulate(oracle_ba_problem* p, int32_t num_threads, int32_t* num_inliers_out) {
    return oracle_retriangulate_ex(p, num_threads, num_inliers_out, nullptr, nullptr);
}

// The estimator's pieces on chosen observation subsets (bit-for-bit parity tests of the device code).  Task t works on point
// task_pt[t] with the sample lists[task_ptr[t] .. task_ptr[t+1]) = positions in that point's observation list (cameras ascending).
// what 0: NonMinimalSolver(sample)           -> out[4t..] = X, 0
// what 1: LeastSquares(sample, X_in[t])      -> out[4t..] = X, Levenberg-Marquardt iterations
// what 2: ScoreModel / GetInliers(X_in[t])   -> out[4t..] = MSAC score at threshold 4, inliers at 4, inliers at 4 sqrt 2, error of observation 0
extern "C" int oracle_tri_probe(oracle_ba_problem* p, int32_t what, int32_t tasks, const int32_t* task_pt, const int32_t* task_ptr, const int32_t* lists,
                                const double* X_in, double* out) {
    const std::vector<std::vector<int64_t>> per_pt = point_lists(p);
    for (int t = 0; t < tasks; t++) {
        const int j = task_pt[t];
        if (j < 0 || j >= p->num_points) return -1;
        const std::vector<TriObs> obs = point_observations(p, per_pt[j]);
        std::vector<int> sample(lists + task_ptr[t], lists + task_ptr[t + 1]);
        for (int k : sample) if (k < 0 || k >= (int)obs.size()) return -2;
        TriSolver S{obs};
        double* o = out + 4 * (size_t)t;
        if (what == 0) { Pt X{}; S.NonMinimalSolver(sample, &X); o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; o[3] = 0; }
        else if (what == 1) {
            Pt X = {X_in[3 * t], X_in[3 * t + 1], X_in[3 * t + 2]};
            TriLSQ P(obs, sample); LMOptions lo; lo.max_num_iterations = 200; lo.max_num_consecutive_invalid_steps = 10;
            const LMSummary sm = lm_minimize(P, lo, X.data());
            o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; o[3] = sm.iterations;
        } else {
            const Pt X = {X_in[3 * t], X_in[3 * t + 1], X_in[3 * t + 2]};
            double sc = 0; int n1 = 0, n2 = 0;
            for (int i = 0; i < (int)obs.size(); i++) { const double e = S.EvaluateModelOnPoint(X, i); sc += std::min(e, 4.0); n1 += e < 4.0; n2 += e < 4.0 * std::sqrt(2.0); }
            o[0] = sc; o[1] = n1; o[2] = n2; o[3] = obs.empty() ? 0.0 : S.EvaluateModelOnPoint(X, 0);
        }
    }
    return 0;
}
