// spherical_sfm_amd -- SfM::Retriangulate on the GPU.
//
// Replaces the cv::parallel_for_ body of SfM::Retriangulate (reference src/sfm.cpp:156-192) and its estimator
// (src/triangulation_estimator.cpp:46-127): per point a LO-MSAC over 2-view DLT hypotheses scored by reprojection error
// (squared threshold 4 px^2, behind-the-camera = +inf), local optimisation, final point-only least squares, and the rule
// "fewer than 3 observations or fewer than 3 inliers -> the point becomes (0,0,0)" (which keeps it out of the next Optimize).
//
// One lane per point.  The reference draws >= 100 random pairs out of the point's K observations with std::mt19937; with
// K ~ 6-10 that visits every pair many times over, so the GPU enumerates the pairs instead (all of them up to 300, a
// deterministic strided subset beyond) -- same best minimal model, no random stream.  Local optimisation follows
// ransac.h:341-407 with the random subsets replaced by rotating windows over the inlier list.  The answer every caller
// consumes is the final least-squares optimum over the inlier set, which does not depend on those choices; parity with the
// oracle (which replays the reference's random streams) is asserted on the points and on the zeroing decisions.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include "ba_handle.h"

namespace ssfm {

constexpr int TRI_TOP = 3;         // minimal models that seed a local optimisation
constexpr int TRI_MAXW = 4;        // 64-bit words of inlier masks -> up to 256 observations per point

struct TriCtx {
    const double* cam; const double* rot; const double2* obs_xy; const int* obs_cam; int j0, n; double f, thr;
};
__device__ __forceinline__ double tri_err(const TriCtx& c, int i, const double* X) {      // EvaluateModelOnPoint
    const int cm = c.obs_cam[c.j0 + i]; const double* R = c.rot + 27 * cm; const double* t = c.cam + 6 * cm;
    const double px = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], py = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
                 pz = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    if (pz < 0) return 1.79769313486231570e308;
    const double2 o = c.obs_xy[c.j0 + i];
    const double r0 = c.f * px / pz - o.x, r1 = c.f * py / pz - o.y;
    return r0 * r0 + r1 * r1;
}
__device__ double tri_score(const TriCtx& c, const double* X) { double s = 0; for (int i = 0; i < c.n; i++) s += fmin(tri_err(c, i, X), c.thr); return s; }
__device__ int tri_inliers(const TriCtx& c, const double* X, double th, unsigned long long* mask) {
    int cnt = 0; for (int w = 0; w < TRI_MAXW; w++) mask[w] = 0ull;
    for (int i = 0; i < c.n; i++) if (tri_err(c, i, X) < th) { mask[i >> 6] |= 1ull << (i & 63); cnt++; }
    return cnt;
}
// smallest eigenvector of the symmetric 4x4 B = A^T A (cyclic Jacobi) = DLT solution (src/triangulation_estimator.cpp:65-86)
__device__ void dlt_from_normal(double B[4][4], double* X) {
    double V[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0; for (int p = 0; p < 3; p++) for (int q = p + 1; q < 4; q++) off += B[p][q] * B[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; p++) for (int q = p + 1; q < 4; q++) {
            if (B[p][q] == 0.0) continue;
            const double th = (B[q][q] - B[p][p]) / (2 * B[p][q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0)), cs = 1 / sqrt(t * t + 1), sn = t * cs;
            for (int k = 0; k < 4; k++) { const double x = B[k][p], y = B[k][q]; B[k][p] = cs * x - sn * y; B[k][q] = sn * x + cs * y; }
            for (int k = 0; k < 4; k++) { const double x = B[p][k], y = B[q][k]; B[p][k] = cs * x - sn * y; B[q][k] = sn * x + cs * y; }
            for (int k = 0; k < 4; k++) { const double x = V[k][p], y = V[k][q]; V[k][p] = cs * x - sn * y; V[k][q] = sn * x + cs * y; }
        }
    }
    int best = 0; for (int k = 1; k < 4; k++) if (B[k][k] < B[best][best]) best = k;
    X[0] = V[0][best] / V[3][best]; X[1] = V[1][best] / V[3][best]; X[2] = V[2][best] / V[3][best];
}
__device__ __forceinline__ void dlt_add_obs(const TriCtx& c, int i, double B[4][4]) {
    const int cm = c.obs_cam[c.j0 + i]; const double* R = c.rot + 27 * cm; const double* t = c.cam + 6 * cm;
    const double2 o = c.obs_xy[c.j0 + i];
    const double p0 = o.x / c.f, p1 = o.y / c.f;
    const double r2[4] = {R[6], R[7], R[8], t[2]};
    const double a0[4] = {r2[0] * p0 - R[0], r2[1] * p0 - R[1], r2[2] * p0 - R[2], r2[3] * p0 - t[0]};
    const double a1[4] = {r2[0] * p1 - R[3], r2[1] * p1 - R[4], r2[2] * p1 - R[5], r2[3] * p1 - t[1]};
    for (int u = 0; u < 4; u++) for (int v = 0; v < 4; v++) B[u][v] += a0[u] * a0[v] + a1[u] * a1[v];
}
// point-only least squares over the observations in mask (TriangulationEstimator::LeastSquares with Ceres' LM rules)
__device__ void tri_lsq(const TriCtx& c, const unsigned long long* mask, double* X) {
    double scale[3] = {1, 1, 1}, A[6], g[3], x_cost = 0;
    auto linearize = [&](const double* x) {
        for (int k = 0; k < 6; k++) A[k] = 0; g[0] = g[1] = g[2] = 0; x_cost = 0;
        for (int i = 0; i < c.n; i++) {
            if (!((mask[i >> 6] >> (i & 63)) & 1ull)) continue;
            const int cm = c.obs_cam[c.j0 + i]; const double* R = c.rot + 27 * cm; const double* t = c.cam + 6 * cm;
            const double px = R[0] * x[0] + R[1] * x[1] + R[2] * x[2] + t[0], py = R[3] * x[0] + R[4] * x[1] + R[5] * x[2] + t[1],
                         pz = R[6] * x[0] + R[7] * x[1] + R[8] * x[2] + t[2];
            const double iz = 1.0 / pz, xp = px * iz, yp = py * iz, a = c.f * iz;
            const double2 o = c.obs_xy[c.j0 + i];
            const double r0 = c.f * xp - o.x, r1 = c.f * yp - o.y;
            double J0[3], J1[3];
            for (int k = 0; k < 3; k++) { J0[k] = (a * R[k] - a * xp * R[6 + k]) * scale[k]; J1[k] = (a * R[3 + k] - a * yp * R[6 + k]) * scale[k]; }
            A[0] += J0[0] * J0[0] + J1[0] * J1[0]; A[1] += J0[0] * J0[1] + J1[0] * J1[1]; A[2] += J0[0] * J0[2] + J1[0] * J1[2];
            A[3] += J0[1] * J0[1] + J1[1] * J1[1]; A[4] += J0[1] * J0[2] + J1[1] * J1[2]; A[5] += J0[2] * J0[2] + J1[2] * J1[2];
            for (int k = 0; k < 3; k++) g[k] += J0[k] * r0 + J1[k] * r1;
            x_cost += 0.5 * (r0 * r0 + r1 * r1);
        }
    };
    auto cost_at = [&](const double* x) {
        double s = 0;
        for (int i = 0; i < c.n; i++) {
            if (!((mask[i >> 6] >> (i & 63)) & 1ull)) continue;
            const int cm = c.obs_cam[c.j0 + i]; const double* R = c.rot + 27 * cm; const double* t = c.cam + 6 * cm;
            const double px = R[0] * x[0] + R[1] * x[1] + R[2] * x[2] + t[0], py = R[3] * x[0] + R[4] * x[1] + R[5] * x[2] + t[1],
                         pz = R[6] * x[0] + R[7] * x[1] + R[8] * x[2] + t[2];
            const double2 o = c.obs_xy[c.j0 + i];
            const double r0 = c.f * px / pz - o.x, r1 = c.f * py / pz - o.y;
            s += 0.5 * (r0 * r0 + r1 * r1);
        }
        return s;
    };
    double x[3] = {X[0], X[1], X[2]};
    linearize(x);
    if (!isfinite(x_cost)) return;
    scale[0] = 1.0 / (1.0 + sqrt(A[0])); scale[1] = 1.0 / (1.0 + sqrt(A[3])); scale[2] = 1.0 / (1.0 + sqrt(A[5]));
    A[0] *= scale[0] * scale[0]; A[1] *= scale[0] * scale[1]; A[2] *= scale[0] * scale[2]; A[3] *= scale[1] * scale[1]; A[4] *= scale[1] * scale[2]; A[5] *= scale[2] * scale[2];
    g[0] *= scale[0]; g[1] *= scale[1]; g[2] *= scale[2];
    double radius = 1e4, decrease = 2.0, x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    int iteration = 0, invalid = 0; bool last_ok = true;
    while (true) {
        if (iteration >= 200) break;
        const double gmax = fmax(fabs(g[0] / scale[0]), fmax(fabs(g[1] / scale[1]), fabs(g[2] / scale[2])));
        if (last_ok && gmax <= 1e-10) break;
        if (radius <= 1e-32) break;
        iteration++;
        double Ad[6] = {A[0], A[1], A[2], A[3], A[4], A[5]};
        Ad[0] += fmin(fmax(A[0], 1e-6), 1e32) / radius; Ad[3] += fmin(fmax(A[3], 1e-6), 1e32) / radius; Ad[5] += fmin(fmax(A[5], 1e-6), 1e32) / radius;
        double Ai[6]; sym3_inverse(Ad, Ai);
        const double st[3] = {-(Ai[0] * g[0] + Ai[1] * g[1] + Ai[2] * g[2]), -(Ai[1] * g[0] + Ai[3] * g[1] + Ai[4] * g[2]), -(Ai[2] * g[0] + Ai[4] * g[1] + Ai[5] * g[2])};
        const double sAs = A[0] * st[0] * st[0] + A[3] * st[1] * st[1] + A[5] * st[2] * st[2] + 2 * (A[1] * st[0] * st[1] + A[2] * st[0] * st[2] + A[4] * st[1] * st[2]);
        const double model = -((g[0] * st[0] + g[1] * st[1] + g[2] * st[2]) + 0.5 * sAs);
        if (!(model > 0.0) || !isfinite(model)) { if (++invalid >= 10) break; radius /= decrease; decrease *= 2.0; last_ok = false; continue; }
        invalid = 0;
        const double xc[3] = {x[0] + st[0] * scale[0], x[1] + st[1] * scale[1], x[2] + st[2] * scale[2]};
        double cand = cost_at(xc); if (!isfinite(cand)) cand = 1.79e308;
        const double sn = sqrt((xc[0] - x[0]) * (xc[0] - x[0]) + (xc[1] - x[1]) * (xc[1] - x[1]) + (xc[2] - x[2]) * (xc[2] - x[2]));
        if (sn <= 1e-8 * (x_norm + 1e-8)) break;
        const double change = x_cost - cand;
        if (fabs(change) <= 1e-6 * x_cost) break;
        const double rho = (cand >= 1.79e308) ? -1.79e308 : change / model;
        if (rho > 1e-3) {
            x[0] = xc[0]; x[1] = xc[1]; x[2] = xc[2]; x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            linearize(x);
            { const double t3 = 2.0 * rho - 1.0; radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3)); } decrease = 2.0; last_ok = true;
        } else { radius /= decrease; decrease *= 2.0; last_ok = false; }
    }
    X[0] = x[0]; X[1] = x[1]; X[2] = x[2];
}
__device__ __forceinline__ int mask_nth(const unsigned long long* mask, int n, int k) {       // index of the k-th set bit
    int seen = 0;
    for (int i = 0; i < n; i++) if ((mask[i >> 6] >> (i & 63)) & 1ull) { if (seen == k) return i; seen++; }
    return -1;
}
// LeastSquaresFit (ransac.h:409-420): inliers at `th`, at most 14 of them, point-only LM
__device__ void tri_lsq_fit(const TriCtx& c, double th, double* X) {
    unsigned long long m[TRI_MAXW]; const int cnt = tri_inliers(c, X, th, m);
    if (cnt < 2) return;
    if (cnt > 14) { int kept = 0; for (int i = 0; i < c.n; i++) if ((m[i >> 6] >> (i & 63)) & 1ull) { if (kept >= 14) m[i >> 6] &= ~(1ull << (i & 63)); kept++; } }
    tri_lsq(c, m, X);
}

__global__ void __launch_bounds__(64)
k_retriangulate(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ focal,
                const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam, const int* __restrict__ pt_start, int nP,
                double sq_thresh, double* __restrict__ pts, int* __restrict__ num_inliers) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nP) return;
    TriCtx c{cam, rot, obs_xy, obs_cam, pt_start[p], min(pt_start[p + 1] - pt_start[p], 64 * TRI_MAXW), focal[0], sq_thresh};
    double best[3] = {0, 0, 0}; double best_score = 1.79769313486231570e308;
    int nin = 0;
    auto consider = [&](const double* m) { const double sc = tri_score(c, m); if (sc < best_score) { best_score = sc; best[0] = m[0]; best[1] = m[1]; best[2] = m[2]; } };
    if (c.n >= 3) {                                                           // src/sfm.cpp:173
        // ---- minimal models: every pair (strided subset when there are more than ~300); the TRI_TOP best ones seed the
        // local optimisation, like the successive "new best" models of the reference's random sequence do
        double top[TRI_TOP][3], top_score[TRI_TOP];
        for (int t = 0; t < TRI_TOP; t++) top_score[t] = 1.79769313486231570e308;
        const int npairs = c.n * (c.n - 1) / 2, stride = max(1, npairs / 300);
        int pi = 0;
        for (int a = 0; a < c.n - 1; a++) for (int b = a + 1; b < c.n; b++, pi++) {
            if (pi % stride) continue;
            double B[4][4]; for (int u = 0; u < 4; u++) for (int v = 0; v < 4; v++) B[u][v] = 0.0;
            dlt_add_obs(c, a, B); dlt_add_obs(c, b, B);
            double X[3]; dlt_from_normal(B, X);
            if (!isfinite(X[0] + X[1] + X[2])) continue;
            double sc = tri_score(c, X);
            for (int t = 0; t < TRI_TOP; t++) if (sc < top_score[t]) {           // insertion into the sorted shortlist
                for (int k = 0; k < 3; k++) { const double tmp = top[t][k]; top[t][k] = X[k]; X[k] = tmp; }
                const double tmp = top_score[t]; top_score[t] = sc; sc = tmp;
            }
        }
        if (top_score[0] < 1.79e308) {
            best_score = top_score[0]; best[0] = top[0][0]; best[1] = top[0][1]; best[2] = top[0][2];
            // ---- local optimisation (ransac.h:341-407; num_lo_steps 10, num_lsq_iterations 4, threshold multiplier sqrt 2)
            const double tm = 1.4142135623730951;
            for (int t = 0; t < TRI_TOP; t++) {
                if (!(top_score[t] < 1.79e308)) break;
                double m_init[3] = {top[t][0], top[t][1], top[t][2]};
                tri_lsq_fit(c, c.thr * tm, m_init);
                consider(m_init);
                unsigned long long base[TRI_MAXW]; const int nbase = tri_inliers(c, m_init, c.thr * tm, base);
                const int nonmin = max(2, min(6, nbase / 2));
                if (nbase < nonmin) continue;
                for (int r = 0; r < 10; r++) {
                    double B[4][4]; for (int u = 0; u < 4; u++) for (int v = 0; v < 4; v++) B[u][v] = 0.0;
                    for (int k = 0; k < nonmin; k++) dlt_add_obs(c, mask_nth(base, c.n, (r + 3 * t + k * max(1, nbase / nonmin)) % nbase), B);
                    double m[3]; dlt_from_normal(B, m);
                    if (!isfinite(m[0] + m[1] + m[2])) continue;
                    consider(m);
                    tri_lsq_fit(c, c.thr, m);
                    double th = tm * c.thr; const double upd = (tm - 1.0) * c.thr / 3.0;
                    for (int i = 0; i < 4; i++) { tri_lsq_fit(c, th, m); consider(m); th -= upd; }
                }
            }
            // ---- final least squares on the inliers of the best model (ransac.h:256-270)
            unsigned long long inl[TRI_MAXW]; nin = tri_inliers(c, best, c.thr, inl);
            double refined[3] = {best[0], best[1], best[2]};
            if (nin > 0) tri_lsq(c, inl, refined);
            const double sc = tri_score(c, refined);
            if (sc < best_score) { best_score = sc; best[0] = refined[0]; best[1] = refined[1]; best[2] = refined[2]; nin = tri_inliers(c, best, c.thr, inl); }
        }
    }
    if (nin < 3) { best[0] = best[1] = best[2] = 0.0; }                        // src/sfm.cpp:186
    pts[3 * p] = best[0]; pts[3 * p + 1] = best[1]; pts[3 * p + 2] = best[2];
    if (num_inliers) num_inliers[p] = nin;
}


// =====================================================================================================================================
// Reference-trace mode (the default): the per-point LocallyOptimizedMSAC of SfM::Retriangulate replayed draw for draw.
//
//   LocallyOptimizedMSAC<Point, ..., TriangulationEstimator>::EstimateModel   include/RansacLib/ransac.h:128-275, options of src/sfm.cpp:175-177:
//                                                                              squared_inlier_threshold_ 4, final_least_squares_, everything else
//                                                                              a LORansacOptions default (10 LO steps, 4 lsq iterations, seed 0)
//   LocalOptimization / LeastSquaresFit / GetInliers / ScoreModel             ransac.h:277-428
//   UniformSampling, RandomShuffleAndResize, NumRequiredIterations            sampling.h:46-135, utils.h:48-140
//   TriangulationEstimator (min_sample_size 2, non_minimal_sample_size 2)     src/triangulation_estimator.cpp:46-127, triangulation_estimator.h:23-27
//
// Every point runs both std::mt19937 streams from seed 0 (ransac.h:142-144), so nothing random is per point:
//  * the sampler's sequence of observation pairs depends only on the NUMBER of observations n of a point.  The host draws it with
//    libstdc++'s own std::mt19937 + std::uniform_int_distribution (what a build of the reference links) once per distinct n, for all
//    max_num_iterations_ = 10000 iterations, and so does utils::NumRequiredIterations for every possible inlier count of that n;
//  * the local optimisation's stream is consumed at a point-dependent rate (the shuffled lists have point-dependent lengths), so the host
//    uploads the raw 32-bit outputs of mt19937(0) once and every lane walks them with its own cursor, reducing each word with Lemire's
//    method as libstdc++'s uniform_int_distribution does (lemire_accept below).  A lane that runs off the table raises a flag and the
//    host repeats the launch with a longer table.
// One lane per point, all lists in a global scratch array (3 n ints per point).
//
// BIT-EXACT ARITHMETIC.  RANSAC draws the same observation pair many times, in both orders, and ransac.h:183-186 starts a local
// optimisation (which consumes random numbers) whenever a score is lower than the best one so far -- by however little.  Which of two
// mathematically equal scores is lower is decided by their last bits, so a replay must round every operation of the minimal solver and
// of the scoring the way the CPU does.  This translation unit is therefore compiled with -ffp-contract=off (no fused multiply-adds), uses
// only IEEE +, -, *, /, sqrt on the device, evaluates every sum in one fixed documented order, and takes everything transcendental
// (the cameras' sin / cos, the iteration-count table) from the host's libm.  tests/test_retriangulate_gpu.py compares DLT points,
// scores and least-squares fits with the oracle's (compiled the same way, oracle/Makefile) for EQUALITY.
// =====================================================================================================================================
constexpr int TC = 32;             // doubles per camera record (tri_camera_record)
constexpr int TRI_MAX_IT = 10000;  // max_num_iterations_ (ransac.h:52)
constexpr double TRI_DMAX = 1.79769313486231570815e308;

// Per-camera record, computed on the HOST (libm's sin / cos):
//  [0..11]  P = [so3exp(r) | t], row-major 3x4 (Pose::P, src/sfm_types.cpp:14-19): EvaluateModelOnPoint and the DLT rows use it
//  [12]     1 when theta^2 > DBL_EPSILON (Rodrigues branch of ceres::AngleAxisRotatePoint), else 0
//  [13..15] cos theta, sin theta, 1 - cos theta
//  [16..18] w = r * (1 / theta)                      (Taylor branch: r itself)
//  [19..27] J = d(AngleAxisRotatePoint(r, X)) / dX as the reference's Jets accumulate it (src/triangulation_estimator.cpp:30):
//           J[i][k] = (c d_ik + [w]x_ik s) + w_i (w_k (1 - c)); it does not depend on X        (Taylor branch: I + [r]x)
// sin / cos come from ONE glibc sincos() call per angle: that is what a GCC build of the reference (and of the oracle) executes wherever both
// functions of an angle are needed, and its sine is not always the bit pattern of sin() (e.g. at 0.83775804095727813).
inline void tri_camera_record(const double* cam, double* T) {
    const double* t = cam; const double* r = cam + 3;
    const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    {   // so3exp (src/so3.cpp:16-23): I + sin K + (1 - cos) K K with K = skew(r / theta); identity below 1e-10
        const double theta = std::sqrt(th2);
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (!(theta < 1e-10)) {
            const double kx = r[0] / theta, ky = r[1] / theta, kz = r[2] / theta;
            double s, cs; ::sincos(theta, &s, &cs);
            const double omc = 1.0 - cs;
            R[0] = 1.0 + omc * (-(ky * ky) - kz * kz); R[1] = -s * kz + omc * (kx * ky);          R[2] = s * ky + omc * (kx * kz);
            R[3] = s * kz + omc * (kx * ky);           R[4] = 1.0 + omc * (-(kx * kx) - kz * kz); R[5] = -s * kx + omc * (ky * kz);
            R[6] = -s * ky + omc * (kx * kz);          R[7] = s * kx + omc * (ky * kz);           R[8] = 1.0 + omc * (-(kx * kx) - ky * ky);
        }
        for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) T[4 * a + b] = R[3 * a + b]; T[4 * a + 3] = t[a]; }
    }
    double* J = T + 19;
    if (th2 > DBL_EPSILON) {
        const double th = std::sqrt(th2), inv = 1.0 / th;
        double s, c; ::sincos(th, &s, &c);
        const double omc = 1.0 - c;
        const double w[3] = {r[0] * inv, r[1] * inv, r[2] * inv};
        T[12] = 1.0; T[13] = c; T[14] = s; T[15] = omc; T[16] = w[0]; T[17] = w[1]; T[18] = w[2];
        const double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) J[3 * i + k] = (((i == k) ? c : 0.0) + wx[3 * i + k] * s) + w[i] * (w[k] * omc);
    } else {
        T[12] = 0.0; T[13] = 1.0; T[14] = 0.0; T[15] = 0.0; T[16] = r[0]; T[17] = r[1]; T[18] = r[2];
        const double wx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
        for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) J[3 * i + k] = ((i == k) ? 1.0 : 0.0) + wx[3 * i + k];
    }
    for (int k = 28; k < TC; k++) T[k] = 0.0;
}

// r05ao: tt_lsq's model-cost pass re-evaluated every row's residual and Jacobian at the point the sweep before it had just linearised at (the same bits: that is why it
// could) -- a third of the per-row work of an LM iteration.  The sweep now leaves r and J of its first TT_NCACHE rows in LDS (8 doubles per row, lane-interleaved: no bank
// conflicts) and the model-cost pass reads them back; rows beyond are still re-evaluated.  5 rows x 8 x 64 lanes x 8 B = 20 KB per wave: seven waves per CU, which the
// 1563 waves of 100 000 points need (6.1 per CU) -- six rows (24.6 KB, six waves per CU) would push the last waves into a second round.
constexpr int TT_NCACHE = 5;
struct TtPoint {                    // one point's observations (cameras ascending)
    const double* ct; const double2* oxy; const int* ocam; int j0, n; double f;
    double* cache = nullptr;        // LDS, this lane's first element (stride 64), or null: re-evaluate
    double* ecache = nullptr;       // LDS [TT_NERR][64]: the errors tt_score saw last (GetInliers at the same model reads them back), or null
};
constexpr int TT_NERR = 5;          // (20 KB + 2.5 KB per wave: still seven waves per CU)
// EvaluateModelOnPoint (src/triangulation_estimator.cpp:46-54): P X + t left to right, behind the camera -> DBL_MAX
__device__ __forceinline__ double tt_err(const TtPoint& c, int i, const double* X) {
    const double* T = c.ct + (size_t)TC * c.ocam[c.j0 + i];
    const double px = T[0] * X[0] + T[1] * X[1] + T[2] * X[2] + T[3], py = T[4] * X[0] + T[5] * X[1] + T[6] * X[2] + T[7],
                 pz = T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11];
    if (pz < 0) return TRI_DMAX;
    const double2 o = c.oxy[c.j0 + i];
    const double r0 = c.f * px / pz - o.x, r1 = c.f * py / pz - o.y;
    return r0 * r0 + r1 * r1;
}
// ScoreModel (ransac.h:295-303): sum of std::min(error, threshold) in observation order (std::min(a, b) = b < a ? b : a keeps a NaN error)
__device__ double tt_score(const TtPoint& c, const double* X, double thr) {
    double s = 0.0;
    for (int i = 0; i < c.n; i++) { const double e = tt_err(c, i, X); if (c.ecache && i < TT_NERR) c.ecache[i * 64] = e; s += (thr < e) ? thr : e; }
    return s;
}
// GetInliers (ransac.h:311-336)
// scored: tt_score's last call was at this very X (the caller knows): the errors of the first TT_NERR observations come from LDS, the same bits
__device__ int tt_inliers(const TtPoint& c, const double* X, double th, int* list, bool scored = false) {
    int cnt = 0;
    for (int i = 0; i < c.n; i++) { const double e = (scored && c.ecache && i < TT_NERR) ? c.ecache[i * 64] : tt_err(c, i, X); if (e < th) list[cnt++] = i; }
    return cnt;
}
// NonMinimalSolver (src/triangulation_estimator.cpp:65-86): rows (P.row(2) x - P.row(0), P.row(2) y - P.row(1)) per sampled observation, in
// sample order; X = last right singular vector.  The SVD is a one-sided Jacobi iteration on the columns (pairs (0,1) (0,2) (0,3) (1,2) (1,3)
// (2,3) per sweep, until the largest normalised column product of a sweep is below 1e-15, at most 60 sweeps); the column of least norm
// wins (first of equals).  MAXR = row capacity (4 for the minimal sample, 12 for the LO samples of up to 6 observations).
template <int MAXR>
__device__ void tt_dlt(const TtPoint& c, const int* sample, int ns, int navail /* entries of sample that exist; beyond: observation 0 */, double* X) {
    double A[MAXR][4], V[4][4];
    const int m = 2 * ns;
#pragma unroll
    for (int nn = 0; nn < MAXR / 2; nn++) {
        if (nn < ns) {
            const int oi = (nn < navail) ? sample[nn] : 0;              // std::vector::resize pads a short sample with zeros (utils.h:69-73)
            const double* T = c.ct + (size_t)TC * c.ocam[c.j0 + oi];
            const double2 o = c.oxy[c.j0 + oi];
            const double p0 = o.x / c.f, p1 = o.y / c.f;
#pragma unroll
            for (int k = 0; k < 4; k++) { A[2 * nn][k] = T[8 + k] * p0 - T[k]; A[2 * nn + 1][k] = T[8 + k] * p1 - T[4 + k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) { A[2 * nn][k] = 0.0; A[2 * nn + 1][k] = 0.0; }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int i = 0; i < MAXR; i++) if (i < m) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
                if (gamma != 0.0) {
                    const double rel = fabs(gamma) / sqrt(alpha * beta + 1e-300);
                    off = (off < rel) ? rel : off;
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta)), cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
                    for (int i = 0; i < MAXR; i++) if (i < m) { const double ap = A[i][p], aq = A[i][q]; A[i][p] = cs * ap - sn * aq; A[i][q] = sn * ap + cs * aq; }
#pragma unroll
                    for (int i = 0; i < 4; i++) { const double vp = V[i][p], vq = V[i][q]; V[i][p] = cs * vp - sn * vq; V[i][q] = sn * vp + cs * vq; }
                }
            }
        if (off < 1e-15) break;
    }
    double bn = 1e300, Xh[4] = {V[0][0], V[1][0], V[2][0], V[3][0]};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double nq = 0.0;
#pragma unroll
        for (int i = 0; i < MAXR; i++) if (i < m) nq += A[i][q] * A[i][q];
        if (nq < bn) { bn = nq; Xh[0] = V[0][q]; Xh[1] = V[1][q]; Xh[2] = V[2][q]; Xh[3] = V[3][q]; }
    }
    X[0] = Xh[0] / Xh[3]; X[1] = Xh[1] / Xh[3]; X[2] = Xh[2] / Xh[3];
}

// TriangulationError (src/triangulation_estimator.cpp:18-44) at X for observation i: residual (2) and, when J0 != nullptr, its 2x3 Jacobian
// as Ceres' Jets produce it.  AngleAxisRotatePoint: PX = (X c + (w x X) s) + w ((w.X)(1 - c)); its derivative is the camera's constant J.
// proj = PX_xy / PX_z:  value a = PX_x g, derivative (dPX_x - a dPX_z) g  with g = 1 / PX_z;  residual = f proj - x.
// JET = false is the cost-only evaluation on plain doubles.
template <bool JET>
__device__ __forceinline__ void tt_residual(const TtPoint& c, int i, const double* X, double* res, double* J0, double* J1) {
    const double* T = c.ct + (size_t)TC * c.ocam[c.j0 + i];
    double PX[3];
    if (T[12] != 0.0) {
        const double cs = T[13], sn = T[14], omc = T[15], w0 = T[16], w1 = T[17], w2 = T[18];
        const double x0 = w1 * X[2] - w2 * X[1], x1 = w2 * X[0] - w0 * X[2], x2 = w0 * X[1] - w1 * X[0];
        const double tmp = (w0 * X[0] + w1 * X[1] + w2 * X[2]) * omc;
        PX[0] = X[0] * cs + x0 * sn + w0 * tmp; PX[1] = X[1] * cs + x1 * sn + w1 * tmp; PX[2] = X[2] * cs + x2 * sn + w2 * tmp;
    } else {
        const double a0 = T[16], a1 = T[17], a2 = T[18];
        PX[0] = X[0] + (a1 * X[2] - a2 * X[1]); PX[1] = X[1] + (a2 * X[0] - a0 * X[2]); PX[2] = X[2] + (a0 * X[1] - a1 * X[0]);
    }
    PX[0] = PX[0] + T[3]; PX[1] = PX[1] + T[7]; PX[2] = PX[2] + T[11];
    const double2 o = c.oxy[c.j0 + i];
    if (JET) {
        // evaluated on Jets: ceres::Jet's operator/ multiplies by the reciprocal of the denominator's value
        const double g = 1.0 / PX[2], a = PX[0] * g, b = PX[1] * g;
        res[0] = a * c.f - o.x; res[1] = b * c.f - o.y;
        const double* J = T + 19;
#pragma unroll
        for (int k = 0; k < 3; k++) { J0[k] = ((J[k] - a * J[6 + k]) * g) * c.f; J1[k] = ((J[3 + k] - b * J[6 + k]) * g) * c.f; }
    } else {
        // evaluated on doubles (a cost-only evaluation): a true division
        res[0] = c.f * (PX[0] / PX[2]) - o.x; res[1] = c.f * (PX[1] / PX[2]) - o.y;
    }
}

// TriangulationEstimator::LeastSquares (src/triangulation_estimator.cpp:88-127) on the observations list[0..cnt): Ceres 2.2's
// TrustRegionMinimizer + LevenbergMarquardtStrategy + DENSE_NORMAL_CHOLESKY on the three point coordinates, 200 iterations, 10 consecutive
// invalid steps (:119-123), defaults otherwise (Jacobi scaling, radius 1e4, function / gradient / parameter tolerance 1e-6 / 1e-10 / 1e-8).
// Every sum runs over the residual rows in list order, two rows per observation.  Returns the number of iterations.
__device__ int tt_lsq(const TtPoint& c, const int* list, int cnt, double* X) {
    double x[3] = {X[0] + 0.0, X[1] + 0.0, X[2] + 0.0};
    double g[3], colsq[3], A[3][3], b[3], scale[3] = {1.0, 1.0, 1.0}, x_cost = 0.0;
    // one sweep over the rows at x: cost, gradient J^T r, squared column norms; with_normal: also Js^T Js and Js^T r of the scaled Jacobian
    auto sweep = [&](const double* xx, bool with_normal, bool reuse = false) -> bool {      // reuse: the cache holds the rows at xx (the sweep just before was at xx)
        double cc = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { g[k] = 0.0; colsq[k] = 0.0; b[k] = 0.0; A[k][0] = 0.0; A[k][1] = 0.0; A[k][2] = 0.0; }
        for (int q = 0; q < cnt; q++) {
            double r[2], J[2][3];
            if (reuse && c.cache && q < TT_NCACHE) {
                const double* w = c.cache + (size_t)q * 8 * 64;
                r[0] = w[0]; r[1] = w[64];
#pragma unroll
                for (int k = 0; k < 3; k++) { J[0][k] = w[(2 + k) * 64]; J[1][k] = w[(5 + k) * 64]; }
            } else {
                tt_residual<true>(c, list[q], xx, r, J[0], J[1]);
                if (c.cache && q < TT_NCACHE) {
                    double* w = c.cache + (size_t)q * 8 * 64;
                    w[0] = r[0]; w[64] = r[1];
#pragma unroll
                    for (int k = 0; k < 3; k++) { w[(2 + k) * 64] = J[0][k]; w[(5 + k) * 64] = J[1][k]; }
                }
            }
#pragma unroll
            for (int a = 0; a < 2; a++) {
#pragma unroll
                for (int k = 0; k < 3; k++) { g[k] += J[a][k] * r[a]; colsq[k] += J[a][k] * J[a][k]; }
                cc += 0.5 * r[a] * r[a];
                if (with_normal) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const double ja = J[a][k] * scale[k];
                        b[k] += ja * r[a];
#pragma unroll
                        for (int l = 0; l < 3; l++) A[k][l] += ja * J[a][l] * scale[l];
                    }
                }
            }
        }
        x_cost = cc;
        return isfinite(cc);
    };
    auto cost_at = [&](const double* xx) {
        double cc = 0.0;
        for (int q = 0; q < cnt; q++) { double r[2]; tt_residual<false>(c, list[q], xx, r, nullptr, nullptr); cc += 0.5 * (r[0] * r[0] + r[1] * r[1]); }
        return cc;
    };
    double x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (!sweep(x, false)) return 0;                                                           // evaluation failure at the start: FAILURE, X untouched
#pragma unroll
    for (int k = 0; k < 3; k++) scale[k] = 1.0 / (1.0 + sqrt(colsq[k]));                      // Jacobi scaling from the iteration-0 Jacobian
    auto gradient_max = [&]() { double m = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { const double cand = x[k] + (-g[k]); m = fmax(m, fabs(x[k] - cand)); }
        return m; };
    double gmax = gradient_max();
    double best[3] = {x[0], x[1], x[2]}, minimum_cost = x_cost;
    (void)sweep(x, true, true);                                                               // the same rows again (from the cache), now with the scaled normal equations
    double radius = 1e4, decrease = 2.0, diag[3] = {0, 0, 0};
    bool reuse_diagonal = false, last_ok = true;
    int iteration = 0, invalid = 0;
    while (true) {
        if (iteration >= 200) break;
        if (last_ok && gmax <= 1e-10) break;
        if (radius <= 1e-32) break;
        iteration++;
        if (!reuse_diagonal) {
#pragma unroll
            for (int k = 0; k < 3; k++) diag[k] = fmin(fmax(colsq[k] * (scale[k] * scale[k]), 1e-6), 1e32);
        }
        double D[3];
#pragma unroll
        for (int k = 0; k < 3; k++) D[k] = sqrt(diag[k] / radius);
        reuse_diagonal = true;
        // (Js^T Js + D^2) y = Js^T r by Cholesky; a non-positive pivot is a failed linear solve
        double M[3][3], L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, z[3], y[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; k++) { M[k][0] = A[k][0]; M[k][1] = A[k][1]; M[k][2] = A[k][2]; M[k][k] += D[k] * D[k]; }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double d = M[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
            if (!(d > 0)) { ok = false; break; }
            L[j][j] = sqrt(d);
#pragma unroll
            for (int i = j + 1; i < 3; i++) { double v = M[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
                L[i][j] = v / L[j][j]; }
        }
        bool valid = false; double model = 0.0, step[3] = {0, 0, 0};
        if (ok) {
#pragma unroll
            for (int i = 0; i < 3; i++) { double v = b[i];
#pragma unroll
                for (int k = 0; k < i; k++) v -= L[i][k] * z[k];
                z[i] = v / L[i][i]; }
#pragma unroll
            for (int i = 2; i >= 0; i--) { double v = z[i];
#pragma unroll
                for (int k = i + 1; k < 3; k++) v -= L[k][i] * y[k];
                y[i] = v / L[i][i]; }
            bool fin = true;
#pragma unroll
            for (int k = 0; k < 3; k++) { step[k] = -y[k]; fin = fin && isfinite(step[k]); }
            if (fin) {
                // model cost change -(Js step)^T (r + Js step / 2), row by row (the rows are re-evaluated: same bits as the stored Jacobian)
                double acc = 0.0;
                for (int q = 0; q < cnt; q++) {
                    double r[2], J[2][3];
                    if (c.cache && q < TT_NCACHE) {                     // what the last sweep at x stored (x only changes through a sweep)
                        const double* w = c.cache + (size_t)q * 8 * 64;
                        r[0] = w[0]; r[1] = w[64];
#pragma unroll
                        for (int k = 0; k < 3; k++) { J[0][k] = w[(2 + k) * 64]; J[1][k] = w[(5 + k) * 64]; }
                    } else tt_residual<true>(c, list[q], x, r, J[0], J[1]);
#pragma unroll
                    for (int a = 0; a < 2; a++) { double m = 0.0;
#pragma unroll
                        for (int k = 0; k < 3; k++) m += J[a][k] * scale[k] * step[k];
                        acc += m * (r[a] + 0.5 * m); }
                }
                model = -acc; valid = model > 0.0;
            }
        }
        if (!valid) {
            if (++invalid >= 10) break;
            radius /= decrease; decrease *= 2.0; reuse_diagonal = true; last_ok = false; continue;
        }
        invalid = 0;
        double cand[3];
#pragma unroll
        for (int k = 0; k < 3; k++) cand[k] = x[k] + step[k] * scale[k];
        double cand_cost = cost_at(cand);
        if (!isfinite(cand_cost)) cand_cost = TRI_DMAX;
        double sn2 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) sn2 += (x[k] - cand[k]) * (x[k] - cand[k]);
        if (sqrt(sn2) <= 1e-8 * (x_norm + 1e-8)) break;                                       // parameter tolerance: the candidate is not taken
        const double change = x_cost - cand_cost;
        if (fabs(change) <= 1e-6 * x_cost) break;                                             // function tolerance: likewise
        const double rel = (cand_cost >= TRI_DMAX) ? -TRI_DMAX : change / model;
        if (rel > 1e-3) {
            x[0] = cand[0]; x[1] = cand[1]; x[2] = cand[2];
            x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            if (!sweep(x, true)) break;
            gmax = gradient_max();
            { const double t3 = 2.0 * rel - 1.0; radius = radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3); }      // the cube by multiplication, like oracle/lm.hpp: IEEE on both sides
            radius = fmin(1e16, radius);
            decrease = 2.0; reuse_diagonal = false; last_ok = true;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; best[0] = x[0]; best[1] = x[1]; best[2] = x[2]; }
        } else { radius /= decrease; decrease *= 2.0; reuse_diagonal = true; last_ok = false; }
    }
    X[0] = best[0]; X[1] = best[1]; X[2] = best[2];
    return iteration;
}

// the local optimisation's std::mt19937(0): raw outputs in W, one cursor per lane
struct TtRng { const unsigned* W; int LW; int pos; bool overflow; };
__device__ __forceinline__ int tt_uniform_int(TtRng& g, int a, int b) {                       // std::uniform_int_distribution<int>(a, b)(rng), libstdc++
    const unsigned range = (unsigned)(b - a) + 1u;
    while (true) {
        if (g.pos >= g.LW) { g.overflow = true; return a; }
        const unsigned long long prod = (unsigned long long)g.W[g.pos++] * (unsigned long long)range;
        const unsigned low = (unsigned)prod;
        if (low < range) { const unsigned thr = (0u - range) % range; if (low < thr) continue; }
        return a + (int)(unsigned)(prod >> 32);
    }
}
// utils::RandomShuffleAndResize (utils.h:48-73): Fisher-Yates over all m entries (m - 1 draws), the first `target` survive.  A swap at a
// position >= target cannot reach the survivors, so only its draw is made.
__device__ void tt_shuffle_resize(int* list, int m, int target, TtRng& g) {
    for (int i = 0; i < m - 1; i++) {
        const int idx = tt_uniform_int(g, i, m - 1);
        if (i < target) { const int t = list[i]; list[i] = list[idx]; list[idx] = t; }
    }
}
struct TtOpts { double thr, mult; int lo_steps, lsq_it, min_sample_mult, non_min_mult; unsigned min_it, lo_start; };
// Round 6 -- one local optimisation per launch.  A wave runs as many local-optimisation rounds as its busiest lane needs, and a round is ~100 RANSAC iterations of
// work: at 100k points x 6 observations 89 % of the points need two and 11 % three, so nearly every wave of 64 ran a third round with ~7 lanes active (PMC: 37.8 of 64
// lanes).  Now a lane whose point asks for another local optimisation after it has had this launch's one SAVES its loop state (below) and its point goes onto the list
// of the next launch, which starts with dense waves of parked points.  The per-point sequence of operations -- and with it every bit of the result -- is unchanged.
struct TtState { double best[3], best_score, best_min[3], best_min_score; unsigned it, max_it; int nin, lo_count, need, pos, ovf, pad; };
// LeastSquaresFit (ransac.h:409-420)
__device__ void tt_lsq_fit(const TtPoint& c, const TtOpts& o, double th, TtRng& g, int* work, double* model, bool scored = false) {
    const int ni = tt_inliers(c, model, th, work, scored);
    if (ni < 2) return;
    const int sz = min(o.min_sample_mult * 2, ni);
    tt_shuffle_resize(work, ni, sz, g);
    (void)tt_lsq(c, work, sz, model);
}
__device__ __forceinline__ void tt_update(double sc, const double* m, double* best_sc, double* best) {
    if (sc < *best_sc) { *best_sc = sc; best[0] = m[0]; best[1] = m[1]; best[2] = m[2]; }
}
// LocalOptimization (ransac.h:341-407)
__device__ void tt_local_optimization(const TtPoint& c, const TtOpts& o, TtRng& g, int* base, int* work, double* best_min, double* score_best) {
    if (2 > c.n) return;
    double m_init[3] = {best_min[0], best_min[1], best_min[2]};
    tt_lsq_fit(c, o, o.thr * o.mult, g, work, m_init);
    tt_update(tt_score(c, m_init, o.thr), m_init, score_best, best_min);
    const int nb = tt_inliers(c, m_init, o.thr * o.mult, base);
    const int nonmin = max(2, min(2 * o.non_min_mult, nb / 2));
    for (int r = 0; r < o.lo_steps; r++) {
        for (int i = 0; i < nb; i++) work[i] = base[i];
        tt_shuffle_resize(work, nb, nonmin, g);
        double m[3];
        tt_dlt<12>(c, work, min(nonmin, 6), nb, m);                                          // NonMinimalSolver always returns 1
        tt_update(tt_score(c, m, o.thr), m, score_best, best_min);
        tt_lsq_fit(c, o, o.thr, g, work, m, true);                                          // (m was scored one line up)
        double th = o.mult * o.thr; const double upd = (o.mult - 1.0) * o.thr / (double)(o.lsq_it - 1);
        for (int i = 0; i < o.lsq_it; i++) {
            tt_lsq_fit(c, o, th, g, work, m, i > 0);                                         // (from the second pass on m is what the pass before scored)
            tt_update(tt_score(c, m, o.thr), m, score_best, best_min);
            th -= upd;
        }
    }
}

template <bool CACHE>
__device__ __forceinline__ void
retriangulate_trace_body(const double* __restrict__ ct, const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
                      const int* __restrict__ pt_start, int nP, const int* __restrict__ order /* lane -> point, or null */, TtOpts o,
                      const int* __restrict__ pt_slot, const unsigned short* __restrict__ samples /* [slot][TRI_MAX_IT][2] */,
                      const int* __restrict__ req_ptr, const unsigned* __restrict__ req_it /* NumRequiredIterations per (slot, inlier count) */,
                      const unsigned* __restrict__ W, int LW, int* __restrict__ lists /* [3 * total observations] */,
                      double* __restrict__ pts, int* __restrict__ num_inliers, unsigned* __restrict__ stats, unsigned char* __restrict__ flags,
                      int* __restrict__ overflow, TtState* __restrict__ state, int resume, int lo_budget, int* __restrict__ next_list, int* __restrict__ next_count) {
    const int lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane_id >= nP) return;
    const int p = order ? order[lane_id] : lane_id;
    __shared__ double s_cache[CACHE ? (TT_NCACHE * 8 + TT_NERR) * 64 : 1];
    TtPoint c{ct, obs_xy, obs_cam, pt_start[p], pt_start[p + 1] - pt_start[p], focal[0]};
    if (CACHE) { c.cache = s_cache + threadIdx.x; c.ecache = s_cache + TT_NCACHE * 8 * 64 + threadIdx.x; }
    double best[3] = {0, 0, 0};
    int nin = 0; unsigned it = 0; int lo_count = 0;
    int* listI = lists + 3 * (size_t)c.j0; int* base = listI + c.n; int* work = base + c.n;
    bool suspended = false;
    if (c.n >= 3) {                                                           // src/sfm.cpp:173
        TtRng g{W, LW, 0, false};
        const int slot = pt_slot[p];
        const unsigned short* smp = samples + (size_t)slot * TRI_MAX_IT * 2;
        const unsigned* req = req_it + req_ptr[slot];
        double best_score = TRI_DMAX, best_min[3] = {0, 0, 0}, best_min_score = TRI_DMAX;
        unsigned max_it = TRI_MAX_IT;                                         // max(max_num_iterations_, min_num_iterations_)
        int need0 = 0;                                                        // a resumed lane comes back parked: it asked for a local optimisation when it was suspended
        if (resume) {
            const TtState& S = state[p];
            best[0] = S.best[0]; best[1] = S.best[1]; best[2] = S.best[2]; best_score = S.best_score;
            best_min[0] = S.best_min[0]; best_min[1] = S.best_min[1]; best_min[2] = S.best_min[2]; best_min_score = S.best_min_score;
            it = S.it; max_it = S.max_it; nin = S.nin; lo_count = S.lo_count; need0 = S.need; g.pos = S.pos; g.overflow = S.ovf != 0;
        }
        int budget = lo_budget;
        // EstimateModel's loop (ransac.h:154-229) as a per-lane state machine.  A lane's sequence of operations is the reference's; what changes is WHEN
        // the wave executes them: a local optimisation is ~100x an iteration, and lanes call it at different iterations (whenever their point finds a
        // new best minimal model after iteration 50) -- run where the loop calls it, the wave executed up to 64 of them one after the other with one lane
        // active each.  Here a lane that needs one PARKS (need != 0) while the others go on iterating; when every lane is parked or finished the parked
        // ones run it together, then all resume.  Rounds = the largest number of local optimisations of any lane of the wave (2-4).
        //   phase 0: top of iteration `it` (loop test, the lo_starting_iterations_ call of ransac.h:162-176)   phase 1: sample, solve, score, update
        int phase = 0; bool done = false;
        if (!resume) it = 0;
        while (true) {
            int need = need0; need0 = 0;                                      // 1: LocalOptimization(best_model) at lo_start; 2: LocalOptimization(best_minimal_model)
            while (!done && need == 0) {
                if (phase == 0) {
                    if (!(it < max_it)) { done = true; break; }
                    if (it == o.lo_start && best_min_score < TRI_DMAX) { need = 1; break; }
                    phase = 1;
                }
                int sample[2] = {smp[2 * it], smp[2 * it + 1]};
                double X[3];
                tt_dlt<4>(c, sample, 2, 2, X);                                // MinimalSolver: one model
                const double sc = tt_score(c, X, o.thr);
                if (sc < best_min_score || it == o.lo_start) {                // ransac.h:183-225
                    const bool best_min_model = sc < best_min_score;
                    if (best_min_model) { best_min_score = sc; best_min[0] = X[0]; best_min[1] = X[1]; best_min[2] = X[2]; tt_update(best_min_score, best_min, &best_score, best); }
                    const bool run_lo = (it >= o.lo_start && best_min_score < TRI_DMAX);
                    if (best_min_model || run_lo) {
                        if (run_lo) { need = 2; break; }
                        nin = tt_inliers(c, best, o.thr, listI);
                        max_it = req[nin];
                    }
                }
                ++it; phase = 0;
            }
            if (need != 0 && budget <= 0) {                                   // this launch's local optimisation is spent: the point goes to the next launch, parked
                TtState& S = state[p];
                S.best[0] = best[0]; S.best[1] = best[1]; S.best[2] = best[2]; S.best_score = best_score;
                S.best_min[0] = best_min[0]; S.best_min[1] = best_min[1]; S.best_min[2] = best_min[2]; S.best_min_score = best_min_score;
                S.it = it; S.max_it = max_it; S.nin = nin; S.lo_count = lo_count; S.need = need; S.pos = g.pos; S.ovf = g.overflow ? 1 : 0;
                next_list[atomicAdd(next_count, 1)] = p;
                suspended = true; done = true; need = 0;
            }
            if (__ballot(need != 0) == 0ull) break;                           // every lane of the wave has finished its loop (or left for the next launch)
            if (need != 0) {                                                  // the parked lanes, together
                --budget;
                double model[3], msc;
                if (need == 1) { model[0] = best[0]; model[1] = best[1]; model[2] = best[2]; msc = best_score; }
                else { model[0] = best_min[0]; model[1] = best_min[1]; model[2] = best_min[2]; msc = best_min_score; }
                ++lo_count;
                tt_local_optimization(c, o, g, base, work, model, &msc);
                if (need == 1) { best[0] = model[0]; best[1] = model[1]; best[2] = model[2]; best_score = msc; phase = 1; }      // then the same iteration goes on with its sample
                else { best_min[0] = model[0]; best_min[1] = model[1]; best_min[2] = model[2]; tt_update(msc, best_min, &best_score, best); }
                nin = tt_inliers(c, best, o.thr, listI);
                max_it = req[nin];
                if (need == 2) { ++it; phase = 0; }
            }
        }
        if (suspended) { if (g.overflow) atomicExch(overflow, 1); return; }
        if (it <= o.lo_start && best_score < TRI_DMAX) {                      // ransac.h:232-243 (never reached with min_num_iterations_ 100 > 50)
            ++lo_count;
            tt_local_optimization(c, o, g, base, work, best, &best_score);
            nin = tt_inliers(c, best, o.thr, listI);
        }
        {   // final_least_squares_ (ransac.h:245-262)
            double refined[3] = {best[0], best[1], best[2]};
            (void)tt_lsq(c, listI, nin, refined);
            const double sc = tt_score(c, refined, o.thr);
            if (sc < best_score) { best_score = sc; best[0] = refined[0]; best[1] = refined[1]; best[2] = refined[2]; nin = tt_inliers(c, best, o.thr, listI); }
        }
        if (g.overflow) atomicExch(overflow, 1);
        if (flags) for (int k = 0; k < nin; k++) flags[c.j0 + listI[k]] = 1;
    }
    if (nin < 3) { best[0] = best[1] = best[2] = 0.0; }                        // src/sfm.cpp:172,186
    pts[3 * (size_t)p] = best[0]; pts[3 * (size_t)p + 1] = best[1]; pts[3 * (size_t)p + 2] = best[2];
    if (num_inliers) num_inliers[p] = nin;
    if (stats) { stats[2 * (size_t)p] = it; stats[2 * (size_t)p + 1] = (unsigned)lo_count; }
}
// The same body at three register budgets (waves per SIMD).  The compiler's free choice is 256 VGPRs + 38 AGPRs = one wave per SIMD; 100 000 points are
// 1563 waves for 1024 SIMDs, i.e. two rounds with the second one half empty.  SSFM_RETRI_WAVES=1|2|3 selects (profiles/r03_notes.md has the measurement).
#define SSFM_RETRI_ARGS                                                                                                                       \
    const double* __restrict__ ct, const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,        \
    const int* __restrict__ pt_start, int nP, const int* __restrict__ order, TtOpts o, const int* __restrict__ pt_slot,                          \
    const unsigned short* __restrict__ samples, const int* __restrict__ req_ptr, const unsigned* __restrict__ req_it, const unsigned* __restrict__ W, \
    int LW, int* __restrict__ lists, double* __restrict__ pts, int* __restrict__ num_inliers, unsigned* __restrict__ stats,                    \
    unsigned char* __restrict__ flags, int* __restrict__ overflow, TtState* __restrict__ state, int resume, int lo_budget, int* __restrict__ next_list, int* __restrict__ next_count
#define SSFM_RETRI_PASS ct, focal, obs_xy, obs_cam, pt_start, nP, order, o, pt_slot, samples, req_ptr, req_it, W, LW, lists, pts, num_inliers, stats, flags, overflow, state, resume, lo_budget, next_list, next_count
__global__ void __launch_bounds__(64) k_retriangulate_trace(SSFM_RETRI_ARGS) { retriangulate_trace_body<true>(SSFM_RETRI_PASS); }
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_retriangulate_trace_w2(SSFM_RETRI_ARGS) { retriangulate_trace_body<true>(SSFM_RETRI_PASS); }
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_retriangulate_trace_w3(SSFM_RETRI_ARGS) { retriangulate_trace_body<false>(SSFM_RETRI_PASS); }      // (three waves per SIMD leave no LDS for the row cache)

// the estimator's pieces, one lane per task (bit-for-bit parity tests; mirrors oracle_tri_probe)
__global__ void k_tri_probe(const double* __restrict__ ct, const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
                            const int* __restrict__ pt_start, int what, int tasks, const int* __restrict__ task_pt, const int* __restrict__ task_ptr,
                            int* __restrict__ lists, const double* __restrict__ X_in, double* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tasks) return;
    const int p = task_pt[t];
    TtPoint c{ct, obs_xy, obs_cam, pt_start[p], pt_start[p + 1] - pt_start[p], focal[0]};
    int* lst = lists + task_ptr[t]; const int cnt = task_ptr[t + 1] - task_ptr[t];
    double* o = out + 4 * (size_t)t;
    if (what == 0) {
        double X[3];
        if (cnt <= 2) tt_dlt<4>(c, lst, cnt, cnt, X); else tt_dlt<12>(c, lst, min(cnt, 6), cnt, X);
        o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; o[3] = 0.0;
    } else if (what == 1) {
        double X[3] = {X_in[3 * t], X_in[3 * t + 1], X_in[3 * t + 2]};
        const int its = tt_lsq(c, lst, cnt, X);
        o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; o[3] = (double)its;
    } else {
        const double X[3] = {X_in[3 * t], X_in[3 * t + 1], X_in[3 * t + 2]};
        double sc = 0.0; int n1 = 0, n2 = 0;
        for (int i = 0; i < c.n; i++) { const double e = tt_err(c, i, X); sc += (4.0 < e) ? 4.0 : e; n1 += e < 4.0; n2 += e < 4.0 * 1.4142135623730951; }
        o[0] = sc; o[1] = n1; o[2] = n2; o[3] = c.n ? tt_err(c, 0, X) : 0.0;
    }
}

}  // namespace ssfm
using namespace ssfm;

// point-major observation lists over ALL points (Retriangulate does not apply Optimize's filters), cameras ascending, last value of a
// repeated (camera, point) key (std::map semantics, src/sfm.cpp:164-169).  obs_index[k] = position of compacted observation k in the caller's arrays.
// returns true when the caller's arrays ARE the lists (point-major, cameras strictly ascending, every id in range: what the SfM mirror hands over) -- ocam / oxy stay empty then
// and the upload goes straight from the caller's memory (two 20 MB copies into fresh vectors were ~5 ms of a 40 ms call at 1 M observations)
static bool tri_point_lists(const ssfm_ba_problem* p, std::vector<int>& pt_start, std::vector<int>& ocam, std::vector<double>& oxy, std::vector<int64_t>* obs_index) {
    const int Nc = p->num_cameras, Np = p->num_points; const int64_t M = p->num_observations;
    pt_start.assign(Np + 1, 0); ocam.clear(); oxy.clear();
    if (obs_index) obs_index->clear();
    bool direct = true, sorted = true;                      // one pass: order, range and the counts
    for (int64_t i = 0; i < M; i++) {
        const int c = p->obs_cam[i], j = p->obs_pt[i];
        if (i > 0 && (j < p->obs_pt[i - 1] || (j == p->obs_pt[i - 1] && c <= p->obs_cam[i - 1]))) { sorted = false; direct = false; break; }
        if (c < 0 || c >= Nc || j < 0 || j >= Np) { direct = false; break; }
        pt_start[j + 1]++;
    }
    if (direct) {
        for (int j = 0; j < Np; j++) pt_start[j + 1] += pt_start[j];
        if (obs_index) { obs_index->resize(M); for (int64_t i = 0; i < M; i++) (*obs_index)[i] = i; }
        return true;
    }
    if (sorted) for (int64_t i = 1; i < M && sorted; i++) if (p->obs_pt[i] < p->obs_pt[i - 1] || (p->obs_pt[i] == p->obs_pt[i - 1] && p->obs_cam[i] <= p->obs_cam[i - 1])) sorted = false;
    pt_start.assign(Np + 1, 0);
    std::vector<int64_t> order(M);
    for (int64_t i = 0; i < M; i++) order[i] = i;
    if (!sorted) std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return p->obs_pt[a] != p->obs_pt[b] ? p->obs_pt[a] < p->obs_pt[b] : p->obs_cam[a] < p->obs_cam[b]; });
    ocam.reserve(M); oxy.reserve(2 * M);
    if (obs_index) obs_index->reserve(M);
    int64_t i = 0;
    for (int j = 0; j < Np; j++) {
        while (i < M && p->obs_pt[order[i]] < j) i++;
        while (i < M && p->obs_pt[order[i]] == j) {
            const int64_t o = order[i]; const int c = p->obs_cam[o];
            const bool dup = (i + 1 < M && p->obs_pt[order[i + 1]] == j && p->obs_cam[order[i + 1]] == c);
            if (!dup && c >= 0 && c < Nc) { ocam.push_back(c); oxy.push_back(p->obs_xy[2 * o]); oxy.push_back(p->obs_xy[2 * o + 1]); if (obs_index) obs_index->push_back(o); }
            i++;
        }
        pt_start[j + 1] = (int)ocam.size();
    }
    return false;
}

// the enumerating kernel of rounds 1-2 (SSFM_RETRI_ENUMERATE=1): no random stream, statistical agreement with the reference
static int retriangulate_enumerate(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t* num_inliers_out) {
    hipStream_t st = ctx->stream;
    const int Nc = p->num_cameras, Np = p->num_points;
    std::vector<int> pt_start, ocam; std::vector<double> oxy;
    if (tri_point_lists(p, pt_start, ocam, oxy, nullptr)) { ocam.assign(p->obs_cam, p->obs_cam + p->num_observations); oxy.assign(p->obs_xy, p->obs_xy + 2 * p->num_observations); }   // (the caller's arrays are the lists)
    if (ocam.empty()) { ocam.push_back(0); oxy.assign(2, 0.0); }
    DevBuf<double> dcam, drot, df, dxy, dpts; DevBuf<int> dcamidx, dps, dnin;
    std::vector<double> cams(p->cameras, p->cameras + (size_t)Nc * 6), fv = {*p->focal};
    auto body = [&]() -> int {
        SSFM_HIP_CHECK(ctx, upload(dcam, cams, st)); SSFM_HIP_CHECK(ctx, upload(df, fv, st)); SSFM_HIP_CHECK(ctx, upload(dxy, oxy, st));
        SSFM_HIP_CHECK(ctx, upload(dcamidx, ocam, st)); SSFM_HIP_CHECK(ctx, upload(dps, pt_start, st));
        SSFM_HIP_CHECK(ctx, drot.alloc((size_t)Nc * 27)); SSFM_HIP_CHECK(ctx, dpts.alloc((size_t)Np * 3)); SSFM_HIP_CHECK(ctx, dnin.alloc(Np));
        hipLaunchKernelGGL(k_cam_rot, dim3((Nc + 63) / 64), dim3(64), 0, st, dcam.p, drot.p, Nc);
        hipLaunchKernelGGL(k_retriangulate, dim3((Np + 63) / 64), dim3(64), 0, st, dcam.p, drot.p, df.p, reinterpret_cast<const double2*>(dxy.p), dcamidx.p, dps.p, Np,
                           4.0, dpts.p, dnin.p);                                   // squared_inlier_threshold_ = 4, src/sfm.cpp:176
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(p->points, dpts.p, (size_t)Np * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        if (num_inliers_out) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(num_inliers_out, dnin.p, (size_t)Np * sizeof(int), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    const int rc = body();
    dcam.free(); drot.free(); df.free(); dxy.free(); dpts.free(); dcamidx.free(); dps.free(); dnin.free();
    return rc;
}

// UniformSampling<TriangulationEstimator>(seed 0) for a point of n observations (sampling.h:46-135): the first `iters` minimal samples.
// libstdc++'s std::mt19937 and std::uniform_int_distribution, as a build of the reference draws them.
static void tri_sample_sequence(int n, int iters, unsigned short* out /* [iters * 2] */) {
    std::mt19937 rng; rng.seed(0u);
    const bool draw = ((double)n / (double)(n - 2)) < M_E;                      // DrawBetterThanShuffle(2, n), sampling.h:66-75
    std::uniform_int_distribution<int> udist(0, n - 1);
    std::vector<int> perm(n);
    for (int it = 0; it < iters; it++) {
        int s[2];
        if (draw) {                                                                // DrawSample, sampling.h:78-95
            for (int i = 0; i < 2; i++) { bool found = true; while (found) { found = false; s[i] = udist(rng); for (int j = 0; j < i; j++) if (s[j] == s[i]) { found = true; break; } } }
        } else {                                                                   // ShuffleSample, sampling.h:100-124 (n == 3 only: sample_size 2 never equals num_data here)
            std::iota(perm.begin(), perm.end(), 0);
            for (int i = 0; i < n - 1; i++) { std::uniform_int_distribution<int> d(i, n - 1); std::swap(perm[i], perm[d(rng)]); }
            s[0] = perm[0]; s[1] = perm[1];
        }
        out[2 * it] = (unsigned short)s[0]; out[2 * it + 1] = (unsigned short)s[1];
    }
}
// The device reduces the raw words of mt19937(0) the way libstdc++ >= 11 does (Lemire's multiply-shift with rejection, bits/uniform_int_dist.h), while the
// sampler tables above come from whatever libstdc++ this library is linked against.  One check per process that the two are the same algorithm: a few
// hundred draws over ranges of every size class against the host restatement of tt_uniform_int.
static bool tri_libstdcxx_is_lemire() {
    static const bool ok = [] {
        std::mt19937 a; a.seed(0u); std::mt19937 w; w.seed(0u);
        const int hi[] = {1, 2, 3, 5, 6, 13, 20, 63, 64, 999, 65534};
        for (int rep = 0; rep < 40; rep++)
            for (int h : hi) {
                std::uniform_int_distribution<int> d(rep % (h + 1), h);
                const int want = d(a), lo = rep % (h + 1);
                const unsigned range = (unsigned)(h - lo) + 1u; int got;
                while (true) {
                    const unsigned long long prod = (unsigned long long)(unsigned)w() * (unsigned long long)range;
                    const unsigned low = (unsigned)prod;
                    if (low < range) { const unsigned thr = (0u - range) % range; if (low < thr) continue; }
                    got = lo + (int)(unsigned)(prod >> 32); break;
                }
                if (got != want) return false;
            }
        return true;
    }();
    return ok;
}
// utils::NumRequiredIterations (utils.h:110-140) with the options of EstimateModel's calls (ransac.h:171-175)
static unsigned tri_num_required_iterations(double ratio, double pmiss, int ssize, unsigned mn, unsigned mx) {
    if (ratio <= 0.0) return mx;
    if (ratio >= 1.0) return mn;
    const double pn = 1.0 - std::pow(ratio, (double)ssize);
    if (pn >= 0.99999999999999) return mx;
    const double it = std::ceil(std::log(pmiss) / std::log(pn) + 0.5);
    return std::max(mn, std::min((unsigned)it, mx));
}

struct TriDevice {                  // the uploaded problem of one call
    DevBuf<double> dct, df, dxy, dpts, dX, dout; DevBuf<int> dcamidx, dps, dnin, dslot, dreqptr, dlists, dovf, dtpt, dtptr, dtl, dorder, dnext0, dnext1, dcount; DevBuf<unsigned> dreq, dW, dstats; DevBuf<TtState> dstate;
    DevBuf<unsigned short> dsamples; DevBuf<unsigned char> dflags;
    void free_all() { dct.free(); df.free(); dxy.free(); dpts.free(); dX.free(); dout.free(); dcamidx.free(); dps.free(); dnin.free(); dslot.free(); dreqptr.free(); dlists.free();
                      dovf.free(); dtpt.free(); dtptr.free(); dtl.free(); dorder.free(); dnext0.free(); dnext1.free(); dcount.free(); dstate.free(); dreq.free(); dW.free(); dstats.free(); dsamples.free(); dflags.free(); }
};
static int tri_upload_problem(ssfm_ctx* ctx, hipStream_t st, const ssfm_ba_problem* p, TriDevice& D, std::vector<int>& pt_start, std::vector<int64_t>* obs_index, int* total_out) {
    const int Nc = p->num_cameras;
    std::vector<int> ocam; std::vector<double> oxy;
    const bool direct = tri_point_lists(p, pt_start, ocam, oxy, obs_index) && p->num_observations > 0;
    *total_out = direct ? (int)p->num_observations : (int)ocam.size();
    if (!direct && ocam.empty()) { ocam.push_back(0); oxy.assign(2, 0.0); }
    std::vector<double> ct((size_t)std::max(Nc, 1) * TC, 0.0), fv = {*p->focal};
    for (int c = 0; c < Nc; c++) tri_camera_record(p->cameras + (size_t)6 * c, &ct[(size_t)TC * c]);
    SSFM_HIP_CHECK(ctx, upload(D.dct, ct, st)); SSFM_HIP_CHECK(ctx, upload(D.df, fv, st));
    if (direct) {                                           // the caller's arrays are the lists: no copy on the host
        const size_t M = (size_t)p->num_observations;
        SSFM_HIP_CHECK(ctx, D.dxy.alloc(2 * M)); SSFM_HIP_CHECK(ctx, D.dcamidx.alloc(M));
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(D.dxy.p, p->obs_xy, 2 * M * sizeof(double), hipMemcpyHostToDevice, st));
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(D.dcamidx.p, p->obs_cam, M * sizeof(int), hipMemcpyHostToDevice, st));
    } else { SSFM_HIP_CHECK(ctx, upload(D.dxy, oxy, st)); SSFM_HIP_CHECK(ctx, upload(D.dcamidx, ocam, st)); }
    SSFM_HIP_CHECK(ctx, upload(D.dps, pt_start, st));
    if (direct) SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));     // (pageable sources: the copies have been staged when the call returns, but be explicit)
    return SSFM_OK;
}

static int retriangulate_trace(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t* num_inliers_out, uint32_t* stats_out, uint8_t* inlier_flags_out) {
    hipStream_t st = ctx->stream;
    const int Np = p->num_points; const int64_t M = p->num_observations;
    TriDevice D; std::vector<int> pt_start; std::vector<int64_t> obs_index; int total = 0;
    TtOpts o; o.thr = 4.0; o.mult = std::sqrt(2.0); o.lo_steps = 10; o.lsq_it = 4; o.min_sample_mult = 7; o.non_min_mult = 3; o.min_it = 100u; o.lo_start = 50u;   // src/sfm.cpp:175-177 + ransac.h:47-88
    if (!tri_libstdcxx_is_lemire())
        return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate: this libstdc++'s uniform_int_distribution is not the multiply-shift reduction the device replays; use SSFM_RETRI_MODE_ENUMERATE");
    auto body = [&]() -> int {
        { const int rc = tri_upload_problem(ctx, st, p, D, pt_start, inlier_flags_out ? &obs_index : nullptr, &total); if (rc) return rc; }
        // one sampler sequence + one iteration-count table per distinct track length
        std::vector<int> slot_of_n, pt_slot(std::max(Np, 1), 0), ns;
        for (int j = 0; j < Np; j++) {
            const int n = pt_start[j + 1] - pt_start[j];
            if (n < 3) continue;
            if (n > 65535) return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate: more than 65535 observations of one point");
            if ((int)slot_of_n.size() <= n) slot_of_n.resize(n + 1, -1);
            if (slot_of_n[n] < 0) { slot_of_n[n] = (int)ns.size(); ns.push_back(n); }
            pt_slot[j] = slot_of_n[n];
        }
        const int S = std::max((int)ns.size(), 1);
        std::vector<unsigned short> samples((size_t)S * TRI_MAX_IT * 2, 0); std::vector<int> req_ptr(S + 1, 0); std::vector<unsigned> req;
        for (int s = 0; s < (int)ns.size(); s++) {
            tri_sample_sequence(ns[s], TRI_MAX_IT, &samples[(size_t)s * TRI_MAX_IT * 2]);
            for (int k = 0; k <= ns[s]; k++) req.push_back(tri_num_required_iterations((double)k / (double)ns[s], 1.0 - 0.9999, 2, o.min_it, (unsigned)TRI_MAX_IT));
            req_ptr[s + 1] = (int)req.size();
        }
        if (req.empty()) req.push_back(0);
        // points of equal track length run the same sampler sequence: adjacent lanes get points of the same length (stable, so neighbours stay neighbours)
        // (a stable counting sort by track length: std::stable_sort took 2-3 ms of a 17 ms call at 100 000 points)
        std::vector<int> order(std::max(Np, 1), 0);
        {
            int nmax = 0; for (int j = 0; j < Np; j++) nmax = std::max(nmax, pt_start[j + 1] - pt_start[j]);
            std::vector<int> first((size_t)nmax + 2, 0);
            for (int j = 0; j < Np; j++) first[(size_t)(pt_start[j + 1] - pt_start[j]) + 1]++;
            for (int k = 0; k <= nmax; k++) first[k + 1] += first[k];
            for (int j = 0; j < Np; j++) order[first[pt_start[j + 1] - pt_start[j]]++] = j;
        }
        SSFM_HIP_CHECK(ctx, upload(D.dslot, pt_slot, st)); SSFM_HIP_CHECK(ctx, upload(D.dsamples, samples, st)); SSFM_HIP_CHECK(ctx, upload(D.dreqptr, req_ptr, st));
        SSFM_HIP_CHECK(ctx, upload(D.dreq, req, st)); SSFM_HIP_CHECK(ctx, upload(D.dorder, order, st));
        SSFM_HIP_CHECK(ctx, D.dlists.alloc((size_t)3 * std::max(total, 1))); SSFM_HIP_CHECK(ctx, D.dpts.alloc((size_t)std::max(Np, 1) * 3)); SSFM_HIP_CHECK(ctx, D.dnin.alloc(std::max(Np, 1)));
        SSFM_HIP_CHECK(ctx, D.dovf.alloc(1)); SSFM_HIP_CHECK(ctx, D.dcount.alloc(1));
        SSFM_HIP_CHECK(ctx, D.dstate.alloc((size_t)std::max(Np, 1))); SSFM_HIP_CHECK(ctx, D.dnext0.alloc((size_t)std::max(Np, 1))); SSFM_HIP_CHECK(ctx, D.dnext1.alloc((size_t)std::max(Np, 1)));
        if (stats_out) SSFM_HIP_CHECK(ctx, D.dstats.alloc((size_t)2 * std::max(Np, 1)));
        if (inlier_flags_out) SSFM_HIP_CHECK(ctx, D.dflags.alloc(std::max(total, 1)));
        // raw words of the local optimisation's std::mt19937(0); a lane that needs more makes the launch repeat with a longer table
        std::vector<unsigned> W; std::mt19937 rng; rng.seed(0u);
        const char* e = getenv("SSFM_RETRI_WORDS");
        size_t LW = e ? (size_t)std::max(atoi(e), 16) : (size_t)1 << 16;
        while (true) {
            while (W.size() < LW) W.push_back((unsigned)rng());
            SSFM_HIP_CHECK(ctx, upload(D.dW, W, st));
            SSFM_HIP_CHECK(ctx, hipMemsetAsync(D.dovf.p, 0, sizeof(int), st));
            if (inlier_flags_out) SSFM_HIP_CHECK(ctx, hipMemsetAsync(D.dflags.p, 0, (size_t)std::max(total, 1), st));
            // register budget in waves per SIMD.  Measured at 100k points x 6 (1563 waves, one round either way): 32.8 / 23.4 / 25.1 ms at 1 / 2 / 3.  Two waves per
            // SIMD hold 8 x CUs = 2048 waves at once; a problem with more (BASELINE configs[2]: 170k points = 2657 waves) would run a second, mostly empty round at
            // two, but still fits one round at three (3072) -- and at sizes of many rounds three has the higher throughput (122 against 88 waves per ms).
            const char* ew = getenv("SSFM_RETRI_WAVES");
            const int waves = ew ? atoi(ew) : (((Np + 63) / 64 <= 8 * ctx->num_cus) ? 2 : 3);
            // round 6: one local optimisation per launch (TtState); the first launch takes every point in `order`, launch r + 1 the points launch r parked, as dense waves
            // (SSFM_RETRI_ROUNDS=0: everything in one launch, rounds 3-5)
            const bool rounds_on = !(getenv("SSFM_RETRI_ROUNDS") && atoi(getenv("SSFM_RETRI_ROUNDS")) == 0);      // (read per call: tests switch it)
            const int budget = rounds_on ? 1 : (1 << 30);
#define SSFM_RETRI_LAUNCH(K, N_, LIST_, RESUME_, NEXT_) hipLaunchKernelGGL(K, dim3(((N_) + 63) / 64), dim3(64), 0, st, D.dct.p, D.df.p, reinterpret_cast<const double2*>(D.dxy.p), D.dcamidx.p, D.dps.p, (N_), \
                                           (LIST_), o, D.dslot.p, D.dsamples.p, D.dreqptr.p, D.dreq.p, D.dW.p, (int)LW, D.dlists.p, D.dpts.p, D.dnin.p,                   \
                                           stats_out ? D.dstats.p : nullptr, inlier_flags_out ? D.dflags.p : nullptr, D.dovf.p, D.dstate.p, (RESUME_), budget, (NEXT_), D.dcount.p)
            int n_act = Np; const int* list = D.dorder.p; int resume = 0, round = 0;
            while (n_act > 0) {
                int* next = (round & 1) ? D.dnext1.p : D.dnext0.p;
                SSFM_HIP_CHECK(ctx, hipMemsetAsync(D.dcount.p, 0, sizeof(int), st));
                if (waves == 2) SSFM_RETRI_LAUNCH(k_retriangulate_trace_w2, n_act, list, resume, next); else if (waves == 3) SSFM_RETRI_LAUNCH(k_retriangulate_trace_w3, n_act, list, resume, next);
                else SSFM_RETRI_LAUNCH(k_retriangulate_trace, n_act, list, resume, next);
                SSFM_HIP_CHECK(ctx, hipGetLastError());
                int cnt = 0;
                SSFM_HIP_CHECK(ctx, hipMemcpyAsync(&cnt, D.dcount.p, sizeof(int), hipMemcpyDeviceToHost, st));
                SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
                n_act = cnt; list = next; resume = 1; round++;
                if (round > 100000) return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate: the local-optimisation rounds do not end");
            }
#undef SSFM_RETRI_LAUNCH
            int ovf = 0;
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(&ovf, D.dovf.p, sizeof(int), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            if (!ovf) break;
            if (LW >= ((size_t)1 << 28)) return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate: random stream table exhausted");
            LW *= 4;
        }
        if (Np > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(p->points, D.dpts.p, (size_t)Np * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        if (num_inliers_out && Np > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(num_inliers_out, D.dnin.p, (size_t)Np * sizeof(int), hipMemcpyDeviceToHost, st));
        if (stats_out && Np > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(stats_out, D.dstats.p, (size_t)2 * Np * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        std::vector<unsigned char> fl;
        if (inlier_flags_out) { fl.resize(std::max(total, 1)); SSFM_HIP_CHECK(ctx, hipMemcpyAsync(fl.data(), D.dflags.p, fl.size(), hipMemcpyDeviceToHost, st)); }
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (inlier_flags_out) { std::memset(inlier_flags_out, 0, (size_t)M); for (int k = 0; k < total; k++) inlier_flags_out[obs_index[k]] = fl[k]; }
        return SSFM_OK;
    };
    const int rc = body();
    D.free_all();
    return rc;
}

extern "C" int ssfm_retriangulate_mode(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t mode, int32_t* num_inliers_out, uint32_t* stats_out, uint8_t* inlier_flags_out) {
    if (!ctx || !p || !p->points || !p->cameras || !p->focal || p->num_points < 0 || p->num_cameras < 0 || p->num_observations < 0) return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate: bad arguments");
    if (mode != SSFM_RETRI_MODE_TRACE && mode != SSFM_RETRI_MODE_ENUMERATE) return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate_mode: unknown mode");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (mode == SSFM_RETRI_MODE_ENUMERATE) {
        if (stats_out || inlier_flags_out) return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate_ex: the enumerating mode has no trace to report");
        return retriangulate_enumerate(ctx, p, num_inliers_out);
    }
    return retriangulate_trace(ctx, p, num_inliers_out, stats_out, inlier_flags_out);
}
// default mode of the two-argument forms: the trace replay; SSFM_RETRI_ENUMERATE=1 (read once) turns the default into the enumerating kernel
static int retri_default_mode() {
    static const int m = [] { const char* e = getenv("SSFM_RETRI_ENUMERATE"); return (e && atoi(e) != 0) ? SSFM_RETRI_MODE_ENUMERATE : SSFM_RETRI_MODE_TRACE; }();
    return m;
}
extern "C" int ssfm_retriangulate_ex(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t* num_inliers_out, uint32_t* stats_out, uint8_t* inlier_flags_out) {
    return ssfm_retriangulate_mode(ctx, p, retri_default_mode(), num_inliers_out, stats_out, inlier_flags_out);
}
extern "C" int ssfm_retriangulate(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t* num_inliers_out) { return ssfm_retriangulate_ex(ctx, p, num_inliers_out, nullptr, nullptr); }

// TriangulationEstimator's pieces on the device, one lane per task (mirrors oracle_tri_probe; bit-for-bit parity tests)
extern "C" int ssfm_tri_probe(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t what, int32_t tasks, const int32_t* task_pt, const int32_t* task_ptr, const int32_t* lists,
                              const double* X_in, double* out) {
    if (!ctx || !p || tasks <= 0 || !task_pt || !task_ptr || !lists || !out || what < 0 || what > 2 || (what != 0 && !X_in)) return fail(ctx, SSFM_ERR_INVALID, "ssfm_tri_probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    TriDevice D; std::vector<int> pt_start; int total = 0;
    auto body = [&]() -> int {
        { const int rc = tri_upload_problem(ctx, st, p, D, pt_start, nullptr, &total); if (rc) return rc; }
        for (int t = 0; t < tasks; t++) {
            if (task_pt[t] < 0 || task_pt[t] >= p->num_points) return fail(ctx, SSFM_ERR_INVALID, "ssfm_tri_probe: point out of range");
            const int n = pt_start[task_pt[t] + 1] - pt_start[task_pt[t]];
            for (int k = task_ptr[t]; k < task_ptr[t + 1]; k++) if (lists[k] < 0 || lists[k] >= n) return fail(ctx, SSFM_ERR_INVALID, "ssfm_tri_probe: observation out of range");
            if (what == 0 && (task_ptr[t + 1] - task_ptr[t] < 1 || task_ptr[t + 1] - task_ptr[t] > 6)) return fail(ctx, SSFM_ERR_INVALID, "ssfm_tri_probe: DLT samples hold 1..6 observations");
        }
        std::vector<int> hp(task_pt, task_pt + tasks), hptr(task_ptr, task_ptr + tasks + 1), hl(lists, lists + task_ptr[tasks]); if (hl.empty()) hl.push_back(0);
        std::vector<double> hX((size_t)3 * tasks, 0.0); if (X_in) std::memcpy(hX.data(), X_in, hX.size() * sizeof(double));
        SSFM_HIP_CHECK(ctx, upload(D.dtpt, hp, st)); SSFM_HIP_CHECK(ctx, upload(D.dtptr, hptr, st)); SSFM_HIP_CHECK(ctx, upload(D.dtl, hl, st)); SSFM_HIP_CHECK(ctx, upload(D.dX, hX, st));
        SSFM_HIP_CHECK(ctx, D.dout.alloc((size_t)4 * tasks));
        hipLaunchKernelGGL(k_tri_probe, dim3((tasks + 63) / 64), dim3(64), 0, st, D.dct.p, D.df.p, reinterpret_cast<const double2*>(D.dxy.p), D.dcamidx.p, D.dps.p, what, tasks,
                           D.dtpt.p, D.dtptr.p, D.dtl.p, D.dX.p, D.dout.p);
        SSFM_HIP_CHECK(ctx, hipGetLastError());
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(out, D.dout.p, (size_t)4 * tasks * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    const int rc = body();
    D.free_all();
    return rc;
}
