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
#include <cstdio>
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
            radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3))); decrease = 2.0; last_ok = true;
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

}  // namespace ssfm
using namespace ssfm;

extern "C" int ssfm_retriangulate(ssfm_ctx* ctx, ssfm_ba_problem* p, int32_t* num_inliers_out) {
    if (!ctx || !p || !p->points) return fail(ctx, SSFM_ERR_INVALID, "ssfm_retriangulate: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int Nc = p->num_cameras, Np = p->num_points; const int64_t M = p->num_observations;
    // point-major observation lists over ALL points (Retriangulate does not apply Optimize's filters), cameras ascending,
    // last value of a repeated (camera, point) key
    std::vector<int64_t> order(M);
    for (int64_t i = 0; i < M; i++) order[i] = i;
    bool sorted = true;
    for (int64_t i = 1; i < M && sorted; i++) if (p->obs_pt[i] < p->obs_pt[i - 1] || (p->obs_pt[i] == p->obs_pt[i - 1] && p->obs_cam[i] <= p->obs_cam[i - 1])) sorted = false;
    if (!sorted) std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return p->obs_pt[a] != p->obs_pt[b] ? p->obs_pt[a] < p->obs_pt[b] : p->obs_cam[a] < p->obs_cam[b]; });
    std::vector<int> pt_start(Np + 1, 0), ocam; std::vector<double> oxy; ocam.reserve(M); oxy.reserve(2 * M);
    { int64_t i = 0;
      for (int j = 0; j < Np; j++) {
          while (i < M && p->obs_pt[order[i]] < j) i++;
          while (i < M && p->obs_pt[order[i]] == j) {
              const int64_t o = order[i]; const int c = p->obs_cam[o];
              const bool dup = (i + 1 < M && p->obs_pt[order[i + 1]] == j && p->obs_cam[order[i + 1]] == c);
              if (!dup && c >= 0 && c < Nc) { ocam.push_back(c); oxy.push_back(p->obs_xy[2 * o]); oxy.push_back(p->obs_xy[2 * o + 1]); }
              i++;
          }
          pt_start[j + 1] = (int)ocam.size();
      } }
    DevBuf<double> dcam, drot, df, dxy, dpts; DevBuf<int> dcamidx, dps, dnin;
    std::vector<double> cams(p->cameras, p->cameras + (size_t)Nc * 6), fv = {*p->focal};
    SSFM_HIP_CHECK(ctx, upload(dcam, cams, st)); SSFM_HIP_CHECK(ctx, upload(df, fv, st)); SSFM_HIP_CHECK(ctx, upload(dxy, oxy, st));
    SSFM_HIP_CHECK(ctx, upload(dcamidx, ocam, st)); SSFM_HIP_CHECK(ctx, upload(dps, pt_start, st));
    SSFM_HIP_CHECK(ctx, drot.alloc((size_t)Nc * 27)); SSFM_HIP_CHECK(ctx, dpts.alloc((size_t)Np * 3)); SSFM_HIP_CHECK(ctx, dnin.alloc(Np));
    hipLaunchKernelGGL(k_cam_rot, dim3((Nc + 63) / 64), dim3(64), 0, st, dcam.p, drot.p, Nc);
    hipLaunchKernelGGL(k_retriangulate, dim3((Np + 63) / 64), dim3(64), 0, st, dcam.p, drot.p, df.p, reinterpret_cast<const double2*>(dxy.p), dcamidx.p, dps.p, Np,
                       4.0, dpts.p, dnin.p);                                   // squared_inlier_threshold_ = 4, src/sfm.cpp:176
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(p->points, dpts.p, (size_t)Np * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (num_inliers_out) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(num_inliers_out, dnin.p, (size_t)Np * sizeof(int), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    dcam.free(); drot.free(); df.free(); dxy.free(); dpts.free(); dcamidx.free(); dps.free(); dnin.free();
    return SSFM_OK;
}
