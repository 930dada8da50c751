// ORACLE (test infrastructure only) -- CPU restatement of the bundle adjustment behind
// sphericalsfm::SfM::Optimize (reference src/sfm.cpp:228-290).
//
//  * residual           : ReprojectionError::operator()      src/sfm.cpp:30-66, evaluated with
//                         dual numbers of width 10 exactly as AutoDiffCostFunction<...,2,1,3,3,3>
//                         (src/sfm.cpp:219) does -> columns [focal | t(3) | r(3) | X(3)];
//  * block structure    : SfM::AddResidual                   src/sfm.cpp:214-226;
//  * which points enter : SfM::Optimize build loop           src/sfm.cpp:240-263 (exists, |X| != 0,
//                         >= 3 observations; all its observations, point-major);
//  * loss / options     : PreOptimize + ConfigureSolverOptions src/sfm.cpp:194-212
//                         (CauchyLoss(1.0), LM, SPARSE_SCHUR, 2000 iterations, 100 invalid steps);
//  * minimiser          : oracle/lm.hpp (Ceres 2.2.0 trust-region loop, restated);
//  * linear solve       : Schur elimination of the point blocks, exact Cholesky of the reduced
//                         camera(+focal) system (oracle/skyline.hpp).
// PARITY UNPINNED for this file (the Ceres path cannot be run here; see ssfm_oracle.h).  This file doubles as bench.py's timed CPU baseline
// ("cpu_baseline.kind = port"): a proxy for the Ceres path, not Ceres.
#include <omp.h>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <array>
#include <vector>
#include "lm.hpp"
#include "rotation.hpp"
#include "skyline.hpp"
#include "ssfm_oracle.h"

namespace oracle {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// rho(s) tables: Ceres CauchyLoss / SoftLOneLoss (loss_function.cc), restated.
static inline void loss_eval(int type, double a, double s, double rho[3]) {
    if (type == 1) {
        const double b = a * a, c = 1.0 / b;
        const double sum = 1.0 + s * c, inv = 1.0 / sum;
        rho[0] = b * std::log(sum);
        rho[1] = std::fmax(std::numeric_limits<double>::min(), inv);
        rho[2] = -c * (inv * inv);
    } else if (type == 2) {
        const double b = a * a, c = 1.0 / b;
        const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
        rho[0] = 2.0 * b * (tmp - 1.0);
        rho[1] = std::fmax(std::numeric_limits<double>::min(), 1.0 / tmp);
        rho[2] = -(c * rho[1]) / (2.0 * sum);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

// src/sfm.cpp:38-63
template <typename T>
static inline void reprojection_error(const T& focal, const T t[3], const T r[3], const T X[3],
                                      double ox, double oy, T res[2]) {
    T p[3];
    AngleAxisRotatePoint(r, X, p);
    p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
    const T xp = p[0] / p[2];
    const T yp = p[1] / p[2];
    res[0] = focal * xp - ox;
    res[1] = focal * yp - oy;
}

struct ObsLin {          // robustified, unscaled
    double r[2];
    double Jf[2];
    double Jc[2][6];     // t(3), r(3)
    double Jp[2][3];
};

struct BAOracle : LMProblem {
    // ---- flattened problem (src/sfm.cpp:240-263)
    int Nc = 0;
    bool focal_free = false;
    int loss_type = 1; double loss_a = 1.0;
    std::vector<double> cam0;            // [Nc*6] constants for fixed blocks
    double focal0 = 0;
    std::vector<int> used_pt;            // compact -> original point id
    std::vector<double> pt0;             // [nP*3] constants
    std::vector<int> pt_start;           // CSR over compact points
    std::vector<int> ob_cam, ob_orig;    // per flattened observation
    std::vector<double> ob_x, ob_y;
    std::vector<int> ob_pt;              // compact point of each observation
    // camera-major lists
    std::vector<int> cam_start, cam_obs;
    // ---- parameter layout: [focal?][cams: t? r?][points]
    int nx = 0, nf = 0;                  // nf = reduced (focal + camera) unknowns, first in x
    int focal_idx = -1;
    std::vector<int> cam_idx;            // [Nc*6] x-index or -1
    std::vector<int> pt_idx;             // [nP] x-index of X or -1 (fixed point)
    // ---- linearisation
    std::vector<ObsLin> lin;
    // ---- reduced system
    std::vector<int> sky_of_x;           // x-index (< nf) -> skyline index
    std::vector<int> cam_pos;            // camera -> position in elimination order
    Skyline S;
    std::vector<double> rhs, Vinv, gp, Wfp;
    // timing
    double t_lin = 0, t_schur = 0, t_chol = 0, t_cost = 0;

    int num_parameters() const override { return nx; }

    inline void unpack(const double* x, int c, double t[3], double r[3]) const {
        for (int k = 0; k < 3; k++) { int i = cam_idx[c * 6 + k]; t[k] = i >= 0 ? x[i] : cam0[c * 6 + k]; }
        for (int k = 0; k < 3; k++) { int i = cam_idx[c * 6 + 3 + k]; r[k] = i >= 0 ? x[i] : cam0[c * 6 + 3 + k]; }
    }
    inline void unpack_pt(const double* x, int p, double X[3]) const {
        int i = pt_idx[p];
        for (int k = 0; k < 3; k++) X[k] = i >= 0 ? x[i + k] : pt0[p * 3 + k];
    }

    bool cost_only(const double* x, double* cost) override {
        const double t0 = now_s();
        const int nP = (int)used_pt.size();
        const double f = focal_free ? x[focal_idx] : focal0;
        std::vector<double> part(nP);
        bool ok = true;
#pragma omp parallel for schedule(static)
        for (int p = 0; p < nP; p++) {
            double X[3]; unpack_pt(x, p, X);
            double acc = 0;
            for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                double t[3], r[3], res[2]; unpack(x, ob_cam[j], t, r);
                reprojection_error<double>(f, t, r, X, ob_x[j], ob_y[j], res);
                double rho[3]; loss_eval(loss_type, loss_a, res[0] * res[0] + res[1] * res[1], rho);
                acc += 0.5 * rho[0];
            }
            part[p] = acc;
        }
        double c = 0; for (int p = 0; p < nP; p++) c += part[p];
        if (!std::isfinite(c)) ok = false;
        *cost = c;
        t_cost += now_s() - t0;
        return ok;
    }

    bool linearize(const double* x, double* cost, double* gradient) override {
        const double t0 = now_s();
        const int nP = (int)used_pt.size();
        typedef Jet<10> J;
        const double fv = focal_free ? x[focal_idx] : focal0;
        std::vector<double> part(nP);
#pragma omp parallel for schedule(static)
        for (int p = 0; p < nP; p++) {
            double Xv[3]; unpack_pt(x, p, Xv);
            double acc = 0;
            for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                double tv[3], rv[3]; unpack(x, ob_cam[j], tv, rv);
                J f(fv, 0), t[3] = {J(tv[0], 1), J(tv[1], 2), J(tv[2], 3)}, r[3] = {J(rv[0], 4), J(rv[1], 5), J(rv[2], 6)},
                  X[3] = {J(Xv[0], 7), J(Xv[1], 8), J(Xv[2], 9)}, res[2];
                reprojection_error<J>(f, t, r, X, ob_x[j], ob_y[j], res);
                double rho[3]; loss_eval(loss_type, loss_a, res[0].a * res[0].a + res[1].a * res[1].a, rho);
                acc += 0.5 * rho[0];
                const double sr = std::sqrt(rho[1]);   // Corrector with rho'' <= 0 (or s == 0): pure scaling
                ObsLin& L = lin[j];
                for (int a = 0; a < 2; a++) {
                    L.r[a] = sr * res[a].a;
                    L.Jf[a] = sr * res[a].v[0];
                    for (int k = 0; k < 6; k++) L.Jc[a][k] = sr * res[a].v[1 + k];
                    for (int k = 0; k < 3; k++) L.Jp[a][k] = sr * res[a].v[7 + k];
                }
            }
            part[p] = acc;
        }
        double c = 0; for (int p = 0; p < nP; p++) c += part[p];
        *cost = c;
        // gradient = J^T r over active columns (sequential: deterministic, cheap)
        std::fill(gradient, gradient + nx, 0.0);
        for (int p = 0; p < nP; p++)
            for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                const ObsLin& L = lin[j]; const int c6 = ob_cam[j] * 6;
                for (int a = 0; a < 2; a++) {
                    if (focal_free) gradient[focal_idx] += L.Jf[a] * L.r[a];
                    for (int k = 0; k < 6; k++) { int i = cam_idx[c6 + k]; if (i >= 0) gradient[i] += L.Jc[a][k] * L.r[a]; }
                    if (pt_idx[p] >= 0) for (int k = 0; k < 3; k++) gradient[pt_idx[p] + k] += L.Jp[a][k] * L.r[a];
                }
            }
        t_lin += now_s() - t0;
        return std::isfinite(c);
    }

    void squared_column_norms(const double* scale, double* out) override {
        std::fill(out, out + nx, 0.0);
        const int nP = (int)used_pt.size();
        for (int p = 0; p < nP; p++)
            for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                const ObsLin& L = lin[j]; const int c6 = ob_cam[j] * 6;
                for (int a = 0; a < 2; a++) {
                    if (focal_free) out[focal_idx] += L.Jf[a] * L.Jf[a];
                    for (int k = 0; k < 6; k++) { int i = cam_idx[c6 + k]; if (i >= 0) out[i] += L.Jc[a][k] * L.Jc[a][k]; }
                    if (pt_idx[p] >= 0) for (int k = 0; k < 3; k++) out[pt_idx[p] + k] += L.Jp[a][k] * L.Jp[a][k];
                }
            }
        if (scale) for (int i = 0; i < nx; i++) out[i] *= scale[i] * scale[i];
    }

    // scaled Jacobian pieces of observation j
    inline void scaled(int j, int p, const double* s, double Jf[2], double Jc[2][6], double Jp[2][3]) const {
        const ObsLin& L = lin[j]; const int c6 = ob_cam[j] * 6;
        const double sf = focal_free ? s[focal_idx] : 0.0;
        for (int a = 0; a < 2; a++) {
            Jf[a] = L.Jf[a] * sf;
            for (int k = 0; k < 6; k++) { int i = cam_idx[c6 + k]; Jc[a][k] = i >= 0 ? L.Jc[a][k] * s[i] : 0.0; }
            for (int k = 0; k < 3; k++) Jp[a][k] = pt_idx[p] >= 0 ? L.Jp[a][k] * s[pt_idx[p] + k] : 0.0;
        }
    }

    bool solve(const double* scale, const double* D, double* y) override {
        double t0 = now_s();
        const int nP = (int)used_pt.size();
        // ---- eliminate points: V^-1, g_p, focal coupling  (Ceres SchurEliminator, restated)
#pragma omp parallel for schedule(static)
        for (int p = 0; p < nP; p++) {
            double* Vi = &Vinv[p * 6]; double* g = &gp[p * 3]; double* wf = &Wfp[p * 3];
            for (int k = 0; k < 6; k++) Vi[k] = 0; for (int k = 0; k < 3; k++) g[k] = wf[k] = 0;
            if (pt_idx[p] < 0) continue;
            double V[6] = {0, 0, 0, 0, 0, 0};   // xx xy xz yy yz zz
            for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                double Jf[2], Jc[2][6], Jp[2][3]; scaled(j, p, scale, Jf, Jc, Jp);
                for (int a = 0; a < 2; a++) {
                    V[0] += Jp[a][0] * Jp[a][0]; V[1] += Jp[a][0] * Jp[a][1]; V[2] += Jp[a][0] * Jp[a][2];
                    V[3] += Jp[a][1] * Jp[a][1]; V[4] += Jp[a][1] * Jp[a][2]; V[5] += Jp[a][2] * Jp[a][2];
                    for (int k = 0; k < 3; k++) { g[k] += Jp[a][k] * lin[j].r[a]; wf[k] += Jf[a] * Jp[a][k]; }
                }
            }
            const double* Dp = &D[pt_idx[p]];
            V[0] += Dp[0] * Dp[0]; V[3] += Dp[1] * Dp[1]; V[5] += Dp[2] * Dp[2];
            // symmetric 3x3 inverse by cofactors
            const double c00 = V[3] * V[5] - V[4] * V[4], c01 = V[2] * V[4] - V[1] * V[5], c02 = V[1] * V[4] - V[2] * V[3];
            const double det = V[0] * c00 + V[1] * c01 + V[2] * c02;
            const double id = 1.0 / det;
            Vi[0] = c00 * id; Vi[1] = c01 * id; Vi[2] = c02 * id;
            Vi[3] = (V[0] * V[5] - V[2] * V[2]) * id; Vi[4] = (V[1] * V[2] - V[0] * V[4]) * id;
            Vi[5] = (V[0] * V[3] - V[1] * V[1]) * id;
        }
        // ---- reduced system, one owner thread per camera row block (deterministic)
        S.zero(); std::fill(rhs.begin(), rhs.end(), 0.0);
        const int fsky = focal_free ? sky_of_x[focal_idx] : -1;
#pragma omp parallel for schedule(dynamic, 1)
        for (int c = 0; c < Nc; c++) {
            int idx[6]; bool any = false;
            for (int k = 0; k < 6; k++) { int i = cam_idx[c * 6 + k]; idx[k] = i >= 0 ? sky_of_x[i] : -1; any |= i >= 0; }
            if (!any) continue;
            for (int k = 0; k < 6; k++) if (idx[k] >= 0) { double d = D[cam_idx[c * 6 + k]]; S.at(idx[k], idx[k]) += d * d; }
            for (int q = cam_start[c]; q < cam_start[c + 1]; q++) {
                const int j = cam_obs[q], p = ob_pt[j];
                double Jf[2], Jc[2][6], Jp[2][3]; scaled(j, p, scale, Jf, Jc, Jp);
                const double* r = lin[j].r;
                // U_c, g_c, focal-camera block
                for (int a = 0; a < 6; a++) {
                    if (idx[a] < 0) continue;
                    rhs[idx[a]] += Jc[0][a] * r[0] + Jc[1][a] * r[1];
                    for (int b = 0; b < 6; b++) {
                        if (idx[b] < 0 || idx[b] > idx[a]) continue;
                        S.at(idx[a], idx[b]) += Jc[0][a] * Jc[0][b] + Jc[1][a] * Jc[1][b];
                    }
                    if (fsky >= 0) S.at(fsky, idx[a]) += Jf[0] * Jc[0][a] + Jf[1] * Jc[1][a];
                }
                if (pt_idx[p] < 0) continue;
                // T = W V^-1, W = Jc^T Jp (6x3)
                const double* Vi = &Vinv[p * 6];
                double W[6][3], T[6][3];
                for (int a = 0; a < 6; a++)
                    for (int k = 0; k < 3; k++) W[a][k] = Jc[0][a] * Jp[0][k] + Jc[1][a] * Jp[1][k];
                for (int a = 0; a < 6; a++) {
                    T[a][0] = W[a][0] * Vi[0] + W[a][1] * Vi[1] + W[a][2] * Vi[2];
                    T[a][1] = W[a][0] * Vi[1] + W[a][1] * Vi[3] + W[a][2] * Vi[4];
                    T[a][2] = W[a][0] * Vi[2] + W[a][1] * Vi[4] + W[a][2] * Vi[5];
                }
                const double* g = &gp[p * 3];
                for (int a = 0; a < 6; a++) if (idx[a] >= 0) rhs[idx[a]] -= T[a][0] * g[0] + T[a][1] * g[1] + T[a][2] * g[2];
                if (fsky >= 0) {
                    const double* wf = &Wfp[p * 3];
                    for (int a = 0; a < 6; a++) if (idx[a] >= 0) S.at(fsky, idx[a]) -= T[a][0] * wf[0] + T[a][1] * wf[1] + T[a][2] * wf[2];
                }
                for (int j2 = pt_start[p]; j2 < pt_start[p + 1]; j2++) {
                    const int c2 = ob_cam[j2];
                    if (cam_pos[c2] > cam_pos[c]) continue;           // the other owner writes it
                    double Jf2[2], Jc2[2][6], Jp2[2][3]; scaled(j2, p, scale, Jf2, Jc2, Jp2);
                    int idx2[6]; for (int k = 0; k < 6; k++) { int i = cam_idx[c2 * 6 + k]; idx2[k] = i >= 0 ? sky_of_x[i] : -1; }
                    for (int b = 0; b < 6; b++) {
                        if (idx2[b] < 0) continue;
                        const double w0 = Jc2[0][b] * Jp2[0][0] + Jc2[1][b] * Jp2[1][0];
                        const double w1 = Jc2[0][b] * Jp2[0][1] + Jc2[1][b] * Jp2[1][1];
                        const double w2 = Jc2[0][b] * Jp2[0][2] + Jc2[1][b] * Jp2[1][2];
                        for (int a = 0; a < 6; a++) {
                            if (idx[a] < 0 || idx2[b] > idx[a]) continue;
                            S.at(idx[a], idx2[b]) -= T[a][0] * w0 + T[a][1] * w1 + T[a][2] * w2;
                        }
                    }
                }
            }
        }
        if (fsky >= 0) {
            double sff = D[focal_idx] * D[focal_idx], rf = 0;
            for (int p = 0; p < nP; p++) {
                for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                    const double sf = scale[focal_idx]; const ObsLin& L = lin[j];
                    sff += sf * sf * (L.Jf[0] * L.Jf[0] + L.Jf[1] * L.Jf[1]);
                    rf += sf * (L.Jf[0] * L.r[0] + L.Jf[1] * L.r[1]);
                }
                const double* Vi = &Vinv[p * 6]; const double* wf = &Wfp[p * 3]; const double* g = &gp[p * 3];
                const double u0 = wf[0] * Vi[0] + wf[1] * Vi[1] + wf[2] * Vi[2];
                const double u1 = wf[0] * Vi[1] + wf[1] * Vi[3] + wf[2] * Vi[4];
                const double u2 = wf[0] * Vi[2] + wf[1] * Vi[4] + wf[2] * Vi[5];
                sff -= u0 * wf[0] + u1 * wf[1] + u2 * wf[2];
                rf -= u0 * g[0] + u1 * g[1] + u2 * g[2];
            }
            S.at(fsky, fsky) += sff; rhs[fsky] += rf;
        }
        t_schur += now_s() - t0; t0 = now_s();
        // ---- exact solve of the reduced system
        if (nf > 0) { if (!S.factor()) { t_chol += now_s() - t0; return false; } S.solve(rhs.data()); }
        t_chol += now_s() - t0; t0 = now_s();
        for (int i = 0; i < nf; i++) y[i] = rhs[sky_of_x[i]];
        // ---- back-substitution
#pragma omp parallel for schedule(static)
        for (int p = 0; p < nP; p++) {
            if (pt_idx[p] < 0) continue;
            double b[3] = {gp[p * 3], gp[p * 3 + 1], gp[p * 3 + 2]};
            for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                double Jf[2], Jc[2][6], Jp[2][3]; scaled(j, p, scale, Jf, Jc, Jp);
                double m[2] = {0, 0};
                const int c6 = ob_cam[j] * 6;
                for (int a = 0; a < 2; a++) {
                    if (focal_free) m[a] += Jf[a] * y[focal_idx];
                    for (int k = 0; k < 6; k++) { int i = cam_idx[c6 + k]; if (i >= 0) m[a] += Jc[a][k] * y[i]; }
                }
                for (int k = 0; k < 3; k++) b[k] -= Jp[0][k] * m[0] + Jp[1][k] * m[1];
            }
            const double* Vi = &Vinv[p * 6]; double* yp = &y[pt_idx[p]];
            yp[0] = Vi[0] * b[0] + Vi[1] * b[1] + Vi[2] * b[2];
            yp[1] = Vi[1] * b[0] + Vi[3] * b[1] + Vi[4] * b[2];
            yp[2] = Vi[2] * b[0] + Vi[4] * b[1] + Vi[5] * b[2];
        }
        t_schur += now_s() - t0;
        return true;
    }

    double model_cost_change(const double* scale, const double* step) override {
        const int nP = (int)used_pt.size();
        std::vector<double> part(nP);
#pragma omp parallel for schedule(static)
        for (int p = 0; p < nP; p++) {
            double acc = 0;
            for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                double Jf[2], Jc[2][6], Jp[2][3]; scaled(j, p, scale, Jf, Jc, Jp);
                const int c6 = ob_cam[j] * 6;
                for (int a = 0; a < 2; a++) {
                    double m = 0;
                    if (focal_free) m += Jf[a] * step[focal_idx];
                    for (int k = 0; k < 6; k++) { int i = cam_idx[c6 + k]; if (i >= 0) m += Jc[a][k] * step[i]; }
                    if (pt_idx[p] >= 0) for (int k = 0; k < 3; k++) m += Jp[a][k] * step[pt_idx[p] + k];
                    acc += m * (lin[j].r[a] + 0.5 * m);
                }
            }
            part[p] = acc;
        }
        double s = 0; for (int p = 0; p < nP; p++) s += part[p];
        return -s;
    }

    void plus(const double* x, const double* delta, double* out) override {
        for (int i = 0; i < nx; i++) out[i] = x[i] + delta[i];
    }

    // returns false when nothing entered the problem (src/sfm.cpp:265-268)
    bool flatten(const oracle_ba_problem& P, std::vector<uint8_t>* obs_used_out) {
        Nc = P.num_cameras;
        const int Np = P.num_points; const int64_t M = P.num_observations;
        focal_free = !P.focal_fixed; focal0 = *P.focal;
        cam0.assign(P.cameras, P.cameras + (size_t)Nc * 6);
        // last observation wins for a repeated (camera, point) key, as with std::map assignment (sfm.cpp:140)
        std::vector<int64_t> order(M);
        for (int64_t i = 0; i < M; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            if (P.obs_pt[a] != P.obs_pt[b]) return P.obs_pt[a] < P.obs_pt[b];
            return P.obs_cam[a] < P.obs_cam[b]; });
        if (obs_used_out) obs_used_out->assign(M, 0);
        pt_start.assign(1, 0);
        int64_t i = 0;
        while (i < M) {
            const int p = P.obs_pt[order[i]];
            int64_t e = i; std::vector<int64_t> keep;
            while (e < M && P.obs_pt[order[e]] == p) {
                int64_t last = e;
                while (last + 1 < M && P.obs_pt[order[last + 1]] == p && P.obs_cam[order[last + 1]] == P.obs_cam[order[e]]) last++;
                keep.push_back(order[last]); e = last + 1;
            }
            const double* X = &P.points[(size_t)p * 3];
            const bool valid = p >= 0 && p < Np && (X[0] * X[0] + X[1] * X[1] + X[2] * X[2]) != 0.0 && keep.size() >= 3;
            if (valid) {
                for (int64_t o : keep) {
                    ob_cam.push_back(P.obs_cam[o]); ob_orig.push_back((int)o); ob_pt.push_back((int)used_pt.size());
                    ob_x.push_back(P.obs_xy[2 * o]); ob_y.push_back(P.obs_xy[2 * o + 1]);
                    if (obs_used_out) (*obs_used_out)[o] = 1;
                }
                used_pt.push_back(p); pt_start.push_back((int)ob_cam.size());
                pt0.push_back(X[0]); pt0.push_back(X[1]); pt0.push_back(X[2]);
            }
            i = e;
        }
        if (ob_cam.empty()) return false;
        const int nP = (int)used_pt.size(), Mu = (int)ob_cam.size();
        // camera-major lists (observations of a camera in point order)
        cam_start.assign(Nc + 1, 0);
        for (int j = 0; j < Mu; j++) cam_start[ob_cam[j] + 1]++;
        for (int c = 0; c < Nc; c++) cam_start[c + 1] += cam_start[c];
        cam_obs.resize(Mu); { std::vector<int> fill(cam_start.begin(), cam_start.end() - 1);
            for (int j = 0; j < Mu; j++) cam_obs[fill[ob_cam[j]]++] = j; }
        // parameter layout; constant blocks leave the program (sfm.cpp:222-225)
        nx = 0; focal_idx = -1;
        if (focal_free) focal_idx = nx++;
        cam_idx.assign((size_t)Nc * 6, -1);
        for (int c = 0; c < Nc; c++) {
            if (cam_start[c + 1] == cam_start[c]) continue;     // camera not in the problem
            if (!(P.trans_fixed && P.trans_fixed[c])) for (int k = 0; k < 3; k++) cam_idx[c * 6 + k] = nx++;
            if (!(P.rot_fixed && P.rot_fixed[c])) for (int k = 0; k < 3; k++) cam_idx[c * 6 + 3 + k] = nx++;
        }
        nf = nx;
        pt_idx.assign(nP, -1);
        for (int p = 0; p < nP; p++) if (!(P.pt_fixed && P.pt_fixed[used_pt[p]])) { pt_idx[p] = nx; nx += 3; }
        lin.resize(Mu); Vinv.assign((size_t)nP * 6, 0); gp.assign((size_t)nP * 3, 0); Wfp.assign((size_t)nP * 3, 0);
        // ---- elimination order of the camera blocks + envelope
        std::vector<std::vector<int>> adj(Nc);
        for (int p = 0; p < nP; p++)
            for (int a = pt_start[p]; a < pt_start[p + 1]; a++)
                for (int b = pt_start[p]; b < pt_start[p + 1]; b++)
                    if (a != b) adj[ob_cam[a]].push_back(ob_cam[b]);
        for (auto& v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
        std::vector<int> order_c = rcm_order(adj);
        cam_pos.assign(Nc, 0); for (int k = 0; k < Nc; k++) cam_pos[order_c[k]] = k;
        sky_of_x.assign(nf, -1);
        int ns = 0; std::vector<int> cam_first(Nc, 0);
        for (int k = 0; k < Nc; k++) { int c = order_c[k]; cam_first[c] = ns; for (int d = 0; d < 6; d++) if (cam_idx[c * 6 + d] >= 0) sky_of_x[cam_idx[c * 6 + d]] = ns++; }
        if (focal_free) sky_of_x[focal_idx] = ns++;          // dense border row goes last
        std::vector<int> first(ns);
        for (int c = 0; c < Nc; c++) {
            int f = cam_first[c];
            for (int c2 : adj[c]) f = std::min(f, cam_first[c2]);
            for (int d = 0; d < 6; d++) if (cam_idx[c * 6 + d] >= 0) first[sky_of_x[cam_idx[c * 6 + d]]] = f;
        }
        if (focal_free) first[ns - 1] = 0;
        S.init(first); rhs.assign(ns, 0.0);
        return true;
    }

    void initial_x(std::vector<double>& x) const {
        x.assign(nx, 0.0);
        if (focal_free) x[focal_idx] = focal0;
        for (size_t i = 0; i < cam_idx.size(); i++) if (cam_idx[i] >= 0) x[cam_idx[i]] = cam0[i];
        for (size_t p = 0; p < pt_idx.size(); p++) if (pt_idx[p] >= 0) for (int k = 0; k < 3; k++) x[pt_idx[p] + k] = pt0[p * 3 + k];
    }
    void scatter(const std::vector<double>& x, oracle_ba_problem& P) const {
        if (focal_free) *P.focal = x[focal_idx];
        for (size_t i = 0; i < cam_idx.size(); i++) if (cam_idx[i] >= 0) P.cameras[i] = x[cam_idx[i]];
        for (size_t p = 0; p < pt_idx.size(); p++) if (pt_idx[p] >= 0) for (int k = 0; k < 3; k++) P.points[(size_t)used_pt[p] * 3 + k] = x[pt_idx[p] + k];
    }
};

static LMOptions to_lm(const oracle_lm_options& o) {
    LMOptions l;
    l.max_num_iterations = o.max_num_iterations;
    l.max_num_consecutive_invalid_steps = o.max_num_consecutive_invalid_steps;
    l.function_tolerance = o.function_tolerance; l.gradient_tolerance = o.gradient_tolerance;
    l.parameter_tolerance = o.parameter_tolerance;
    l.initial_trust_region_radius = o.initial_trust_region_radius;
    l.max_trust_region_radius = o.max_trust_region_radius; l.min_trust_region_radius = o.min_trust_region_radius;
    l.min_lm_diagonal = o.min_lm_diagonal; l.max_lm_diagonal = o.max_lm_diagonal;
    l.min_relative_decrease = o.min_relative_decrease; l.jacobi_scaling = o.jacobi_scaling != 0; l.verbose = o.verbose;
    return l;
}

}  // namespace oracle

using namespace oracle;

extern "C" void oracle_ba_default_options(oracle_lm_options* o) {
    o->max_num_iterations = 2000;                    // src/sfm.cpp:205
    o->max_num_consecutive_invalid_steps = 100;      // src/sfm.cpp:206
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->min_relative_decrease = 1e-3;
    o->loss_type = 1; o->loss_scale = 1.0;           // src/sfm.cpp:196
    o->jacobi_scaling = 1;
    o->num_threads = 16;                             // src/sfm.cpp:209
    o->verbose = 0;
}

static int set_threads(int want) {
    int hw = omp_get_num_procs();
    int t = std::max(1, std::min(want > 0 ? want : hw, hw));
    omp_set_num_threads(t);
    return t;
}

extern "C" int oracle_ba_solve(oracle_ba_problem* p, const oracle_lm_options* o, oracle_summary* s) {
    std::memset(s, 0, sizeof(*s));
    const double t0 = now_s();
    s->threads_used = set_threads(o->num_threads);
    if (p->num_cameras == 0 || p->num_points == 0) { s->termination = 3; return 0; }   // sfm.cpp:230
    BAOracle B; B.loss_type = o->loss_type; B.loss_a = o->loss_scale;
    if (!B.flatten(*p, nullptr)) { s->termination = 3; s->t_flatten_s = now_s() - t0; return 0; }   // sfm.cpp:265-268
    s->t_flatten_s = now_s() - t0;
    s->num_residual_blocks = (int64_t)B.ob_cam.size(); s->num_points_used = (int)B.used_pt.size();
    std::vector<double> x; B.initial_x(x);
    LMSummary r = lm_minimize(B, to_lm(*o), x.data());
    B.scatter(x, *p);
    s->termination = r.termination; s->iterations = r.iterations;
    s->num_successful_steps = r.num_successful_steps; s->num_unsuccessful_steps = r.num_unsuccessful_steps;
    s->num_linear_solves = r.num_linear_solves; s->initial_cost = r.initial_cost; s->final_cost = r.final_cost;
    s->t_linearize_s = B.t_lin; s->t_schur_s = B.t_schur; s->t_cholesky_s = B.t_chol; s->t_cost_s = B.t_cost;
    s->t_total_s = now_s() - t0;
    return 0;
}

extern "C" int oracle_ba_evaluate(const oracle_ba_problem* p, const oracle_lm_options* o, int32_t raw,
                                  double* cost, double* residuals, double* jacobians, uint8_t* obs_used) {
    set_threads(o->num_threads);
    BAOracle B; B.loss_type = raw ? 0 : o->loss_type; B.loss_a = o->loss_scale;
    std::vector<uint8_t> used;
    const int64_t M = p->num_observations;
    if (residuals) std::fill(residuals, residuals + 2 * M, 0.0);
    if (jacobians) std::fill(jacobians, jacobians + 20 * M, 0.0);
    if (!B.flatten(*p, &used)) { if (cost) *cost = 0; if (obs_used) std::fill(obs_used, obs_used + M, 0); return 1; }
    if (obs_used) std::copy(used.begin(), used.end(), obs_used);
    std::vector<double> x, g(B.nx); B.initial_x(x);
    double c = 0; B.linearize(x.data(), &c, g.data());
    if (raw) {   // cost still reported with the configured loss
        BAOracle C; C.loss_type = o->loss_type; C.loss_a = o->loss_scale; C.flatten(*p, nullptr);
        std::vector<double> xc; C.initial_x(xc); C.cost_only(xc.data(), &c);
    }
    if (cost) *cost = c;
    for (size_t j = 0; j < B.ob_cam.size(); j++) {
        const int64_t o_ = B.ob_orig[j]; const ObsLin& L = B.lin[j];
        for (int a = 0; a < 2; a++) {
            if (residuals) residuals[2 * o_ + a] = L.r[a];
            if (jacobians) {
                double* J = &jacobians[20 * o_ + 10 * a];
                J[0] = L.Jf[a]; for (int k = 0; k < 6; k++) J[1 + k] = L.Jc[a][k]; for (int k = 0; k < 3; k++) J[7 + k] = L.Jp[a][k];
            }
        }
    }
    return 0;
}

// Dense reduced camera system of the UNSCALED robustified Jacobian at the given state, in a fixed layout
// (index 0 = focal, 1 + 6*c + k = camera c dof k; constant / absent parameters give zero rows):
//   S = F^T F - F^T E (E^T E + mu I)^-1 E^T F,   rhs = F^T r - F^T E (E^T E + mu I)^-1 E^T r.
// It is a plain sum over points, which is what the multi-GPU sharding relies on (tests/test_multirank_cpu.py).
extern "C" int oracle_ba_reduced_system(const oracle_ba_problem* p, const oracle_lm_options* o, double mu, double* S_out, double* rhs_out) {
    set_threads(1);
    const int Nc = p->num_cameras, n = 6 * Nc + 1;
    std::fill(S_out, S_out + (size_t)n * n, 0.0); std::fill(rhs_out, rhs_out + n, 0.0);
    BAOracle B; B.loss_type = o->loss_type; B.loss_a = o->loss_scale;
    if (!B.flatten(*p, nullptr)) return 1;
    std::vector<double> x, g(B.nx); B.initial_x(x);
    double c = 0; B.linearize(x.data(), &c, g.data());
    const int nP = (int)B.used_pt.size();
    for (int q = 0; q < nP; q++) {
        const int j0 = B.pt_start[q], K = B.pt_start[q + 1] - j0;
        const bool pfree = B.pt_idx[q] >= 0;
        double V[9] = {mu, 0, 0, 0, mu, 0, 0, 0, mu}, gpv[3] = {0, 0, 0};
        std::vector<double> F((size_t)K * 2 * 7), E((size_t)K * 2 * 3);   // per residual row: [Jf, Jc(6)] and Jp(3)
        for (int k = 0; k < K; k++) {
            const ObsLin& L = B.lin[j0 + k]; const int c6 = B.ob_cam[j0 + k] * 6;
            for (int a = 0; a < 2; a++) {
                double* f = &F[((size_t)k * 2 + a) * 7]; double* e = &E[((size_t)k * 2 + a) * 3];
                f[0] = B.focal_free ? L.Jf[a] : 0.0;
                for (int d = 0; d < 6; d++) f[1 + d] = B.cam_idx[c6 + d] >= 0 ? L.Jc[a][d] : 0.0;
                for (int d = 0; d < 3; d++) e[d] = pfree ? L.Jp[a][d] : 0.0;
                for (int u = 0; u < 3; u++) { gpv[u] += e[u] * L.r[a]; for (int v = 0; v < 3; v++) V[u * 3 + v] += e[u] * e[v]; }
            }
        }
        const double Vs[6] = {V[0], V[1], V[2], V[4], V[5], V[8]};
        double Vi6[6];
        { const double c00 = Vs[3] * Vs[5] - Vs[4] * Vs[4], c01 = Vs[2] * Vs[4] - Vs[1] * Vs[5], c02 = Vs[1] * Vs[4] - Vs[2] * Vs[3];
          const double id = 1.0 / (Vs[0] * c00 + Vs[1] * c01 + Vs[2] * c02);
          Vi6[0] = c00 * id; Vi6[1] = c01 * id; Vi6[2] = c02 * id; Vi6[3] = (Vs[0] * Vs[5] - Vs[2] * Vs[2]) * id;
          Vi6[4] = (Vs[1] * Vs[2] - Vs[0] * Vs[4]) * id; Vi6[5] = (Vs[0] * Vs[3] - Vs[1] * Vs[1]) * id; }
        const double Vi[9] = {Vi6[0], Vi6[1], Vi6[2], Vi6[1], Vi6[3], Vi6[4], Vi6[2], Vi6[4], Vi6[5]};
        // global column index of local f-column (k, d)
        auto col = [&](int k, int d) { return d == 0 ? 0 : 1 + B.ob_cam[j0 + k] * 6 + (d - 1); };
        // W_k = sum_a f_row^T e_row (7x3) per observation
        std::vector<double> W((size_t)K * 21, 0.0);
        for (int k = 0; k < K; k++)
            for (int a = 0; a < 2; a++) {
                const double* f = &F[((size_t)k * 2 + a) * 7]; const double* e = &E[((size_t)k * 2 + a) * 3];
                const double r = B.lin[j0 + k].r[a];
                for (int d = 0; d < 7; d++) {
                    rhs_out[col(k, d)] += f[d] * r;
                    for (int d2 = 0; d2 < 7; d2++) S_out[(size_t)col(k, d) * n + col(k, d2)] += f[d] * f[d2];
                    for (int u = 0; u < 3; u++) W[(size_t)k * 21 + d * 3 + u] += f[d] * e[u];
                }
            }
        for (int k = 0; k < K; k++)
            for (int d = 0; d < 7; d++) {
                double T[3];
                for (int u = 0; u < 3; u++) T[u] = W[(size_t)k * 21 + d * 3] * Vi[u] + W[(size_t)k * 21 + d * 3 + 1] * Vi[3 + u] + W[(size_t)k * 21 + d * 3 + 2] * Vi[6 + u];
                rhs_out[col(k, d)] -= T[0] * gpv[0] + T[1] * gpv[1] + T[2] * gpv[2];
                for (int k2 = 0; k2 < K; k2++)
                    for (int d2 = 0; d2 < 7; d2++)
                        S_out[(size_t)col(k, d) * n + col(k2, d2)] -= T[0] * W[(size_t)k2 * 21 + d2 * 3] + T[1] * W[(size_t)k2 * 21 + d2 * 3 + 1] + T[2] * W[(size_t)k2 * 21 + d2 * 3 + 2];
            }
    }
    return 0;
}

// The reference's own problem-building loop (src/sfm.cpp:240-263) on its own storage (std::map-based SparseVector / SparseMatrix,
// include/sphericalsfm/sparse.hpp): for every point, every camera is probed twice.  Timed only -- it quantifies the host overhead
// of SfM::Optimize() that the port's O(M) flatten does not have.  Returns the number of residual blocks it would add.
extern "C" int64_t oracle_reference_style_flatten(const oracle_ba_problem* p, double* seconds) {
    std::map<int, std::array<double, 6>> cameras; std::map<int, std::array<double, 3>> points;
    std::map<int, std::map<int, std::array<double, 2>>> observations;                 // [camera][point]
    for (int i = 0; i < p->num_cameras; i++) { std::array<double, 6> c; for (int k = 0; k < 6; k++) c[k] = p->cameras[6 * (size_t)i + k]; cameras[i] = c; }
    for (int j = 0; j < p->num_points; j++) { std::array<double, 3> X; for (int k = 0; k < 3; k++) X[k] = p->points[3 * (size_t)j + k]; points[j] = X; }
    for (int64_t o = 0; o < p->num_observations; o++) observations[p->obs_cam[o]][p->obs_pt[o]] = {p->obs_xy[2 * o], p->obs_xy[2 * o + 1]};
    auto exists2 = [&](int r, int c) { auto it = observations.find(r); if (it == observations.end()) return false; return it->second.find(c) != it->second.end(); };
    const double t0 = now_s();
    int64_t added = 0;
    for (int j = 0; j < p->num_points; j++) {
        auto pit = points.find(j); if (pit == points.end()) continue;
        const auto& X = pit->second; if (std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]) == 0) continue;
        int nobs = 0;
        for (int i = 0; i < p->num_cameras; i++) { if (cameras.find(i) == cameras.end()) continue; if (!exists2(i, j)) continue; nobs++; }
        if (nobs < 3) continue;
        for (int i = 0; i < p->num_cameras; i++) { if (cameras.find(i) == cameras.end()) continue; if (!exists2(i, j)) continue; added++; }
    }
    if (seconds) *seconds = now_s() - t0;
    return added;
}

extern "C" void oracle_so3exp(const double r[3], double R[9]) { so3exp(r, R); }
extern "C" void oracle_so3ln(const double R[9], double r[3]) { so3ln(R, r); }
extern "C" void oracle_angle_axis_rotate_point(const double aa[3], const double pt[3], double out[3]) { AngleAxisRotatePoint<double>(aa, pt, out); }
extern "C" void oracle_angle_axis_to_rotation_matrix(const double aa[3], double R[9]) { AngleAxisToRotationMatrix<double>(aa, R); }
extern "C" void oracle_rotation_matrix_to_angle_axis(const double R[9], double aa[3]) { RotationMatrixToAngleAxis<double>(R, aa); }
