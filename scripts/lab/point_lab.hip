// Lab bench for the point-side kernels of the BA iteration (k_point_lin and candidates for its replacement) on data shaped like BASELINE
// config 2 (300 cameras on a circle, 100k points, 6 observations each, stride 4): hipEvent averages, s_memtime stamps of a copy of the
// production kernel, and the outputs of every candidate against the production kernel's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I../../spherical_sfm_amd/csrc point_lab.hip -o point_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "ba_kernels.h"
using namespace ssfm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- the production kernel with stamps (lane 0 of every wave) ----
__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
template <int OBS_UNROLL>
static __global__ void __launch_bounds__(256)
k_point_lin_stamped(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
            const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
            const int* __restrict__ pt_start, int nP, const double* __restrict__ scale_pt, const double* __restrict__ scale_f,
            int loss, double la, double radius, double min_diag, double max_diag,
            double* __restrict__ Vinv, double* __restrict__ Vs, double* __restrict__ gp, double* __restrict__ Wf, double* __restrict__ scal,
            unsigned long long* __restrict__ stamps, int smask = 15) {
    __shared__ double red[5 * 4];
    unsigned long long T[8]; int nt = 0;
    T[nt++] = now();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    double gmax = 0.0;
    if (p < nP) {
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double sp[3] = {scale_pt[3 * p], scale_pt[3 * p + 1], scale_pt[3 * p + 2]};
        const double f = focal[0], sf = scale_f[0];
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, wf[3] = {0, 0, 0};
        const int js = pt_start[p], je = pt_start[p + 1];
        if (X[0] + sp[0] + f + sf + js + je == 1e300) T[7] = 1;          // forces the loads to have landed
        T[nt++] = now();
        for (int jb = js; jb < je; jb += OBS_UNROLL) {
            int cc[OBS_UNROLL]; double2 oo[OBS_UNROLL]; double tR[OBS_UNROLL][12];
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) { const int jj = min(jb + u, je - 1); cc[u] = obs_cam[jj]; oo[u] = obs_xy[jj]; }
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
#pragma unroll
                for (int k = 0; k < 3; k++) tR[u][k] = cam[6 * (size_t)cc[u] + k];
#pragma unroll
                for (int k = 0; k < 9; k++) tR[u][3 + k] = rot[27 * (size_t)cc[u] + k];
            }
            double chk = 0; for (int u = 0; u < OBS_UNROLL; u++) for (int k = 0; k < 12; k++) chk += tR[u][k];
            if (chk == 1e300) T[7] = 2;
            if (nt < 6) T[nt++] = now();
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
                const double wgt = (jb + u < je) ? 1.0 : 0.0;
                ObsPoint L; lin_obs_point(f, tR[u], tR[u] + 3, X, oo[u].x, oo[u].y, loss, la, L);
                acc[0] += wgt * L.half_rho;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const double j0 = L.Jp[a][0] * sp[0] * wgt, j1 = L.Jp[a][1] * sp[1] * wgt, j2 = L.Jp[a][2] * sp[2] * wgt, jf = L.Jf[a] * sf * wgt;
                    V[0] += j0 * j0; V[1] += j0 * j1; V[2] += j0 * j2; V[3] += j1 * j1; V[4] += j1 * j2; V[5] += j2 * j2;
                    g[0] += j0 * L.r[a]; g[1] += j1 * L.r[a]; g[2] += j2 * L.r[a];
                    wf[0] += jf * j0; wf[1] += jf * j1; wf[2] += jf * j2;
                    acc[1] += jf * jf; acc[2] += jf * L.r[a];
                }
            }
            if (V[0] == 1e300) T[7] = 3;
            if (nt < 6) T[nt++] = now();
        }
        while (nt < 6) T[nt++] = now();
        if (sp[0] > 0.0) {
            gmax = fmax(fabs(g[0] / sp[0]), fmax(fabs(g[1] / sp[1]), fabs(g[2] / sp[2])));
            V[0] += fmin(fmax(V[0], min_diag), max_diag) / radius;
            V[3] += fmin(fmax(V[3], min_diag), max_diag) / radius;
            V[5] += fmin(fmax(V[5], min_diag), max_diag) / radius;
        } else { V[0] = V[3] = V[5] = 1.0; }
        double Vi[6]; sym3_inverse(V, Vi);
        const double u0 = wf[0] * Vi[0] + wf[1] * Vi[1] + wf[2] * Vi[2];
        const double u1 = wf[0] * Vi[1] + wf[1] * Vi[3] + wf[2] * Vi[4];
        const double u2 = wf[0] * Vi[2] + wf[1] * Vi[4] + wf[2] * Vi[5];
        acc[3] = u0 * wf[0] + u1 * wf[1] + u2 * wf[2];
        acc[4] = u0 * g[0] + u1 * g[1] + u2 * g[2];
        if (smask & 1) for (int k = 0; k < 6; k++) Vinv[6 * p + k] = Vi[k];
        double* ps = Vs + 12 * (size_t)p;
        if (smask & 2) {
        ps[0] = Vi[0] * sp[0] * sp[0]; ps[1] = Vi[1] * sp[0] * sp[1]; ps[2] = Vi[2] * sp[0] * sp[2];
        ps[3] = Vi[3] * sp[1] * sp[1]; ps[4] = Vi[4] * sp[1] * sp[2]; ps[5] = Vi[5] * sp[2] * sp[2];
        ps[6] = sp[0] * (Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2]); ps[7] = sp[1] * (Vi[1] * g[0] + Vi[3] * g[1] + Vi[4] * g[2]);
        ps[8] = sp[2] * (Vi[2] * g[0] + Vi[4] * g[1] + Vi[5] * g[2]);
        ps[9] = sp[0] * u0; ps[10] = sp[1] * u1; ps[11] = sp[2] * u2; }
        if (smask & 4) for (int k = 0; k < 3; k++) gp[3 * p + k] = g[k];
        if (smask & 8) for (int k = 0; k < 3; k++) Wf[3 * p + k] = wf[k];
        if (smask == 0 && Vi[0] + Vi[4] + u0 == 1e300) gp[0] = 1;
    }
    T[6] = now();
    block_sum<5>(acc, red);
    gmax = wave_max(gmax);
    double* sl = scal_slot(scal);
    if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&sl[SC_GMAX], gmax);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(&sl[SC_COST], acc[0]); unsafeAtomicAdd(&sl[SC_FJJ], acc[1]); unsafeAtomicAdd(&sl[SC_FJR], acc[2]);
        unsafeAtomicAdd(&sl[SC_FWW], acc[3]); unsafeAtomicAdd(&sl[SC_FWG], acc[4]);
    }
    T[7] = now();
    if ((threadIdx.x & 63) == 0) { unsigned long long* o = stamps + 8 * (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)); for (int k = 0; k < 8; k++) o[k] = T[k]; }
}


// ---- candidate: no workgroup barrier at the end (every wave folds its own sums and issues its own atomics), any workgroup size ----
template <int OBS_UNROLL, int NT>
static __global__ void __launch_bounds__(256)
k_point_lin_v2(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
            const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
            const int* __restrict__ pt_start, int nP, const double* __restrict__ scale_pt, const double* __restrict__ scale_f,
            int loss, double la, double radius, double min_diag, double max_diag,
            double* __restrict__ Vs, double* __restrict__ gp, double* __restrict__ scal) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    double gmax = 0.0;
    if (p < nP) {
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double sp[3] = {scale_pt[3 * p], scale_pt[3 * p + 1], scale_pt[3 * p + 2]};
        const double f = focal[0], sf = scale_f[0];
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, wf[3] = {0, 0, 0};
        const int js = pt_start[p], je = pt_start[p + 1];
        for (int jb = js; jb < je; jb += OBS_UNROLL) {
            int cc[OBS_UNROLL]; double2 oo[OBS_UNROLL]; double tR[OBS_UNROLL][12];
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) { const int jj = min(jb + u, je - 1); cc[u] = obs_cam[jj]; oo[u] = obs_xy[jj]; }
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
#pragma unroll
                for (int k = 0; k < 3; k++) tR[u][k] = cam[6 * (size_t)cc[u] + k];
#pragma unroll
                for (int k = 0; k < 9; k++) tR[u][3 + k] = rot[27 * (size_t)cc[u] + k];
            }
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
                const double wgt = (jb + u < je) ? 1.0 : 0.0;
                ObsPoint L; lin_obs_point(f, tR[u], tR[u] + 3, X, oo[u].x, oo[u].y, loss, la, L);
                acc[0] += wgt * L.half_rho;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const double j0 = L.Jp[a][0] * sp[0] * wgt, j1 = L.Jp[a][1] * sp[1] * wgt, j2 = L.Jp[a][2] * sp[2] * wgt, jf = L.Jf[a] * sf * wgt;
                    V[0] += j0 * j0; V[1] += j0 * j1; V[2] += j0 * j2; V[3] += j1 * j1; V[4] += j1 * j2; V[5] += j2 * j2;
                    g[0] += j0 * L.r[a]; g[1] += j1 * L.r[a]; g[2] += j2 * L.r[a];
                    wf[0] += jf * j0; wf[1] += jf * j1; wf[2] += jf * j2;
                    acc[1] += jf * jf; acc[2] += jf * L.r[a];
                }
            }
        }
        if (sp[0] > 0.0) {
            gmax = fmax(fabs(g[0] / sp[0]), fmax(fabs(g[1] / sp[1]), fabs(g[2] / sp[2])));
            V[0] += fmin(fmax(V[0], min_diag), max_diag) / radius;
            V[3] += fmin(fmax(V[3], min_diag), max_diag) / radius;
            V[5] += fmin(fmax(V[5], min_diag), max_diag) / radius;
        } else { V[0] = V[3] = V[5] = 1.0; }
        double Vi[6]; sym3_inverse(V, Vi);
        const double u0 = wf[0] * Vi[0] + wf[1] * Vi[1] + wf[2] * Vi[2];
        const double u1 = wf[0] * Vi[1] + wf[1] * Vi[3] + wf[2] * Vi[4];
        const double u2 = wf[0] * Vi[2] + wf[1] * Vi[4] + wf[2] * Vi[5];
        acc[3] = u0 * wf[0] + u1 * wf[1] + u2 * wf[2];
        acc[4] = u0 * g[0] + u1 * g[1] + u2 * g[2];
        double* ps = Vs + 12 * (size_t)p;
        double o[15];
        o[0] = Vi[0] * sp[0] * sp[0]; o[1] = Vi[1] * sp[0] * sp[1]; o[2] = Vi[2] * sp[0] * sp[2];
        o[3] = Vi[3] * sp[1] * sp[1]; o[4] = Vi[4] * sp[1] * sp[2]; o[5] = Vi[5] * sp[2] * sp[2];
        o[6] = sp[0] * (Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2]); o[7] = sp[1] * (Vi[1] * g[0] + Vi[3] * g[1] + Vi[4] * g[2]);
        o[8] = sp[2] * (Vi[2] * g[0] + Vi[4] * g[1] + Vi[5] * g[2]);
        o[9] = sp[0] * u0; o[10] = sp[1] * u1; o[11] = sp[2] * u2;
        o[12] = g[0]; o[13] = g[1]; o[14] = g[2];
        if (NT) { for (int k = 0; k < 12; k++) __builtin_nontemporal_store(o[k], ps + k); for (int k = 0; k < 3; k++) __builtin_nontemporal_store(o[12 + k], gp + 3 * p + k); }
        else { for (int k = 0; k < 12; k++) ps[k] = o[k]; for (int k = 0; k < 3; k++) gp[3 * p + k] = o[12 + k]; }
    }
    const double t = wave_transpose_sum(acc);
    gmax = wave_max(gmax);
    const int slot = wave_tr_index();
    double* sl = scal + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (SC_NSLOT - 1)) * SC_TOTAL;
    if (slot < 5) unsafeAtomicAdd(&sl[slot == 0 ? SC_COST : slot == 1 ? SC_FJJ : slot == 2 ? SC_FJR : slot == 3 ? SC_FWW : SC_FWG], t);
    if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&sl[SC_GMAX], gmax);
}

template <int OBS_UNROLL, int NT>
static __global__ void __launch_bounds__(256)
k_point_lin_v3(const double* __restrict__ cam, const double* __restrict__ rot, const double* __restrict__ pts,
            const double* __restrict__ focal, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam,
            const int* __restrict__ pt_start, int nP, const double* __restrict__ scale_pt, const double* __restrict__ scale_f,
            int loss, double la, double radius, double min_diag, double max_diag,
            double* __restrict__ Vs, double* __restrict__ gp, double* __restrict__ scal) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    double gmax = 0.0;
    if (p < nP) {
        const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
        const double sp[3] = {scale_pt[3 * p], scale_pt[3 * p + 1], scale_pt[3 * p + 2]};
        const double f = focal[0], sf = scale_f[0];
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, wf[3] = {0, 0, 0};
        const int js = pt_start[p], je = pt_start[p + 1];
        for (int jb = js; jb < je; jb += OBS_UNROLL) {
            int cc[OBS_UNROLL]; double2 oo[OBS_UNROLL]; double tR[OBS_UNROLL][12];
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) { const int jj = min(jb + u, je - 1); cc[u] = obs_cam[jj]; oo[u] = obs_xy[jj]; }
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
                // neighbouring points mostly see the same cameras: when the whole wave does, the table comes through the scalar cache
                // (one s_load per wave instead of 64 lanes x 96 B through the vector L1)
                const int cs = __builtin_amdgcn_readfirstlane(cc[u]);
                if (__builtin_amdgcn_ballot_w64(cc[u] != cs) == 0) {
                    const double* ct = cam + 6 * (size_t)cs; const double* cr = rot + 27 * (size_t)cs;
#pragma unroll
                    for (int k = 0; k < 3; k++) tR[u][k] = ct[k];
#pragma unroll
                    for (int k = 0; k < 9; k++) tR[u][3 + k] = cr[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) tR[u][k] = cam[6 * (size_t)cc[u] + k];
#pragma unroll
                    for (int k = 0; k < 9; k++) tR[u][3 + k] = rot[27 * (size_t)cc[u] + k];
                }
            }
#pragma unroll
            for (int u = 0; u < OBS_UNROLL; u++) {
                const double wgt = (jb + u < je) ? 1.0 : 0.0;
                ObsPoint L; lin_obs_point(f, tR[u], tR[u] + 3, X, oo[u].x, oo[u].y, loss, la, L);
                acc[0] += wgt * L.half_rho;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const double j0 = L.Jp[a][0] * sp[0] * wgt, j1 = L.Jp[a][1] * sp[1] * wgt, j2 = L.Jp[a][2] * sp[2] * wgt, jf = L.Jf[a] * sf * wgt;
                    V[0] += j0 * j0; V[1] += j0 * j1; V[2] += j0 * j2; V[3] += j1 * j1; V[4] += j1 * j2; V[5] += j2 * j2;
                    g[0] += j0 * L.r[a]; g[1] += j1 * L.r[a]; g[2] += j2 * L.r[a];
                    wf[0] += jf * j0; wf[1] += jf * j1; wf[2] += jf * j2;
                    acc[1] += jf * jf; acc[2] += jf * L.r[a];
                }
            }
        }
        if (sp[0] > 0.0) {
            gmax = fmax(fabs(g[0] / sp[0]), fmax(fabs(g[1] / sp[1]), fabs(g[2] / sp[2])));
            V[0] += fmin(fmax(V[0], min_diag), max_diag) / radius;
            V[3] += fmin(fmax(V[3], min_diag), max_diag) / radius;
            V[5] += fmin(fmax(V[5], min_diag), max_diag) / radius;
        } else { V[0] = V[3] = V[5] = 1.0; }
        double Vi[6]; sym3_inverse(V, Vi);
        const double u0 = wf[0] * Vi[0] + wf[1] * Vi[1] + wf[2] * Vi[2];
        const double u1 = wf[0] * Vi[1] + wf[1] * Vi[3] + wf[2] * Vi[4];
        const double u2 = wf[0] * Vi[2] + wf[1] * Vi[4] + wf[2] * Vi[5];
        acc[3] = u0 * wf[0] + u1 * wf[1] + u2 * wf[2];
        acc[4] = u0 * g[0] + u1 * g[1] + u2 * g[2];
        double* ps = Vs + 12 * (size_t)p;
        double o[15];
        o[0] = Vi[0] * sp[0] * sp[0]; o[1] = Vi[1] * sp[0] * sp[1]; o[2] = Vi[2] * sp[0] * sp[2];
        o[3] = Vi[3] * sp[1] * sp[1]; o[4] = Vi[4] * sp[1] * sp[2]; o[5] = Vi[5] * sp[2] * sp[2];
        o[6] = sp[0] * (Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2]); o[7] = sp[1] * (Vi[1] * g[0] + Vi[3] * g[1] + Vi[4] * g[2]);
        o[8] = sp[2] * (Vi[2] * g[0] + Vi[4] * g[1] + Vi[5] * g[2]);
        o[9] = sp[0] * u0; o[10] = sp[1] * u1; o[11] = sp[2] * u2;
        o[12] = g[0]; o[13] = g[1]; o[14] = g[2];
        if (NT) { for (int k = 0; k < 12; k++) __builtin_nontemporal_store(o[k], ps + k); for (int k = 0; k < 3; k++) __builtin_nontemporal_store(o[12 + k], gp + 3 * p + k); }
        else { for (int k = 0; k < 12; k++) ps[k] = o[k]; for (int k = 0; k < 3; k++) gp[3 * p + k] = o[12 + k]; }
    }
    const double t = wave_transpose_sum(acc);
    gmax = wave_max(gmax);
    const int slot = wave_tr_index();
    double* sl = scal + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (SC_NSLOT - 1)) * SC_TOTAL;
    if (slot < 5) unsafeAtomicAdd(&sl[slot == 0 ? SC_COST : slot == 1 ? SC_FJJ : slot == 2 ? SC_FJR : slot == 3 ? SC_FWW : SC_FWG], t);
    if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&sl[SC_GMAX], gmax);
}

static __global__ void k_empty(int* x) { if (x && threadIdx.x == 9999) x[0] = 1; }
static __global__ void k_store_only(double* __restrict__ Vs, double* __restrict__ gp, int nP) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nP) { for (int k = 0; k < 12; k++) Vs[12 * (size_t)p + k] = (double)(p + k); for (int k = 0; k < 3; k++) gp[3 * (size_t)p + k] = (double)p; }
}
static __global__ void k_load_only(const double* __restrict__ pts, const double2* __restrict__ obs_xy, const int* __restrict__ obs_cam, const int* __restrict__ pt_start, int nP, double* out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x; double a = 0;
    if (p < nP) { a = pts[3 * p] + pts[3 * p + 1] + pts[3 * p + 2]; for (int j = pt_start[p]; j < pt_start[p + 1]; j++) a += obs_xy[j].x + obs_xy[j].y + obs_cam[j]; }
    if (a == 1e300) out[0] = a;
}
static void rodrigues(const double* r, double* R) {
    const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 1e-12) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0); return; }
    const double k[3] = {r[0] / th, r[1] / th, r[2] / th}, c = std::cos(th), s = std::sin(th), v = 1 - c;
    R[0] = c + k[0] * k[0] * v; R[1] = k[0] * k[1] * v - k[2] * s; R[2] = k[0] * k[2] * v + k[1] * s;
    R[3] = k[1] * k[0] * v + k[2] * s; R[4] = c + k[1] * k[1] * v; R[5] = k[1] * k[2] * v - k[0] * s;
    R[6] = k[2] * k[0] * v - k[1] * s; R[7] = k[2] * k[1] * v + k[0] * s; R[8] = c + k[2] * k[2] * v;
}

int main(int argc, char** argv) {
    const int Nc = 300, Np = argc > 1 ? atoi(argv[1]) : 100000, K = 6, stride = 4, reps = 50;
    const int64_t M = (int64_t)Np * K;
    std::mt19937_64 rng(11); std::uniform_real_distribution<double> Uxy(-0.45, 0.45), Ud(4.0, 8.0); std::normal_distribution<double> N01(0.0, 1.0);
    std::vector<double> cam((size_t)Nc * 6), rot((size_t)Nc * 27, 0.0), pts((size_t)Np * 3), xy((size_t)M * 2), sp((size_t)Np * 3, 1.0);
    std::vector<int> oc(M), ps(Np + 1);
    for (int c = 0; c < Nc; c++) { double a = 2 * M_PI * c / Nc; if (a > M_PI) a -= 2 * M_PI; double* q = &cam[6 * c]; q[0] = 0; q[1] = 0; q[2] = -1; q[3] = 0.004 * N01(rng); q[4] = a + 0.004 * N01(rng); q[5] = 0.004 * N01(rng); rodrigues(q + 3, &rot[27 * c]); }
    for (int n = 0; n < Np; n++) {
        const int a = (int)((int64_t)n * Nc / Np); double Rg[9]; const double rg[3] = {0, 2 * M_PI * a / Nc, 0}; rodrigues(rg, Rg);
        const double d = Ud(rng), pc[3] = {Uxy(rng) * d, Uxy(rng) * d, d + 1.0};       // pc - t with t = (0,0,-1)
        double X[3]; for (int i = 0; i < 3; i++) X[i] = Rg[0 * 3 + i] * pc[0] + Rg[1 * 3 + i] * pc[1] + Rg[2 * 3 + i] * pc[2];
        int cs[K]; for (int k = 0; k < K; k++) cs[k] = ((a + stride * (k - K / 2)) % Nc + Nc) % Nc;
        std::sort(cs, cs + K);
        ps[n] = n * K;
        for (int k = 0; k < K; k++) {
            const double* R = &rot[27 * cs[k]]; const double* t = &cam[6 * cs[k]];
            double P[3]; for (int i = 0; i < 3; i++) P[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + t[i];
            oc[n * K + k] = cs[k]; xy[2 * (n * K + k)] = 1000.0 * P[0] / P[2] + 0.5 * N01(rng); xy[2 * (n * K + k) + 1] = 1000.0 * P[1] / P[2] + 0.5 * N01(rng);
        }
        for (int i = 0; i < 3; i++) { pts[3 * n + i] = X[i] * (1.0 + 0.01 * N01(rng)); sp[3 * n + i] = 1.0 / (1.0 + 30.0 + 5 * Uxy(rng)); }
    }
    ps[Np] = Np * K;
    const double focal = 1000.0, sf = 1.0 / 400.0;
    double *dcam, *drot, *dpts, *dfocal, *dsp, *dsf, *dVinv, *dVs, *dgp, *dWf, *dscal; double2* dxy; int *doc, *dps; unsigned long long* dst;
    const int nwaves = (Np + 63) / 64 + 8;
    CK(hipMalloc(&dcam, cam.size() * 8)); CK(hipMalloc(&drot, rot.size() * 8)); CK(hipMalloc(&dpts, pts.size() * 8)); CK(hipMalloc(&dfocal, 8)); CK(hipMalloc(&dsp, sp.size() * 8)); CK(hipMalloc(&dsf, 8));
    CK(hipMalloc(&dVinv, (size_t)Np * 6 * 8)); CK(hipMalloc(&dVs, (size_t)Np * 12 * 8)); CK(hipMalloc(&dgp, (size_t)Np * 3 * 8)); CK(hipMalloc(&dWf, (size_t)Np * 3 * 8)); CK(hipMalloc(&dscal, SC_NSLOT * SC_TOTAL * 8));
    CK(hipMalloc(&dxy, xy.size() * 8)); CK(hipMalloc(&doc, oc.size() * 4)); CK(hipMalloc(&dps, ps.size() * 4)); CK(hipMalloc(&dst, (size_t)nwaves * 8 * 8));
    CK(hipMemcpy(dcam, cam.data(), cam.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drot, rot.data(), rot.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dpts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfocal, &focal, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsf, &sf, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsp, sp.data(), sp.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dxy, xy.data(), xy.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(doc, oc.data(), oc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dps, ps.data(), ps.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = (Np + 255) / 256;
    auto timeit = [&](const char* tag, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; i++) launch(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-28s %7.2f us per launch\n", tag, 1e3 * ms / reps);
    };
    auto outputs = [&](std::vector<double>& o) {
        o.resize((size_t)Np * 24 + SC_TOTAL); std::vector<double> sc(SC_NSLOT * SC_TOTAL);
        CK(hipMemcpy(o.data(), dVinv, (size_t)Np * 6 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data() + (size_t)Np * 6, dVs, (size_t)Np * 12 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o.data() + (size_t)Np * 18, dgp, (size_t)Np * 3 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data() + (size_t)Np * 21, dWf, (size_t)Np * 3 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sc.data(), dscal, sc.size() * 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < SC_TOTAL; k++) { double s = 0; for (int q = 0; q < SC_NSLOT; q++) { if (k == SC_GMAX) s = std::max(s, sc[q * SC_TOTAL + k]); else s += sc[q * SC_TOTAL + k]; } o[(size_t)Np * 24 + k] = s; }
    };
    auto clear = [&]() { CK(hipMemsetAsync(dscal, 0, SC_NSLOT * SC_TOTAL * 8, st)); CK(hipMemsetAsync(dVinv, 0, (size_t)Np * 6 * 8, st)); CK(hipMemsetAsync(dVs, 0, (size_t)Np * 12 * 8, st)); };
    std::vector<double> ref, got;
    auto prod = [&]() { hipLaunchKernelGGL(k_point_lin<3>, dim3(grid), dim3(256), 0, st, dcam, drot, dpts, dfocal, dxy, doc, dps, Np, dsp, dsf, 1, 1.0, 1e4, 1e-6, 1e32, dVs, dgp, dscal, (const double*)nullptr); };
    clear(); prod(); CK(hipStreamSynchronize(st)); outputs(ref);
    printf("cost %.6f gmax %.6g\n", ref[(size_t)Np * 24 + SC_COST], ref[(size_t)Np * 24 + SC_GMAX]);
    timeit("k_point_lin<3> (production)", prod);
    auto stamped = [&]() { hipLaunchKernelGGL(k_point_lin_stamped<3>, dim3(grid), dim3(256), 0, st, dcam, drot, dpts, dfocal, dxy, doc, dps, Np, dsp, dsf, 1, 1.0, 1e4, 1e-6, 1e32, dVinv, dVs, dgp, dWf, dscal, dst, 15); };
    timeit("k_point_lin_stamped<3>", stamped);
    for (int m : {15, 7, 6, 2, 0}) {
        char tag[64]; snprintf(tag, 64, "stamped, store mask %d", m);
        timeit(tag, [&]() { hipLaunchKernelGGL(k_point_lin_stamped<3>, dim3(grid), dim3(256), 0, st, dcam, drot, dpts, dfocal, dxy, doc, dps, Np, dsp, dsf, 1, 1.0, 1e4, 1e-6, 1e32, dVinv, dVs, dgp, dWf, dscal, dst, m); });
    }
    stamped(); CK(hipStreamSynchronize(st));
    {
        std::vector<unsigned long long> T((size_t)nwaves * 8); CK(hipMemcpy(T.data(), dst, T.size() * 8, hipMemcpyDeviceToHost));
        const int nw = (Np + 63) / 64; unsigned long long t0 = ~0ull, t1 = 0; double d[8] = {0};
        for (int w = 0; w < nw; w++) { t0 = std::min(t0, T[8 * w]); t1 = std::max(t1, T[8 * w + 7]); for (int k = 1; k < 8; k++) d[k] += (double)(T[8 * w + k] - T[8 * w + k - 1]); }
        printf("stamps (cycle counter ticks, average over %d waves): first loads %.0f | group1 loads %.0f | group1 math %.0f | group2 loads %.0f | group2 math %.0f | inverse+stores %.0f | sums %.0f ; first start -> last end %llu\n",
               nw, d[1] / nw, d[2] / nw, d[3] / nw, d[4] / nw, d[5] / nw, d[6] / nw, d[7] / nw, t1 - t0);
        // start-time histogram: how late do waves start?
        std::vector<unsigned long long> starts(nw); for (int w = 0; w < nw; w++) starts[w] = T[8 * w] - t0; std::sort(starts.begin(), starts.end());
        printf("wave start offsets: median %llu, 90%% %llu, max %llu ; ", starts[nw / 2], starts[nw * 9 / 10], starts[nw - 1]);
        std::vector<unsigned long long> life(nw); for (int w = 0; w < nw; w++) life[w] = T[8 * w + 7] - T[8 * w]; std::sort(life.begin(), life.end());
        printf("wave life: median %llu, max %llu\n", life[nw / 2], life[nw - 1]);
    }
    for (int bs : {256, 64}) {
        char tag[64]; snprintf(tag, 64, "empty kernel, block %d", bs); timeit(tag, [&]() { hipLaunchKernelGGL(k_empty, dim3((Np + bs - 1) / bs), dim3(bs), 0, st, (int*)nullptr); });
        snprintf(tag, 64, "stores only (15 dbl/pt), block %d", bs); timeit(tag, [&]() { hipLaunchKernelGGL(k_store_only, dim3((Np + bs - 1) / bs), dim3(bs), 0, st, dVs, dgp, Np); });
        snprintf(tag, 64, "loads only, block %d", bs); timeit(tag, [&]() { hipLaunchKernelGGL(k_load_only, dim3((Np + bs - 1) / bs), dim3(bs), 0, st, dpts, dxy, doc, dps, Np, dVs); });
    }
    for (int nt = 0; nt < 1; nt++) for (int bs : {256, 64}) {
        char tag[64]; snprintf(tag, 64, "v2 (no barrier), block %d nt %d", bs, nt);
        auto v2 = [&]() { if (nt) hipLaunchKernelGGL((k_point_lin_v2<3, 1>), dim3((Np + bs - 1) / bs), dim3(bs), 0, st, dcam, drot, dpts, dfocal, dxy, doc, dps, Np, dsp, dsf, 1, 1.0, 1e4, 1e-6, 1e32, dVs, dgp, dscal);
                          else hipLaunchKernelGGL((k_point_lin_v2<3, 0>), dim3((Np + bs - 1) / bs), dim3(bs), 0, st, dcam, drot, dpts, dfocal, dxy, doc, dps, Np, dsp, dsf, 1, 1.0, 1e4, 1e-6, 1e32, dVs, dgp, dscal); };
        clear(); v2(); CK(hipStreamSynchronize(st)); outputs(got);
        double worst = 0; for (size_t i = (size_t)Np * 6; i < got.size(); i++) { if (i >= (size_t)Np * 21 && i < (size_t)Np * 24) continue; worst = std::max(worst, std::fabs(got[i] - ref[i]) / (1e-300 + std::max(std::fabs(ref[i]), 1.0))); }
        printf("max rel difference to production %.3g ; ", worst);
        timeit(tag, v2);
    }
    for (int bs : {256, 64}) {
        char tag[64]; snprintf(tag, 64, "v3 (scalar camera tables), block %d", bs);
        auto v3 = [&]() { hipLaunchKernelGGL((k_point_lin_v3<3, 0>), dim3((Np + bs - 1) / bs), dim3(bs), 0, st, dcam, drot, dpts, dfocal, dxy, doc, dps, Np, dsp, dsf, 1, 1.0, 1e4, 1e-6, 1e32, dVs, dgp, dscal); };
        clear(); v3(); CK(hipStreamSynchronize(st)); outputs(got);
        double worst = 0; for (size_t i = (size_t)Np * 6; i < got.size(); i++) { if (i >= (size_t)Np * 21 && i < (size_t)Np * 24) continue; worst = std::max(worst, std::fabs(got[i] - ref[i]) / (1e-300 + std::max(std::fabs(ref[i]), 1.0))); }
        printf("max rel difference to production %.3g ; ", worst);
        timeit(tag, v3);
    }
    // ---- k_point_backsub: group sizes x workgroup sizes ----
    {
        hipLaunchKernelGGL(k_cam_rot, dim3((Nc + 63) / 64), dim3(64), 0, st, dcam, drot, Nc);
        clear(); prod(); CK(hipStreamSynchronize(st));               // Vs, gp of the current state
        std::vector<double> y((size_t)Nc * 6 + 1), sc((size_t)Nc * 6);
        for (auto& v : y) v = 1e-3 * N01(rng); for (auto& v : sc) v = 1.0 / (1.0 + 200.0 + 20 * Uxy(rng));
        double *dy, *dsc, *dptsc, *dfc; CK(hipMalloc(&dy, y.size() * 8)); CK(hipMalloc(&dsc, sc.size() * 8)); CK(hipMalloc(&dptsc, pts.size() * 8)); CK(hipMalloc(&dfc, 8));
        CK(hipMemcpy(dy, y.data(), y.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), sc.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dfc, &focal, 8, hipMemcpyHostToDevice));
        std::vector<double> pref, pgot(pts.size());
        auto run_bs = [&](auto kernel, int bs, const char* tag) {
            auto l = [&]() { hipLaunchKernelGGL(kernel, dim3((Np + bs - 1) / bs), dim3(bs), 0, st, dcam, drot, dpts, dfocal, dxy, doc, dps, Np, dsc, dsp, dsf, dVs, dgp, dy, Nc, 1, 1.0, dcam, drot, dfc, dptsc, dscal,
                                                   (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0.0, (double*)nullptr, (double*)nullptr); };
            CK(hipMemsetAsync(dscal, 0, SC_NSLOT * SC_TOTAL * 8, st)); l(); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(pgot.data(), dptsc, pgot.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> scv(SC_NSLOT * SC_TOTAL); CK(hipMemcpy(scv.data(), dscal, scv.size() * 8, hipMemcpyDeviceToHost));
            double model = 0, cand = 0; for (int q = 0; q < SC_NSLOT; q++) { model += scv[q * SC_TOTAL + SC_MODEL]; cand += scv[q * SC_TOTAL + SC_CAND_COST]; }
            if (pref.empty()) pref = pgot;
            double worst = 0; for (size_t i = 0; i < pgot.size(); i++) worst = std::max(worst, std::fabs(pgot[i] - pref[i]));
            printf("model %.9g cand %.9g max |dX| vs first %.2g ; ", model, cand, worst);
            timeit(tag, l);
        };
        run_bs(k_point_backsub<6>, 256, "backsub block 256"); run_bs(k_point_backsub<6>, 64, "backsub block 64");
    }
    return 0;
}
