// Lab: per-phase cycle stamps of k_band_chol_v2 (workgroup 0) on the config-2-shaped system.  Build via scripts/lab/make_stamped.py +
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -I../../spherical_sfm_amd/csrc -I../../include chol_stamps.hip -o chol_stamps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include "band_kernels2_stamped.h"
using namespace ssfm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    constexpr int DC = 6, BB = 36, NR = 2; const int ncomp = 4, ncam = 75, b = 10, N = ncomp * ncam, W = b + 1, n = N * DC, nw = 9;
    std::mt19937_64 rng(7); std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> band((size_t)N * W * BB, 0.0), Y((size_t)NR * n);
    for (int i = 0; i < N; i++) { for (int d = 1; d <= b && (i % ncam) - d >= 0; d++) for (int e = 0; e < BB; e++) band[((size_t)i * W + d) * BB + e] = U(rng);
                                  for (int r = 0; r < DC; r++) for (int q = 0; q < DC; q++) band[((size_t)i * W) * BB + r * DC + q] = (r == q) ? 200.0 : 0.1; }
    for (auto& v : Y) v = U(rng);
    std::vector<int> comp(ncomp + 1); for (int c = 0; c <= ncomp; c++) comp[c] = c * ncam;
    std::vector<int> pairs; for (int ir = 1; ir <= b; ir++) for (int kr = 1; kr <= ir; kr++) pairs.push_back(ir | (kr << 16));
    double *dband, *dG, *dY; int *dp, *dc, *df; long long* ddbg;
    CK(hipMalloc(&dband, band.size() * 8)); CK(hipMalloc(&dG, (size_t)N * BB * 8)); CK(hipMalloc(&dY, Y.size() * 8)); CK(hipMalloc(&dp, pairs.size() * 4 + 8));
    CK(hipMalloc(&dc, comp.size() * 4)); CK(hipMalloc(&df, 4)); CK(hipMalloc(&ddbg, (size_t)ncam * nw * 4 * 8));
    CK(hipMemcpy(dp, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, comp.data(), comp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(df, 0, 4));
    const size_t lds = ((size_t)(b + 1) * W * BB + (size_t)b * BB + (size_t)(b + 1) * NR * DC + NR * DC + 2 * BB) * 8 + ((size_t)b * (b + 1) / 2 + 4) * 4 + (size_t)ncam * nw * 4 * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2<DC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int remap = getenv("REMAP") ? atoi(getenv("REMAP")) : 1;
    const CholWaveMap wm = remap ? chol_wave_map(nw, nw - 4, 4) : chol_wave_map(0, 0, 0);
    printf("wave map:"); for (int i = 0; i < nw; i++) printf(" p%d->role %d", i, wm.v[i]); printf("\n");
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemcpy(dband, band.data(), band.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k_band_chol_v2<DC, 2>), dim3(ncomp), dim3(nw * 64), lds, 0, dband, dG, dY, dp, dc, dc + 1, dc + 1, (const int*)nullptr, N, b, df, wm, (const int*)nullptr, (const int*)nullptr, (int*)nullptr, 0, ddbg);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> dbg((size_t)ncam * nw * 4); CK(hipMemcpy(dbg.data(), ddbg, dbg.size() * 8, hipMemcpyDeviceToHost));
    const char* role[10] = {"look-ahead", "trail", "trail", "trail", "trail", "trail(rhs)", "loader", "loader", "writer", ""};
    for (int j = 30; j < 33; j++) {
        const long long t0 = dbg[((size_t)j * nw) * 4];
        for (int w = 0; w < nw; w++) { printf("step %d wave %d %-10s:", j, w, role[w]); for (int k = 0; k < 4; k++) printf(" %6lld", dbg[((size_t)j * nw + w) * 4 + k] - t0); printf("\n"); }
    }
    // averages over steps 10..60: phase B, barrier-1 wait, role work, barrier-2 wait (wave 0 = reference for the step length)
    double len = 0; for (int j = 10; j < 60; j++) len += dbg[((size_t)(j + 1) * nw) * 4] - dbg[((size_t)j * nw) * 4];
    printf("average step: %.0f cycles\n", len / 50);
    for (int w = 0; w < nw; w++) {
        double B = 0, w1 = 0, work = 0, w2 = 0;
        for (int j = 10; j < 60; j++) { const long long* s = &dbg[((size_t)j * nw + w) * 4]; const long long nx = dbg[((size_t)(j + 1) * nw + w) * 4];
                                        B += s[1] - s[0]; w1 += s[2] - s[1]; work += s[3] - s[2]; w2 += nx - s[3]; }
        printf("wave %d %-10s  panel %.0f  wait1 %.0f  work %.0f  wait2+loop %.0f\n", w, role[w], B / 50, w1 / 50, work / 50, w2 / 50);
    }
    return 0;
}
