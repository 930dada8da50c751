// Lab bench for the block-banded Cholesky kernels: random SPD block-banded systems shaped like the reduced camera system of
// BASELINE config 2 (4 rings x 75 cameras, band 12, 6x6 blocks), second-generation kernels over wave counts, checked against a dense
// CPU Cholesky.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../spherical_sfm_amd/csrc chol_lab.hip -o chol_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHOL3_STAMPS 1
#include "ba_flatten.h"
#include "band_kernels3.h"
#include "band_kernels2p.h"
using namespace ssfm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int DC>
int run(int ncomp, int ncam, int b, int reps) {
    constexpr int BB = DC * DC, NR = 2;
    const int N = ncomp * ncam, W = b + 1, n = N * DC;
    std::mt19937_64 rng(7); std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> band((size_t)N * W * BB, 0.0), Y((size_t)NR * n);
    std::vector<int> comp_ptr(ncomp + 1);
    for (int c = 0; c <= ncomp; c++) comp_ptr[c] = c * ncam;
    // dense per component for the reference
    std::vector<std::vector<double>> dense(ncomp, std::vector<double>((size_t)ncam * DC * ncam * DC, 0.0));
    for (int c = 0; c < ncomp; c++) {
        const int m = ncam * DC; auto& A = dense[c];
        for (int i = 0; i < ncam; i++) for (int d = 1; d <= b && i - d >= 0; d++)
            for (int r = 0; r < DC; r++) for (int q = 0; q < DC; q++) { const double v = U(rng); A[(size_t)(i * DC + r) * m + (i - d) * DC + q] = v; A[(size_t)((i - d) * DC + q) * m + i * DC + r] = v; }
        for (int i = 0; i < ncam; i++) for (int r = 0; r < DC; r++) for (int q = 0; q <= r; q++) { const double v = U(rng); A[(size_t)(i * DC + r) * m + i * DC + q] = v; A[(size_t)(i * DC + q) * m + i * DC + r] = v; }
        for (int i = 0; i < m; i++) { double s = 0; for (int k = 0; k < m; k++) if (k != i) s += std::fabs(A[(size_t)i * m + k]); A[(size_t)i * m + i] = s + 1.0 + std::fabs(U(rng)); }
        for (int i = 0; i < ncam; i++) for (int d = 0; d <= b && i - d >= 0; d++)
            for (int r = 0; r < DC; r++) for (int q = 0; q < DC; q++) band[(((size_t)(c * ncam + i)) * W + d) * BB + r * DC + q] = A[(size_t)(i * DC + r) * m + (i - d) * DC + q];
    }
    for (auto& v : Y) v = U(rng);
    // CPU reference solve
    std::vector<double> Xref(Y);
    for (int c = 0; c < ncomp; c++) {
        const int m = ncam * DC; std::vector<double> L = dense[c];
        for (int k = 0; k < m; k++) {
            double d = L[(size_t)k * m + k]; for (int p = 0; p < k; p++) d -= L[(size_t)k * m + p] * L[(size_t)k * m + p];
            d = std::sqrt(d); L[(size_t)k * m + k] = d;
            for (int i = k + 1; i < m; i++) { double s = L[(size_t)i * m + k]; for (int p = std::max(0, k - (b + 1) * DC); p < k; p++) s -= L[(size_t)i * m + p] * L[(size_t)k * m + p]; L[(size_t)i * m + k] = s / d; }
        }
        for (int r = 0; r < NR; r++) {
            double* x = Xref.data() + (size_t)r * n + (size_t)c * m;
            for (int i = 0; i < m; i++) { double s = x[i]; for (int p = 0; p < i; p++) s -= L[(size_t)i * m + p] * x[p]; x[i] = s / L[(size_t)i * m + i]; }
            for (int i = m - 1; i >= 0; i--) { double s = x[i]; for (int p = i + 1; p < m; p++) s -= L[(size_t)p * m + i] * x[p]; x[i] = s / L[(size_t)i * m + i]; }
        }
    }
    std::vector<int> pairs; for (int ir = 1; ir <= b; ir++) for (int kr = 1; kr <= ir; kr++) pairs.push_back(ir | (kr << 16));
    double *dband0, *dband, *dG, *dY0, *dY; int *dpairs, *dcomp, *dfail;
    CK(hipMalloc(&dband0, band.size() * 8)); CK(hipMalloc(&dband, band.size() * 8)); CK(hipMalloc(&dG, (size_t)N * BB * 8));
    CK(hipMalloc(&dY0, Y.size() * 8)); CK(hipMalloc(&dY, Y.size() * 8)); CK(hipMalloc(&dpairs, pairs.size() * 4 + 4)); CK(hipMalloc(&dcomp, comp_ptr.size() * 4)); CK(hipMalloc(&dfail, 4));
    CK(hipMemcpy(dband0, band.data(), band.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dY0, Y.data(), Y.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcomp, comp_ptr.data(), comp_ptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dfail, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    { static long long* dst_ = nullptr; if (!dst_) { CK(hipMalloc(&dst_, 16 * 8 * 8)); CK(hipMemset(dst_, 0, 16 * 8 * 8)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_chol3_stamps), &dst_, sizeof(dst_))); } }
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    auto check = [&](const char* tag) {
        std::vector<double> X(Y.size()); CK(hipMemcpy(X.data(), dY, X.size() * 8, hipMemcpyDeviceToHost));
        double num = 0, den = 0; for (size_t i = 0; i < X.size(); i++) { num = std::max(num, std::fabs(X[i] - Xref[i])); den = std::max(den, std::fabs(Xref[i])); }
        int fl; CK(hipMemcpy(&fl, dfail, 4, hipMemcpyDeviceToHost));
        printf("  %-28s max rel err %.3e  fail=%d\n", tag, num / den, fl);
        return num / den;
    };
    const size_t lds_new = ((size_t)(b + 1) * W * BB + (size_t)b * BB + (size_t)(b + 1) * NR * DC + NR * DC + 2 * BB) * 8 + ((size_t)b * (b + 1) / 2 + 2) * 4;
    if (lds_new <= 140 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2<DC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
    printf("DC=%d ncomp=%d ncam=%d b=%d  lds %zu\n", DC, ncomp, ncam, b, lds_new);
    auto bench = [&](const char* tag, auto chol, auto back) {
        float tc = 0, tb = 0;
        for (int it = 0; it < reps + 3; it++) {
            CK(hipMemcpyAsync(dband, dband0, band.size() * 8, hipMemcpyDeviceToDevice, st)); CK(hipMemcpyAsync(dY, dY0, Y.size() * 8, hipMemcpyDeviceToDevice, st));
            CK(hipEventRecord(e0, st)); chol(); CK(hipEventRecord(e1, st)); back(); CK(hipEventRecord(e2, st));
            CK(hipStreamSynchronize(st)); CK(hipGetLastError());
            float a, c; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&c, e1, e2));
            if (it >= 3) { tc += a; tb += c; }
        }
        printf("%-24s chol %.1f us   back %.1f us   (per step %.2f / %.2f us)\n", tag, tc / reps * 1e3, tb / reps * 1e3, tc / reps * 1e3 / ncam, tb / reps * 1e3 / ncam);
        check(tag);
    };
    const size_t lds_sub2 = (size_t)(2 * (size_t)b * 2 * DC + 2 * DC) * sizeof(double);
    auto back_any = [&] { if (b * DC > 192) hipLaunchKernelGGL((k_band_back_lds<DC, 2>), dim3(ncomp), dim3(256), lds_sub2, st, dband, dG, dY, dcomp, N, b);
                          else if (b * DC > 128) hipLaunchKernelGGL((k_band_back_v2<DC, 3>), dim3(ncomp, NR), dim3(64), 0, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b);
                          else if (b * DC > 64) hipLaunchKernelGGL((k_band_back_v2<DC, 2>), dim3(ncomp, NR), dim3(64), 0, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b);
                          else hipLaunchKernelGGL((k_band_back_v2<DC, 1>), dim3(ncomp, NR), dim3(64), 0, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b); };
    if constexpr (DC == 6) {
        const size_t lds2p = chol2p_lds_bytes(b, NR);
        if (lds2p <= 160 * 1024) {
            const int nblk = (b * (b + 1) / 2) * 4 - 4 + b * DC;
            printf("v2p: lds %zu B, tasks %d\n", lds2p, nblk);
#define V2P(NT_, NPB_, PRE_, NW_) do { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2p<2, NT_, NPB_, PRE_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2p)); \
            bench("v2p <" #NT_ "," #NPB_ "," #PRE_ "> " #NW_ " waves", [&] { hipLaunchKernelGGL((k_band_chol_v2p<2, NT_, NPB_, PRE_>), dim3(ncomp), dim3(64 * NW_), lds2p, st, dband, dG, dY, dpairs, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b, dfail); }, back_any); } while (0)
#define V2P6(NPB_, PRE_, NW_) do { const size_t l6 = chol2p_lds_bytes(b, NR, true); if (l6 <= 160 * 1024) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2p<2, 1, NPB_, PRE_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l6)); \
            bench("v2p 6x6 <1," #NPB_ "," #PRE_ "> " #NW_ " waves", [&] { hipLaunchKernelGGL((k_band_chol_v2p<2, 1, NPB_, PRE_, true>), dim3(ncomp), dim3(64 * NW_), l6, st, dband, dG, dY, dpairs, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b, dfail); }, back_any); } } while (0)
            if (b <= 14) { V2P6(1, 5, 8); V2P6(1, 5, 12); } else if (b <= 26) { V2P6(2, 8, 12); } else V2P6(2, 9, 12);
            if (b <= 14) { V2P(1, 1, 5, 12); V2P(2, 1, 5, 9); }
            else if (b <= 26) { V2P(3, 1, 8, 16); V2P(2, 1, 8, 16); }
            else V2P(3, 2, 9, 16);
#undef V2P
#undef V2P6
        }
    }
    if (lds_new <= 140 * 1024 && b * BB <= BB + 9 * 64 - 64) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2<DC, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
        bench("v2 early look-ahead (9 waves)", [&] { hipLaunchKernelGGL((k_band_chol_v2<DC, 2, 4>), dim3(ncomp), dim3(9 * 64), lds_new, st, dband, dG, dY, dpairs, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b, dfail, chol_wave_map(9, 9 - 2 - CHOL2_LOADERS, 9 - 2 - CHOL2_LOADERS)); }, back_any);
    }
    if (lds_new <= 140 * 1024) {     // 3x2 tiles: six tasks per block
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2<DC, 2, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
        const int blk6 = (b * (b + 1) / 2) * 6 - 6, w6 = (blk6 + 63) / 64 + 1;
        for (int nw : {4 + w6, 4 + w6 + 1}) if (nw <= 16) {
            char tag[64]; snprintf(tag, 64, "v2 3x2 tiles (%d waves)", nw);
            bench(tag, [&] { hipLaunchKernelGGL((k_band_chol_v2<DC, 2, 0, 2>), dim3(ncomp), dim3(nw * 64), lds_new, st, dband, dG, dY, dpairs, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b, dfail, chol_wave_map(0, 0, 0)); }, back_any);
        }
    }
    if (lds_new <= 140 * 1024)      // (the product's limit for the square window ring)
    for (int nw : {9}) {
        char tag[64]; snprintf(tag, 64, "v2 (%d waves)", nw);
        bench(tag, [&] { hipLaunchKernelGGL((k_band_chol_v2<DC, 2>), dim3(ncomp), dim3(nw * 64), lds_new, st, dband, dG, dY, dpairs, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b, dfail, chol_wave_map(nw, nw - 2 - CHOL2_LOADERS, nw - 2 - CHOL2_LOADERS)); },
              [&] { if (b * DC > 64) hipLaunchKernelGGL((k_band_back_v2<DC, 2>), dim3(ncomp, NR), dim3(64), 0, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b);
                    else hipLaunchKernelGGL((k_band_back_v2<DC, 1>), dim3(ncomp, NR), dim3(64), 0, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b); });
    }

    if constexpr (DC == 6) if (b <= 14) {
        const size_t lds3 = chol3_lds_doubles(b, NR) * 8;
        auto back = [&] { if (b * DC > 64) hipLaunchKernelGGL((k_band_back_v2<DC, 2>), dim3(ncomp, NR), dim3(64), 0, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b);
                          else hipLaunchKernelGGL((k_band_back_v2<DC, 1>), dim3(ncomp, NR), dim3(64), 0, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b); };
        const int tw = chol3_trailing_waves(b);
        printf("v3: lds %zu B, trailing waves %d\n", lds3, tw);
#define V3(TW_, PRE_) bench("v3 <" #TW_ "," #PRE_ ">", [&] { hipLaunchKernelGGL((k_band_chol_v3<2, TW_, PRE_>), dim3(ncomp), dim3(64 * (TW_ + 4)), lds3, st, dband, dG, dY, dcomp, dcomp + 1, dcomp + 1, (const int*)nullptr, N, b, dfail); }, back)
        if (tw <= 2 && (b + 1) * 36 + 12 <= 64 * 7) V3(2, 7);
        else if (tw <= 3 && (b + 1) * 36 + 12 <= 64 * 10) V3(3, 10);
        else if (tw <= 4 && (b + 1) * 36 + 12 <= 64 * 10) V3(4, 10);
#undef V3
        {   // stamps of one step (block 0, step 10) per role: 0 loop head | 1 after phase B | 2 after barrier A | 3 after the role's work | 4 after barrier B | 5 (role 0) before the factorisation
            long long hs[16 * 8]; long long* dptr; CK(hipMemcpyFromSymbol(&dptr, HIP_SYMBOL(g_chol3_stamps), sizeof(dptr)));
            CK(hipMemcpy(hs, dptr, sizeof(hs), hipMemcpyDeviceToHost));
            const long long t0 = hs[0];
            for (int r = 0; r < tw + 4; r++) printf("  role %d: phaseB %5lld  barA %5lld  work %5lld  barB %5lld   (head at %+lld%s)\n", r, hs[r * 8 + 1] - hs[r * 8 + 0], hs[r * 8 + 2] - hs[r * 8 + 1],
                                                   hs[r * 8 + 3] - hs[r * 8 + 2], hs[r * 8 + 4] - hs[r * 8 + 3], hs[r * 8 + 0] - t0, r == 0 ? ", D update" : "");
            printf("  role 0: D update %lld, factor+inverse %lld;  step %lld (s_memtime ticks: 100 MHz -> x cycles/tick)\n", hs[5] - hs[2], hs[3] - hs[5], hs[4] - hs[0]);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int reps = 50;
    run<6>(4, 75, 10, reps);
    run<6>(8, 32, 10, reps);
    run<6>(4, 75, 12, reps);
    run<6>(4, 75, 14, reps);
    run<6>(1, 300, 22, 20);
    run<6>(1, 300, 26, 20);
    run<6>(1, 300, 28, 20);
    run<6>(2, 150, 30, 20);
    return 0;
}
