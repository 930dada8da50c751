// Lab: the 16x16 factor-and-invert of the separator chain, lane-per-row (wave_chol_inverse16) against the matrix-core elimination (wave_ldl_inverse16_mfma):
// s_memtime cycles of one wave and the error of G = L^-1 against a long-double Cholesky.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../spherical_sfm_amd/csrc ldl16_lab.hip -o ldl16_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "ba_flatten.h"
#include "band_sub.h"
using namespace ssfm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// A: [nb][16][16] row-major SPD; G out likewise (lower); one wave per block, `reps` repetitions for timing
__global__ void __launch_bounds__(64) k_old(const double* A, double* Gout, long long* cyc, int reps) {
    const int lane = threadIdx.x, b = blockIdx.x;
    long long t = 0; double g[16];
    for (int r = 0; r < reps; r++) {
        double row[16];
        for (int c = 0; c < 16; c++) row[c] = lane < 16 ? A[(size_t)b * 256 + (lane > c ? lane : c) * 16 + (lane > c ? c : lane)] : (lane == c ? 1.0 : 0.0);
        __builtin_amdgcn_s_waitcnt(0);
        const long long t0 = __builtin_amdgcn_s_memtime();
        wave_chol_inverse16(row, g);
        asm volatile("" :: "v"(g[0]), "v"(g[15]));
        t += __builtin_amdgcn_s_memtime() - t0;
    }
    if (lane < 16) for (int r = 0; r < 16; r++) Gout[(size_t)b * 256 + r * 16 + lane] = (r >= lane) ? g[r] : 0.0;
    if (lane == 0) cyc[b] = t / reps;
}
__global__ void __launch_bounds__(64) k_new(const double* A, double* Gout, long long* cyc, int reps, int* okout) {
    const int lane = threadIdx.x, b = blockIdx.x, li = lane & 15, lk = lane >> 4;
    long long t = 0; v4d_t G; bool ok = true;
    for (int r = 0; r < reps; r++) {
        v4d_t S;
        for (int q = 0; q < 4; q++) S[q] = A[(size_t)b * 256 + (lk + 4 * q) * 16 + li];
        __builtin_amdgcn_s_waitcnt(0);
        const long long t0 = __builtin_amdgcn_s_memtime();
        ok = wave_ldl_inverse16_mfma(S, G);
        asm volatile("" :: "v"(G[0]), "v"(G[3]));
        t += __builtin_amdgcn_s_memtime() - t0;
    }
    for (int q = 0; q < 4; q++) Gout[(size_t)b * 256 + (lk + 4 * q) * 16 + li] = (lk + 4 * q >= li) ? G[q] : 0.0;
    if (lane == 0) { cyc[b] = t / reps; okout[b] = ok ? 1 : 0; }
}
int main() {
    const int nb = 64; std::mt19937_64 rng(3); std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> A((size_t)nb * 256), Gref((size_t)nb * 256, 0.0);
    for (int b = 0; b < nb; b++) {
        double M[16][20];
        for (int i = 0; i < 16; i++) for (int k = 0; k < 20; k++) M[i][k] = U(rng) * std::pow(10.0, (i % 4) - 1.5 * (b % 3));      // rows of different scales
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 20; k++) s += M[i][k] * M[j][k]; A[(size_t)b * 256 + i * 16 + j] = s; }
        for (int i = 0; i < 16; i++) for (int j = 0; j < i; j++) A[(size_t)b * 256 + j * 16 + i] = A[(size_t)b * 256 + i * 16 + j];
        long double L[16][16] = {}, Gi[16][16] = {};
        for (int j = 0; j < 16; j++) {
            long double d = A[(size_t)b * 256 + j * 16 + j]; for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
            L[j][j] = sqrtl(d);
            for (int i = j + 1; i < 16; i++) { long double v = A[(size_t)b * 256 + i * 16 + j]; for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k]; L[i][j] = v / L[j][j]; }
        }
        for (int c = 0; c < 16; c++) for (int r = c; r < 16; r++) { long double acc = (r == c) ? 1.0L : 0.0L; for (int k = c; k < r; k++) acc -= L[r][k] * Gi[k][c]; Gi[r][c] = acc / L[r][r]; }
        for (int r = 0; r < 16; r++) for (int c = 0; c <= r; c++) Gref[(size_t)b * 256 + r * 16 + c] = (double)Gi[r][c];
    }
    double *dA, *dG; long long* dc; int* dok;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dG, A.size() * 8)); CK(hipMalloc(&dc, nb * 8)); CK(hipMalloc(&dok, nb * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> G(A.size()); std::vector<long long> cyc(nb); std::vector<int> okv(nb);
    auto report = [&](const char* name) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(G.data(), dG, G.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc.data(), dc, nb * 8, hipMemcpyDeviceToHost));
        double worst = 0; long long cmin = cyc[0], cmax = 0;
        for (int b = 0; b < nb; b++) {
            double nrm = 0, err = 0;
            for (int e = 0; e < 256; e++) { nrm = std::max(nrm, std::fabs(Gref[(size_t)b * 256 + e])); err = std::max(err, std::fabs(G[(size_t)b * 256 + e] - Gref[(size_t)b * 256 + e])); }
            worst = std::max(worst, err / nrm); cmin = std::min(cmin, cyc[b]); cmax = std::max(cmax, cyc[b]);
        }
        printf("%-28s cycles per call %lld..%lld   max |G - Gref| / max |Gref| = %.2e\n", name, cmin, cmax, worst);
    };
    for (int grid : {1, nb}) {
        printf("-- %d wave(s) in flight\n", grid);
        hipLaunchKernelGGL(k_old, dim3(grid), dim3(64), 0, 0, dA, dG, dc, 20); if (grid == nb) report("lane per row (round 2)"); else { CK(hipDeviceSynchronize()); CK(hipMemcpy(cyc.data(), dc, 8, hipMemcpyDeviceToHost)); printf("lane per row: %lld cycles\n", cyc[0]); }
        hipLaunchKernelGGL(k_new, dim3(grid), dim3(64), 0, 0, dA, dG, dc, 20, dok); if (grid == nb) { report("matrix-core elimination"); CK(hipMemcpy(okv.data(), dok, nb * 4, hipMemcpyDeviceToHost)); int n_ok = 0; for (int v : okv) n_ok += v; printf("ok flags %d / %d\n", n_ok, nb); }
        else { CK(hipDeviceSynchronize()); CK(hipMemcpy(cyc.data(), dc, 8, hipMemcpyDeviceToHost)); printf("matrix cores: %lld cycles\n", cyc[0]); }
    }
    return 0;
}
