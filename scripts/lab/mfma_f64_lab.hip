// Lab: operand / result layout and issue rate of the two FP64 matrix instructions of gfx950.
//   v_mfma_f64_16x16x4_f64      one 16x16x4 product, A / B one f64 per lane, D four
//   v_mfma_f64_4x4x4_4b_f64     four independent 4x4x4 products, A / B / D one f64 per lane
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_lab scripts/lab/mfma_f64_lab.hip && /tmp/mfma_f64_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

// layout probe of the 4x4x4 form: A = 1 on lane a only, B = 1 on lane b only; which lanes see a non-zero D?
__global__ void k_probe4(unsigned long long* out) {
    const int lane = threadIdx.x;
    for (int a = 0; a < 64; a++)
        for (int b = 0; b < 64; b++) {
            const double A = (lane == a) ? 1.0 : 0.0, B = (lane == b) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[a * 64 + b] = m;
        }
}
template <int WHICH>
__global__ void k_rate(double* sink, long long* cycles, int n) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}; double s[4] = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (WHICH == 0) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
            else s[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[u], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double r = 0; for (int u = 0; u < 4; u++) r += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3] + s[u];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[WHICH] = t1 - t0;
}
int main() {
    unsigned long long* d_out; hipMalloc(&d_out, 4096 * 8);
    k_probe4<<<1, 64>>>(d_out);
    std::vector<unsigned long long> h(4096); hipMemcpy(h.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost);
    // for every (a, b) with a non-zero result: which lane(s) of D; print compactly: for a in one block, the b lanes that pair with it and the D lanes
    for (int a = 0; a < 64; a += 1) {
        int shown = 0;
        for (int b = 0; b < 64 && shown < 4; b++) if (h[a * 64 + b]) { std::printf("A lane %2d x B lane %2d -> D lanes mask %016llx\n", a, b, h[a * 64 + b]); shown++; }
        if (a == 7) a = 15; if (a == 17) a = 62;
    }
    double* sink; long long* cyc; hipMalloc(&sink, 1024 * 256 * 8); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
    const int n = 4096;
    for (int waves : {1, 2, 4}) {
        k_rate<0><<<1, 64 * waves>>>(sink, cyc, n); k_rate<1><<<1, 64 * waves>>>(sink, cyc, n); hipDeviceSynchronize();
        long long hc[2]; hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
        std::printf("%d wave(s) per workgroup: 16x16x4 %.1f cycles per instruction, 4x4x4_4b %.1f (shader clock counter, %d x 4 independent chains)\n", waves, (double)hc[0] / (4.0 * n), (double)hc[1] / (4.0 * n), n);
    }
    return 0;
}
