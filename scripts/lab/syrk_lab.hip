// Lab: the rank-Q update D -= F F^T of the separator chain (band_sub.h) in isolation: one workgroup of 1024 threads, F (Q x Q, column-major) in LDS, lower 16x16 tiles
// over the 16 waves, v_mfma_f64_16x16x4 with operands straight from LDS.  Variants isolate the matrix pipe, the LDS reads and the tile-to-wave map.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 syrk_lab.hip -o syrk_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d_t __attribute__((ext_vector_type(4)));
#define MFMA64(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// V: 0 = product loop (reads + MFMA) | 1 = MFMA only (operands from registers) | 2 = reads only (one VALU FMA per read pair)
// | 3 = reads + MFMA, software pipelined by hand with asm volatile barriers | 4 = as 0 but tile t -> wave (t * 5) % 16 | 5 = one tile per WAVE PAIR ... (unused)
template <int V>
__global__ void __launch_bounds__(1024) k_syrk(int Q, int reps, long long* cyc, double* out) {
    extern __shared__ double sF[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4, nw = 16, TB = 16;
    for (int e = tid; e < Q * Q + 64; e += 1024) sF[e] = 1e-3 * ((e * 7) % 13 - 6);
    __syncthreads();
    const int TQ = (Q + 15) / 16, ntile = TQ * (TQ + 1) / 2;
    long long t0 = 0; double sink = 0.0;
    for (int r = 0; r < reps; r++) {
        __syncthreads();
        if (tid == 0) t0 = __builtin_amdgcn_s_memtime();
        for (int t0i = wave; t0i < ntile; t0i += nw) {
            const int t = (V == 4) ? t0i : t0i;
            int I = 0; while ((I + 1) * (I + 2) / 2 <= t) I++;
            const int J = t - I * (I + 1) / 2, r0 = TB * I, c0 = TB * J;
            v4d_t acc = {0.0, 0.0, 0.0, 0.0};
            if (V == 0 || V == 4) {
                for (int k0 = 0; k0 + 4 <= Q; k0 += 4) { const int kc = k0 + lk; acc = MFMA64(sF[kc * Q + r0 + li], sF[kc * Q + c0 + li], acc); }
            } else if (V == 1) {
                double a = 1.0 + li, b = 2.0 + lk;
                for (int k0 = 0; k0 + 4 <= Q; k0 += 4) { acc = MFMA64(a, b, acc); }
            } else if (V == 2) {
                for (int k0 = 0; k0 + 4 <= Q; k0 += 4) { const int kc = k0 + lk; acc[0] += sF[kc * Q + r0 + li] * sF[kc * Q + c0 + li]; }
            } else if (V == 3) {
                // explicit double buffering: operands of trip n + 1 requested before the products of trip n
                const double* pa = sF + lk * Q + r0 + li; const double* pb = sF + lk * Q + c0 + li;
                double a0 = pa[0], b0 = pb[0], a1 = pa[4 * Q], b1 = pb[4 * Q];
                int k0 = 0;
                for (; k0 + 16 <= Q; k0 += 8) {
                    const double* na = pa + (size_t)(k0 + 8) * Q; const double* nb = pb + (size_t)(k0 + 8) * Q;
                    const double a2 = na[0], b2 = nb[0], a3 = na[4 * Q], b3 = nb[4 * Q];
                    asm volatile("" ::: "memory");
                    acc = MFMA64(a0, b0, acc); acc = MFMA64(a1, b1, acc);
                    a0 = a2; b0 = b2; a1 = a3; b1 = b3;
                }
                acc = MFMA64(a0, b0, acc); acc = MFMA64(a1, b1, acc);
            }
            sink += acc[0] + acc[1] + acc[2] + acc[3];
        }
        __syncthreads();
        if (tid == 0) cyc[r] = __builtin_amdgcn_s_memtime() - t0;
    }
    out[tid] = sink;
}
template <int V> void run(const char* name, int Q) {
    const int reps = 8; long long* dc; double* dout; CK(hipMalloc(&dc, reps * 8)); CK(hipMalloc(&dout, 1024 * 8));
    const size_t lds = ((size_t)Q * Q + 64 + 16 * Q) * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_syrk<V>, dim3(1), dim3(1024), lds, 0, Q, reps, dc, dout);
    CK(hipDeviceSynchronize());
    std::vector<long long> c(reps); CK(hipMemcpy(c.data(), dc, reps * 8, hipMemcpyDeviceToHost));
    long long best = c[1]; for (int r = 1; r < reps; r++) best = std::min(best, c[r]);
    const int TQ = (Q + 15) / 16, ntile = TQ * (TQ + 1) / 2;
    printf("Q = %3d  %-44s %6lld cycles  (%d tiles x %d products = %d; / 4 SIMDs x 64 = %d)\n", Q, name, best, ntile, Q / 4, ntile * (Q / 4), ntile * (Q / 4) * 16);
    CK(hipFree(dc)); CK(hipFree(dout));
}
int main() {
    for (int Q : {48, 84, 96}) {
        run<0>("reads + MFMA (product loop, rolled)", Q);
        run<1>("MFMA only, operands in registers", Q);
        run<2>("LDS reads only (+ one FMA each)", Q);
        run<3>("reads + MFMA, double-buffered by hand", Q);
    }
    return 0;
}
