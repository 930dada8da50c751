// accuracy of v_rsq_f64 / v_rcp_f64 and of one / two Newton steps on top (lab): hipcc --offload-arch=gfx950 -O3 rsq_lab.hip -o rsq_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const double d = x[i];
    double y0 = __builtin_amdgcn_rsq(d); const double h = 0.5 * d;
    double y1 = y0 * (1.5 - h * y0 * y0); double y2 = y1 * (1.5 - h * y1 * y1);
    double r0 = __builtin_amdgcn_rcp(d); double r1 = r0 * (2.0 - d * r0); double r2 = r1 * (2.0 - d * r1);
    { const double e = fma(-d * y0, y0, 1.0); y1 = fma(y0 * e, fma(e, 0.375, 0.5), y0); }      // third-order step in slot 1
    out[6 * i + 0] = y0; out[6 * i + 1] = y1; out[6 * i + 2] = y2; out[6 * i + 3] = r0; out[6 * i + 4] = r1; out[6 * i + 5] = r2;
}
int main() {
    const int n = 1 << 20; std::vector<double> x(n), o(6 * (size_t)n);
    for (int i = 0; i < n; i++) x[i] = std::exp(-20.0 + 40.0 * (i + 0.37) / n) * (1.0 + 1e-3 * (i % 977));
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
        for (int k2 = 0; k2 < 3; k2++) { e[k2] = fmax(e[k2], (double)fabsl(((long double)o[6 * (size_t)i + k2] - rs) / rs)); e[3 + k2] = fmax(e[3 + k2], (double)fabsl(((long double)o[6 * (size_t)i + 3 + k2] - rc) / rc)); }
    }
    printf("max relative error: rsq %.3e, +1 Newton %.3e, +2 Newton %.3e | rcp %.3e, +1 Newton %.3e, +2 Newton %.3e (eps 2.2e-16)\n", e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}
