// spherical_sfm_amd -- forward-mode dual numbers for device code (value + N partials).
// The pose-graph residuals (reference src/rotation_averaging.cpp:15-42, src/uncalibrated_pose_graph.cpp:33-114)
// go through matrix -> quaternion -> angle-axis with several value-dependent branches; the reference differentiates
// them with ceres::Jet, and the edge kernels here do the same on a lane (a few hundred fp64 flops per edge).
#pragma once
#include <cfloat>
#include <cmath>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SSFM_DHD __host__ __device__ __forceinline__
#else
#define SSFM_DHD inline
#endif

namespace ssfm {

template <int N>
struct Dual {
    double a, v[N];
    SSFM_DHD Dual() : a(0.0) { for (int i = 0; i < N; i++) v[i] = 0.0; }
    SSFM_DHD Dual(double x) : a(x) { for (int i = 0; i < N; i++) v[i] = 0.0; }
    SSFM_DHD Dual(double x, int k) : a(x) { for (int i = 0; i < N; i++) v[i] = (i == k) ? 1.0 : 0.0; }
};
#define SSFM_DUAL_BIN(op, expr_a, expr_v)                                                                         \
    template <int N> SSFM_DHD Dual<N> operator op(const Dual<N>& f, const Dual<N>& g) {                           \
        Dual<N> h; h.a = expr_a; for (int i = 0; i < N; i++) h.v[i] = expr_v; return h; }
SSFM_DUAL_BIN(+, f.a + g.a, f.v[i] + g.v[i])
SSFM_DUAL_BIN(-, f.a - g.a, f.v[i] - g.v[i])
SSFM_DUAL_BIN(*, f.a * g.a, f.a * g.v[i] + f.v[i] * g.a)
#undef SSFM_DUAL_BIN
template <int N> SSFM_DHD Dual<N> operator/(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h; const double gi = 1.0 / g.a; h.a = f.a * gi; for (int i = 0; i < N; i++) h.v[i] = (f.v[i] - h.a * g.v[i]) * gi; return h; }
template <int N> SSFM_DHD Dual<N> operator-(const Dual<N>& f) { Dual<N> h; h.a = -f.a; for (int i = 0; i < N; i++) h.v[i] = -f.v[i]; return h; }
template <int N> SSFM_DHD Dual<N> operator*(const Dual<N>& f, double s) { Dual<N> h; h.a = f.a * s; for (int i = 0; i < N; i++) h.v[i] = f.v[i] * s; return h; }
template <int N> SSFM_DHD Dual<N> operator*(double s, const Dual<N>& f) { return f * s; }
template <int N> SSFM_DHD Dual<N> operator+(const Dual<N>& f, double s) { Dual<N> h = f; h.a += s; return h; }
template <int N> SSFM_DHD Dual<N> operator+(double s, const Dual<N>& f) { Dual<N> h = f; h.a += s; return h; }
template <int N> SSFM_DHD Dual<N> operator-(const Dual<N>& f, double s) { Dual<N> h = f; h.a -= s; return h; }
template <int N> SSFM_DHD Dual<N> operator-(double s, const Dual<N>& f) { Dual<N> h = -f; h.a += s; return h; }
template <int N> SSFM_DHD Dual<N> operator/(const Dual<N>& f, double s) { return f * (1.0 / s); }
template <int N> SSFM_DHD Dual<N> operator/(double s, const Dual<N>& g) { return Dual<N>(s) / g; }
template <int N> SSFM_DHD Dual<N> dsqrt(const Dual<N>& f) { Dual<N> h; h.a = sqrt(f.a); const double d = 0.5 / h.a; for (int i = 0; i < N; i++) h.v[i] = f.v[i] * d; return h; }
template <int N> SSFM_DHD Dual<N> dsin(const Dual<N>& f) { Dual<N> h; h.a = sin(f.a); const double d = cos(f.a); for (int i = 0; i < N; i++) h.v[i] = f.v[i] * d; return h; }
template <int N> SSFM_DHD Dual<N> dcos(const Dual<N>& f) { Dual<N> h; h.a = cos(f.a); const double d = -sin(f.a); for (int i = 0; i < N; i++) h.v[i] = f.v[i] * d; return h; }
template <int N> SSFM_DHD Dual<N> datan2(const Dual<N>& y, const Dual<N>& x) {
    Dual<N> h; h.a = atan2(y.a, x.a); const double d = 1.0 / (x.a * x.a + y.a * y.a);
    for (int i = 0; i < N; i++) h.v[i] = (x.a * y.v[i] - y.a * x.v[i]) * d; return h; }
SSFM_DHD double dsqrt(double x) { return sqrt(x); }
SSFM_DHD double dsin(double x) { return sin(x); }
SSFM_DHD double dcos(double x) { return cos(x); }
SSFM_DHD double datan2(double y, double x) { return atan2(y, x); }
SSFM_DHD double dval(double x) { return x; }
template <int N> SSFM_DHD double dval(const Dual<N>& x) { return x.a; }

// Ceres-convention conversions on T = double or Dual<N>; matrices ROW-major here (m[3*i+j]).
template <typename T>
SSFM_DHD void aa_to_matrix_t(const T* aa, T* R) {
    const T t2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (dval(t2) > DBL_EPSILON) {
        const T th = dsqrt(t2), wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
        const T c = dcos(th), s = dsin(th), omc = 1.0 - c;
        R[0] = c + wx * wx * omc;       R[1] = wx * wy * omc - wz * s;  R[2] = wy * s + wx * wz * omc;
        R[3] = wz * s + wx * wy * omc;  R[4] = c + wy * wy * omc;       R[5] = wy * wz * omc - wx * s;
        R[6] = wx * wz * omc - wy * s;  R[7] = wx * s + wy * wz * omc;  R[8] = c + wz * wz * omc;
    } else {
        R[0] = T(1.0); R[1] = -aa[2]; R[2] = aa[1]; R[3] = aa[2]; R[4] = T(1.0); R[5] = -aa[0]; R[6] = -aa[1]; R[7] = aa[0]; R[8] = T(1.0);
    }
}
template <typename T>
SSFM_DHD void matrix_to_aa_t(const T* R, T* aa) {
    T q0, q1, q2, q3;
    const T trace = R[0] + R[4] + R[8];
    if (dval(trace) >= 0.0) {
        T t = dsqrt(trace + 1.0); q0 = 0.5 * t; t = 0.5 / t;
        q1 = (R[7] - R[5]) * t; q2 = (R[2] - R[6]) * t; q3 = (R[3] - R[1]) * t;
    } else {
        int i = 0; if (dval(R[4]) > dval(R[0])) i = 1; if (dval(R[8]) > dval(R[4 * i])) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        T t = dsqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        T q[4]; q[i + 1] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * t; q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t; q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
    }
    const T s2 = q1 * q1 + q2 * q2 + q3 * q3;
    if (dval(s2) > 0.0) {
        const T s = dsqrt(s2);
        const T two_theta = 2.0 * ((dval(q0) < 0.0) ? datan2(-s, -q0) : datan2(s, q0));
        const T k = two_theta / s; aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
    } else { aa[0] = q1 * 2.0; aa[1] = q2 * 2.0; aa[2] = q3 * 2.0; }
}
template <typename T> SSFM_DHD void mat3_mul_t(const T* A, const T* B, T* C) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j]; }
template <typename T> SSFM_DHD void mat3_mul_bt_t(const T* A, const T* B, T* C) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2]; }

}  // namespace ssfm
