// spherical_sfm_amd -- small fixed-size fp64 math shared by host code and HIP kernels.
//
// Everything the reference gets from Eigen / ceres/rotation.h on the hot path, written out for
// 3-vectors and 3x3 / 6x6 blocks so that it lives in registers on a CDNA4 lane:
//   so3exp / so3ln                      reference src/so3.cpp:16-69
//   angle-axis -> R (+ derivative aid)  Ceres AngleAxisRotatePoint / AngleAxisToRotationMatrix, used at
//                                       src/sfm.cpp:47 and src/rotation_averaging.cpp:28-29
//   R -> angle-axis                     Ceres RotationMatrixToAngleAxis, src/rotation_averaging.cpp:32
//   robust losses                       Ceres CauchyLoss / SoftLOneLoss, src/sfm.cpp:196, rotation_averaging.cpp:58
// Matrices are row-major double[9] (m[3*i+j]) unless a function says otherwise.
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SSFM_HD __host__ __device__ __forceinline__
#else
#define SSFM_HD inline
#endif

namespace ssfm {

SSFM_HD void mat3_mul(const double* A, const double* B, double* C) {          // C = A B
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
SSFM_HD void mat3_mul_bt(const double* A, const double* B, double* C) {       // C = A B^T
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
SSFM_HD void mat3_mul_at(const double* A, const double* B, double* C) {       // C = A^T B
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
SSFM_HD void mat3_vec(const double* A, const double* x, double* y) {
    for (int i = 0; i < 3; i++) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
SSFM_HD void mat3_tvec(const double* A, const double* x, double* y) {         // y = A^T x
    for (int i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
SSFM_HD void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
SSFM_HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
SSFM_HD double norm3(const double* a) { return sqrt(dot3(a, a)); }

// ---- reference SO(3) helpers (src/so3.cpp) -------------------------------------------------------
SSFM_HD void so3exp(const double* r, double* R) {                            // src/so3.cpp:16-23
    const double theta = norm3(r);
    if (theta < 1e-10) { R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1; return; }
    const double kx = r[0] / theta, ky = r[1] / theta, kz = r[2] / theta;
    const double s = sin(theta), omc = 1.0 - cos(theta);
    // I + s K + omc K^2 with K = skew(k)
    R[0] = 1.0 + omc * (-(ky * ky) - kz * kz); R[1] = -s * kz + omc * (kx * ky);      R[2] = s * ky + omc * (kx * kz);
    R[3] = s * kz + omc * (kx * ky);           R[4] = 1.0 + omc * (-(kx * kx) - kz * kz); R[5] = -s * kx + omc * (ky * kz);
    R[6] = -s * ky + omc * (kx * kz);          R[7] = s * kx + omc * (ky * kz);       R[8] = 1.0 + omc * (-(kx * kx) - ky * ky);
}

SSFM_HD void so3ln(const double* R, double* out) {                           // src/so3.cpp:25-69
    const double cos_angle = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    out[0] = (R[7] - R[5]) / 2; out[1] = (R[2] - R[6]) / 2; out[2] = (R[3] - R[1]) / 2;
    const double sin_abs = norm3(out);
    const double kSqrtHalf = 0.70710678118654752440;
    if (cos_angle > kSqrtHalf) {
        if (sin_abs > 0) { const double k = asin(sin_abs) / sin_abs; out[0] *= k; out[1] *= k; out[2] *= k; }
    } else if (cos_angle > -kSqrtHalf) {
        const double k = acos(cos_angle) / sin_abs; out[0] *= k; out[1] *= k; out[2] *= k;
    } else {
        const double angle = 3.14159265358979323846 - asin(sin_abs);
        const double d0 = R[0] - cos_angle, d1 = R[4] - cos_angle, d2 = R[8] - cos_angle;
        double r2[3];
        if (fabs(d0) > fabs(d1) && fabs(d0) > fabs(d2)) { r2[0] = d0; r2[1] = (R[3] + R[1]) / 2; r2[2] = (R[2] + R[6]) / 2; }
        else if (fabs(d1) > fabs(d2)) { r2[0] = (R[3] + R[1]) / 2; r2[1] = d1; r2[2] = (R[7] + R[5]) / 2; }
        else { r2[0] = (R[2] + R[6]) / 2; r2[1] = (R[7] + R[5]) / 2; r2[2] = d2; }
        if (dot3(r2, out) < 0) { r2[0] = -r2[0]; r2[1] = -r2[1]; r2[2] = -r2[2]; }
        const double n = norm3(r2);
        out[0] = angle * r2[0] / n; out[1] = angle * r2[1] / n; out[2] = angle * r2[2] / n;
    }
}

// ---- Ceres-convention angle-axis (threshold theta^2 > DBL_EPSILON, Taylor branch otherwise) -------
// R = rotation used for VALUES (Rodrigues, or I + [r]x near zero, exactly what the reference's
// templated code evaluates).  Row-major.
SSFM_HD void angle_axis_to_matrix(const double* aa, double* R) {
    const double t2 = dot3(aa, aa);
    if (t2 > DBL_EPSILON) {
        const double th = sqrt(t2), wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
        const double c = cos(th), s = sin(th), omc = 1.0 - c;
        R[0] = c + wx * wx * omc;      R[1] = wx * wy * omc - wz * s; R[2] = wy * s + wx * wz * omc;
        R[3] = wz * s + wx * wy * omc; R[4] = c + wy * wy * omc;      R[5] = -wx * s + wy * wz * omc;
        R[6] = -wy * s + wx * wz * omc; R[7] = wx * s + wy * wz * omc; R[8] = c + wz * wz * omc;
    } else {
        R[0] = 1; R[1] = -aa[2]; R[2] = aa[1]; R[3] = aa[2]; R[4] = 1; R[5] = -aa[0]; R[6] = -aa[1]; R[7] = aa[0]; R[8] = 1;
    }
}

// Per-camera constants for the analytic derivative of q = R(r) X with respect to the additive
// angle-axis parameters (the reference differentiates AngleAxisRotatePoint with Jets, src/sfm.cpp:47,219):
//   Rodrigues branch:  dq/dr = -R [X]x M,  M = (r r^T + (R^T - I)[r]x) / theta^2   (Gallego & Yezzi 2015)
//   Taylor branch   :  q = X + r x X  =>  dq/dr = -[X]x          (Rd = I, M = I)
// Output: R (values), Rd (rotation multiplying the derivative), M.
SSFM_HD void angle_axis_derivative_aid(const double* aa, double* R, double* Rd, double* M) {
    angle_axis_to_matrix(aa, R);
    const double t2 = dot3(aa, aa);
    if (t2 > DBL_EPSILON) {
        for (int i = 0; i < 9; i++) Rd[i] = R[i];
        // A = R^T - I ; S = [r]x ; M = (r r^T + A S)/t2
        double A[9] = {R[0] - 1.0, R[3], R[6], R[1], R[4] - 1.0, R[7], R[2], R[5], R[8] - 1.0};
        const double S[9] = {0, -aa[2], aa[1], aa[2], 0, -aa[0], -aa[1], aa[0], 0};
        double AS[9]; mat3_mul(A, S, AS);
        const double inv = 1.0 / t2;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[3 * i + j] = (aa[i] * aa[j] + AS[3 * i + j]) * inv;
    } else {
        for (int i = 0; i < 9; i++) { Rd[i] = (i % 4 == 0) ? 1.0 : 0.0; M[i] = Rd[i]; }
    }
}

// Ceres RotationMatrixToAngleAxis (matrix -> quaternion (Shoemake) -> angle-axis), row-major input.
SSFM_HD void matrix_to_angle_axis(const double* R, double* aa) {
    double q0, q1, q2, q3;
    const double trace = R[0] + R[4] + R[8];
    if (trace >= 0.0) {
        double t = sqrt(trace + 1.0); q0 = 0.5 * t; t = 0.5 / t;
        q1 = (R[7] - R[5]) * t; q2 = (R[2] - R[6]) * t; q3 = (R[3] - R[1]) * t;
    } else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        double q[4]; q[i + 1] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * t; q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t; q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
    }
    const double s2 = q1 * q1 + q2 * q2 + q3 * q3;
    if (s2 > 0.0) {
        const double s = sqrt(s2);
        const double two_theta = 2.0 * ((q0 < 0.0) ? atan2(-s, -q0) : atan2(s, q0));
        const double k = two_theta / s; aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
    } else { aa[0] = q1 * 2.0; aa[1] = q2 * 2.0; aa[2] = q3 * 2.0; }
}

// ---- robust losses: returns rho(s) and rho'(s) (Ceres loss_function.cc; rho'' <= 0 for both, so the
// Corrector is the plain sqrt(rho') scaling).  type: 0 trivial, 1 Cauchy(a), 2 SoftLOne(a).
SSFM_HD void robust_loss(int type, double a, double s, double& rho0, double& rho1) {
    if (type == 1) {
        const double b = a * a, sum = 1.0 + s / b; rho0 = b * log(sum); rho1 = fmax(DBL_MIN, 1.0 / sum);
    } else if (type == 2) {
        const double b = a * a, sum = 1.0 + s / b, tmp = sqrt(sum); rho0 = 2.0 * b * (tmp - 1.0); rho1 = fmax(DBL_MIN, 1.0 / tmp);
    } else { rho0 = s; rho1 = 1.0; }
}

// symmetric 3x3 (xx xy xz yy yz zz) inverse by cofactors
SSFM_HD void sym3_inverse(const double* V, double* Vi) {
    const double c00 = V[3] * V[5] - V[4] * V[4], c01 = V[2] * V[4] - V[1] * V[5], c02 = V[1] * V[4] - V[2] * V[3];
    const double id = 1.0 / (V[0] * c00 + V[1] * c01 + V[2] * c02);
    Vi[0] = c00 * id; Vi[1] = c01 * id; Vi[2] = c02 * id;
    Vi[3] = (V[0] * V[5] - V[2] * V[2]) * id; Vi[4] = (V[1] * V[2] - V[0] * V[4]) * id; Vi[5] = (V[0] * V[3] - V[1] * V[1]) * id;
}

}  // namespace ssfm
