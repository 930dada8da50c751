// ORACLE (test infrastructure only) -- rotation helpers.
//
// Two families, kept apart exactly as the reference keeps them apart:
//  (1) the Ceres 2.2.0 `ceres/rotation.h` templates the residual functors call
//      (AngleAxisRotatePoint   <- src/sfm.cpp:47,
//       AngleAxisToRotationMatrix / RotationMatrixToAngleAxis
//                              <- src/rotation_averaging.cpp:28-32,
//                                 src/uncalibrated_pose_graph.cpp:58-69,97-105,
//                                 src/spherical_estimator.cpp:35-40).
//      Ceres is NOT in /root/reference (docker/Dockerfile:50-56 pins 2.2.0); these are a
//      restatement of its published algorithm: Rodrigues for theta^2 > DBL_EPSILON, first-order
//      Taylor otherwise; matrix -> quaternion (Shoemake) -> angle-axis with the cos<0 flip.
//      Matrices are COLUMN-MAJOR (R[i + 3*j]) because the reference hands Eigen's .data().
//  (2) the reference's own so3exp / so3ln / skew3 (src/so3.cpp:6-69), double only.
#pragma once
#include <cfloat>
#include <cmath>
#include "jet.hpp"

namespace oracle {

template <typename T>
inline void AngleAxisRotatePoint(const T aa[3], const T pt[3], T out[3]) {
    const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > DBL_EPSILON) {
        const T theta = jsqrt(theta2);
        const T c = jcos(theta);
        const T s = jsin(theta);
        const T inv = T(1.0) / theta;
        const T w[3] = {aa[0] * inv, aa[1] * inv, aa[2] * inv};
        const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                          w[0] * pt[1] - w[1] * pt[0]};
        const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - c);
        out[0] = pt[0] * c + wxp[0] * s + w[0] * tmp;
        out[1] = pt[1] * c + wxp[1] * s + w[1] * tmp;
        out[2] = pt[2] * c + wxp[2] * s + w[2] * tmp;
    } else {
        const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2],
                          aa[0] * pt[1] - aa[1] * pt[0]};
        out[0] = pt[0] + wxp[0];
        out[1] = pt[1] + wxp[1];
        out[2] = pt[2] + wxp[2];
    }
}

// column-major 3x3 output
template <typename T>
inline void AngleAxisToRotationMatrix(const T aa[3], T R[9]) {
#define RM(i, j) R[(i) + 3 * (j)]
    const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > DBL_EPSILON) {
        const T theta = jsqrt(theta2);
        const T wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
        const T c = jcos(theta), s = jsin(theta);
        const T omc = T(1.0) - c;
        RM(0, 0) = c + wx * wx * omc;
        RM(1, 0) = wz * s + wx * wy * omc;
        RM(2, 0) = -wy * s + wx * wz * omc;
        RM(0, 1) = wx * wy * omc - wz * s;
        RM(1, 1) = c + wy * wy * omc;
        RM(2, 1) = wx * s + wy * wz * omc;
        RM(0, 2) = wy * s + wx * wz * omc;
        RM(1, 2) = -wx * s + wy * wz * omc;
        RM(2, 2) = c + wz * wz * omc;
    } else {
        RM(0, 0) = T(1.0); RM(1, 0) = aa[2];   RM(2, 0) = -aa[1];
        RM(0, 1) = -aa[2]; RM(1, 1) = T(1.0);  RM(2, 1) = aa[0];
        RM(0, 2) = aa[1];  RM(1, 2) = -aa[0];  RM(2, 2) = T(1.0);
    }
#undef RM
}

template <typename T>
inline void RotationMatrixToQuaternion(const T R[9], T q[4]) {
#define RM(i, j) R[(i) + 3 * (j)]
    const T trace = RM(0, 0) + RM(1, 1) + RM(2, 2);
    if (trace >= 0.0) {
        T t = jsqrt(trace + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (RM(2, 1) - RM(1, 2)) * t;
        q[2] = (RM(0, 2) - RM(2, 0)) * t;
        q[3] = (RM(1, 0) - RM(0, 1)) * t;
    } else {
        int i = 0;
        if (RM(1, 1) > RM(0, 0)) i = 1;
        if (RM(2, 2) > RM(i, i)) i = 2;
        const int j = (i + 1) % 3;
        const int k = (j + 1) % 3;
        T t = jsqrt(RM(i, i) - RM(j, j) - RM(k, k) + 1.0);
        q[i + 1] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (RM(k, j) - RM(j, k)) * t;
        q[j + 1] = (RM(j, i) + RM(i, j)) * t;
        q[k + 1] = (RM(k, i) + RM(i, k)) * t;
    }
#undef RM
}

template <typename T>
inline void QuaternionToAngleAxis(const T q[4], T aa[3]) {
    const T s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2 > 0.0) {
        const T s = jsqrt(s2);
        const T& c = q[0];
        const T two_theta = 2.0 * ((c < 0.0) ? jatan2(-s, -c) : jatan2(s, c));
        const T k = two_theta / s;
        aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
    } else {
        aa[0] = q[1] * 2.0; aa[1] = q[2] * 2.0; aa[2] = q[3] * 2.0;
    }
}

template <typename T>
inline void RotationMatrixToAngleAxis(const T R[9], T aa[3]) {
    T q[4];
    RotationMatrixToQuaternion(R, q);
    QuaternionToAngleAxis(q, aa);
}

// C = A * B, all column-major 3x3
template <typename T>
inline void mat3_mul(const T A[9], const T B[9], T C[9]) {
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            C[i + 3 * j] = A[i] * B[3 * j] + A[i + 3] * B[1 + 3 * j] + A[i + 6] * B[2 + 3 * j];
}
// C = A * B^T
template <typename T>
inline void mat3_mul_bt(const T A[9], const T B[9], T C[9]) {
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            C[i + 3 * j] = A[i] * B[j] + A[i + 3] * B[j + 3] + A[i + 6] * B[j + 6];
}

// ---- reference's own SO(3) helpers (src/so3.cpp), column-major doubles ---------------------
inline void so3exp(const double r[3], double R[9]) {   // src/so3.cpp:16-23
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int i = 0; i < 9; i++) R[i] = 0.0;
    R[0] = R[4] = R[8] = 1.0;
    if (theta < 1e-10) return;
    const double k[3] = {r[0] / theta, r[1] / theta, r[2] / theta};
    double K[9] = {0, k[2], -k[1], -k[2], 0, k[0], k[1], -k[0], 0};   // column-major skew
    double KK[9];
    mat3_mul(K, K, KK);
    double s, cs; sincos_pair(theta, &s, &cs);
    const double omc = 1.0 - cs;
    for (int i = 0; i < 9; i++) R[i] += s * K[i] + omc * KK[i];
}

inline void so3ln(const double R[9], double out[3]) {   // src/so3.cpp:25-69
#define RM(i, j) R[(i) + 3 * (j)]
    const double cos_angle = (RM(0, 0) + RM(1, 1) + RM(2, 2) - 1.0) * 0.5;
    out[0] = (RM(2, 1) - RM(1, 2)) / 2;
    out[1] = (RM(0, 2) - RM(2, 0)) / 2;
    out[2] = (RM(1, 0) - RM(0, 1)) / 2;
    const double sin_abs = std::sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]);
    if (cos_angle > M_SQRT1_2) {
        if (sin_abs > 0) {
            const double k = std::asin(sin_abs) / sin_abs;
            out[0] *= k; out[1] *= k; out[2] *= k;
        }
    } else if (cos_angle > -M_SQRT1_2) {
        const double k = std::acos(cos_angle) / sin_abs;
        out[0] *= k; out[1] *= k; out[2] *= k;
    } else {
        const double angle = M_PI - std::asin(sin_abs);
        const double d0 = RM(0, 0) - cos_angle, d1 = RM(1, 1) - cos_angle, d2 = RM(2, 2) - cos_angle;
        double r2[3];
        if (std::fabs(d0) > std::fabs(d1) && std::fabs(d0) > std::fabs(d2)) {
            r2[0] = d0; r2[1] = (RM(1, 0) + RM(0, 1)) / 2; r2[2] = (RM(0, 2) + RM(2, 0)) / 2;
        } else if (std::fabs(d1) > std::fabs(d2)) {
            r2[0] = (RM(1, 0) + RM(0, 1)) / 2; r2[1] = d1; r2[2] = (RM(2, 1) + RM(1, 2)) / 2;
        } else {
            r2[0] = (RM(0, 2) + RM(2, 0)) / 2; r2[1] = (RM(2, 1) + RM(1, 2)) / 2; r2[2] = d2;
        }
        if (r2[0] * out[0] + r2[1] * out[1] + r2[2] * out[2] < 0) { r2[0] = -r2[0]; r2[1] = -r2[1]; r2[2] = -r2[2]; }
        const double n = std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
        out[0] = angle * r2[0] / n; out[1] = angle * r2[1] / n; out[2] = angle * r2[2] / n;
    }
#undef RM
}

}  // namespace oracle
