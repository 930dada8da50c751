// ORACLE (test infrastructure only) -- forward-mode dual numbers.
//
// The reference evaluates every residual through ceres::AutoDiffCostFunction, i.e. with
// ceres::Jet<double,N> (Ceres 2.2.0, not vendored in /root/reference; call sites:
// src/sfm.cpp:219, src/rotation_averaging.cpp:66, src/uncalibrated_pose_graph.cpp:134,170,
// src/spherical_estimator.cpp:135).  This header restates that arithmetic: a value plus N
// partial derivatives, propagated exactly by the chain rule.  Comparisons look at the value
// only, as Ceres' Jet comparisons do, so the branch structure of the templated residuals
// (theta^2 > eps, trace >= 0, ...) is reproduced.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/.
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0.0) { for (int i = 0; i < N; i++) v[i] = 0.0; }
    Jet(double x) : a(x) { for (int i = 0; i < N; i++) v[i] = 0.0; }  // NOLINT implicit
    Jet(double x, int k) : a(x) { for (int i = 0; i < N; i++) v[i] = 0.0; v[k] = 1.0; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; i++) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; i++) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
    Jet<N> h; h.a = -f.a; for (int i = 0; i < N; i++) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; i++) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
    // (f/g)' = (f' - (f/g) g') / g
    Jet<N> h; const double gi = 1.0 / g.a; h.a = f.a * gi;
    for (int i = 0; i < N; i++) h.v[i] = (f.v[i] - h.a * g.v[i]) * gi; return h; }

template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
    Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; i++) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline Jet<N>& operator-=(Jet<N>& f, const Jet<N>& g) { f = f - g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, const Jet<N>& g) { f = f * g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, double s) { f = f * s; return f; }

template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator>=(const Jet<N>& f, const Jet<N>& g) { return f.a >= g.a; }
template <int N> inline bool operator>(const Jet<N>& f, double s) { return f.a > s; }
template <int N> inline bool operator<(const Jet<N>& f, double s) { return f.a < s; }
template <int N> inline bool operator>=(const Jet<N>& f, double s) { return f.a >= s; }

template <int N> inline Jet<N> jsqrt(const Jet<N>& f) {
    Jet<N> h; h.a = std::sqrt(f.a); const double d = 0.5 / h.a;
    for (int i = 0; i < N; i++) h.v[i] = f.v[i] * d; return h; }
// sin and cos of one angle.  Ceres' Jet cos() / sin() evaluate both functions of the same argument, and so does every caller here; GCC at
// -O2 and above turns such a pair into ONE call of glibc's sincos(), whose sine is not always bit-identical to sin()'s (it differs in the last
// bit for e.g. 0.83775804095727813).  The pair is requested explicitly so that the bits do not depend on what the optimiser happened to merge
// (the device replay of SfM::Retriangulate takes the cameras' sin / cos from the same call, retriangulate.hip).
inline void sincos_pair(double x, double* s, double* c) { ::sincos(x, s, c); }
template <int N> inline Jet<N> jsin(const Jet<N>& f) {
    double sn, cs; sincos_pair(f.a, &sn, &cs);
    Jet<N> h; h.a = sn; const double d = cs;
    for (int i = 0; i < N; i++) h.v[i] = f.v[i] * d; return h; }
template <int N> inline Jet<N> jcos(const Jet<N>& f) {
    double sn, cs; sincos_pair(f.a, &sn, &cs);
    Jet<N> h; h.a = cs; const double d = -sn;
    for (int i = 0; i < N; i++) h.v[i] = f.v[i] * d; return h; }
template <int N> inline Jet<N> jatan2(const Jet<N>& y, const Jet<N>& x) {
    // d atan2(y,x) = (x dy - y dx) / (x^2 + y^2)
    Jet<N> h; h.a = std::atan2(y.a, x.a); const double d = 1.0 / (x.a * x.a + y.a * y.a);
    for (int i = 0; i < N; i++) h.v[i] = (x.a * y.v[i] - y.a * x.v[i]) * d; return h; }

// scalar overloads so the same templated residual code runs on plain doubles
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jsin(double x) { double sn, cs; sincos_pair(x, &sn, &cs); return sn; }
inline double jcos(double x) { double sn, cs; sincos_pair(x, &sn, &cs); return cs; }
inline double jatan2(double y, double x) { return std::atan2(y, x); }

template <typename T> struct JetTraits { static double value(const T& x) { return x.a; } };
template <> struct JetTraits<double> { static double value(const double& x) { return x; } };

}  // namespace oracle
