// oracle/jet.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Forward-mode dual number with N derivative slots: the arithmetic ceres::Jet<double,N> performs
// when ceres::AutoDiffCostFunction differentiates the reference functor
// (reference src/BundleAdjustment/BundleAdjustment.h:199-222 instantiates
// AutoDiffCostFunction<OurCostFunctionBundle, 2, 17, 6, 3> -> Jet<double,26>).
// Ceres Solver 2.1.0 (pinned by reference installation/Dockerfile:105) is NOT in the container;
// this restates the published Jet rules (value + first derivatives, comparisons on the value).
#pragma once
#include <cmath>

namespace lo {

template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT: implicit like ceres::Jet
  Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
// ceres: h = f/g: g_a_inverse = 1/g.a; f_a_by_g_a = f.a*g_a_inverse; h.v = (f.v - f_a_by_g_a*g.v)*g_a_inverse
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h; }

template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { const double si = 1.0 / s; return f * si; }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) {
  Jet<N> h; const double gi = 1.0 / g.a; h.a = s * gi; const double m = -s * gi * gi;
  for (int i = 0; i < N; ++i) h.v[i] = m * g.v[i]; return h; }

template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline Jet<N>& operator-=(Jet<N>& f, const Jet<N>& g) { f = f - g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, const Jet<N>& g) { f = f * g; return f; }
template <int N> inline Jet<N>& operator/=(Jet<N>& f, const Jet<N>& g) { f = f / g; return f; }

// comparisons act on the scalar part only (ceres::Jet semantics) — this is what makes the
// reference's sign folding "if(fL < T(0.0)) fL = -fL;" (BundleAdjustment.h:123-133) flip the
// derivative together with the value.
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator<(const Jet<N>& f, double s) { return f.a < s; }
template <int N> inline bool operator>(const Jet<N>& f, double s) { return f.a > s; }

template <int N> inline Jet<N> sin(const Jet<N>& f) {
  Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline Jet<N> cos(const Jet<N>& f) {
  Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
// pow(jet, constant exponent p): d = p * f.a^(p-1) * f.v  (ceres::pow(Jet, double))
template <int N> inline Jet<N> pow(const Jet<N>& f, double p) {
  Jet<N> h; h.a = std::pow(f.a, p); const double d = p * std::pow(f.a, p - 1.0);
  for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }

inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double pow(double x, double p) { return std::pow(x, p); }

inline double scalar_of(double x) { return x; }
template <int N> inline double scalar_of(const Jet<N>& x) { return x.a; }

}  // namespace lo
