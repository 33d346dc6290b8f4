// linesearch.hpp — host-side helpers of the Armijo line search ceres runs inside its trust-region loop for
// bound-constrained problems (TrustRegionMinimizer::DoLineSearch -> ArmijoLineSearch, CUBIC interpolation;
// Ceres 2.1.0, out of tree, restated).  Reference call site: the bounds of src/CameraCalibration.cpp:943-952.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace lifcal {

struct LsSample { double x = 0, value = 0, gradient = 0; bool value_valid = false, gradient_valid = false; };

inline double ls_polyval(const std::vector<double>& c, double x) { double v = 0; for (double a : c) v = v * x + a; return v; }

// polynomial through the samples' values and gradients (decreasing powers), full-pivot elimination
inline std::vector<double> ls_interpolate(const std::vector<LsSample>& smp) {
  int m = 0; for (auto& s : smp) m += (s.value_valid ? 1 : 0) + (s.gradient_valid ? 1 : 0);
  const int deg = m - 1;
  std::vector<double> A((size_t)m * m, 0.0), b(m, 0.0);
  int row = 0;
  for (auto& s : smp) {
    if (s.value_valid) { for (int j = 0; j <= deg; ++j) A[(size_t)row * m + j] = std::pow(s.x, deg - j); b[row++] = s.value; }
    if (s.gradient_valid) { for (int j = 0; j < deg; ++j) A[(size_t)row * m + j] = (deg - j) * std::pow(s.x, deg - j - 1); b[row++] = s.gradient; }
  }
  std::vector<int> perm(m); for (int i = 0; i < m; ++i) perm[i] = i;
  for (int k = 0; k < m; ++k) {
    int pr = k, pc = k; double best = 0;
    for (int i = k; i < m; ++i) for (int j = k; j < m; ++j) if (std::fabs(A[(size_t)i * m + j]) > best) { best = std::fabs(A[(size_t)i * m + j]); pr = i; pc = j; }
    if (best == 0) break;
    for (int j = 0; j < m; ++j) std::swap(A[(size_t)k * m + j], A[(size_t)pr * m + j]);
    std::swap(b[k], b[pr]);
    for (int i = 0; i < m; ++i) std::swap(A[(size_t)i * m + k], A[(size_t)i * m + pc]);
    std::swap(perm[k], perm[pc]);
    for (int i = k + 1; i < m; ++i) { const double f = A[(size_t)i * m + k] / A[(size_t)k * m + k]; for (int j = k; j < m; ++j) A[(size_t)i * m + j] -= f * A[(size_t)k * m + j]; b[i] -= f * b[k]; }
  }
  std::vector<double> y(m, 0.0), c(m, 0.0);
  for (int i = m - 1; i >= 0; --i) { double s = b[i]; for (int j = i + 1; j < m; ++j) s -= A[(size_t)i * m + j] * y[j]; y[i] = A[(size_t)i * m + i] != 0 ? s / A[(size_t)i * m + i] : 0.0; }
  for (int i = 0; i < m; ++i) c[perm[i]] = y[i];
  return c;
}

inline void ls_root_real_parts(std::vector<double> c, std::vector<double>* roots) {
  roots->clear();
  while (!c.empty() && c.front() == 0.0) c.erase(c.begin());
  const int d = (int)c.size() - 1;
  if (d < 1) return;
  if (d == 1) { roots->push_back(-c[1] / c[0]); return; }
  if (d == 2) {
    const double D = c[1] * c[1] - 4 * c[0] * c[2];
    if (D >= 0) { const double sq = std::sqrt(D); const double q = -0.5 * (c[1] + (c[1] >= 0 ? sq : -sq)); roots->push_back(q / c[0]); if (q != 0) roots->push_back(c[2] / q); }
    else { roots->push_back(-c[1] / (2 * c[0])); roots->push_back(-c[1] / (2 * c[0])); }
    return;
  }
  std::vector<std::complex<double>> z(d), a(d + 1);
  for (int i = 0; i <= d; ++i) a[i] = c[i] / c[0];
  double rad = 0; for (int i = 1; i <= d; ++i) rad = std::max(rad, std::abs(a[i])); rad = 1 + rad;
  for (int i = 0; i < d; ++i) z[i] = std::polar(rad * 0.7, 2 * M_PI * i / d + 0.4);
  for (int it = 0; it < 500; ++it) {   // Durand–Kerner
    double change = 0;
    for (int i = 0; i < d; ++i) {
      std::complex<double> pv = a[0]; for (int k = 1; k <= d; ++k) pv = pv * z[i] + a[k];
      std::complex<double> den = 1; for (int j = 0; j < d; ++j) if (j != i) den *= (z[i] - z[j]);
      if (std::abs(den) == 0) continue;
      const std::complex<double> dz = pv / den; z[i] -= dz; change = std::max(change, std::abs(dz));
    }
    if (change < 1e-15 * rad) break;
  }
  for (auto& r : z) roots->push_back(r.real());
}

// ceres MinimizeInterpolatingPolynomial: interval ends, midpoint and the stationary points inside the interval
inline double ls_minimize(const std::vector<LsSample>& smp, double x_min, double x_max) {
  const std::vector<double> poly = ls_interpolate(smp);
  double best_x = 0.5 * (x_min + x_max), best_v = ls_polyval(poly, best_x);
  const double vmin = ls_polyval(poly, x_min); if (vmin < best_v) { best_v = vmin; best_x = x_min; }
  const double vmax = ls_polyval(poly, x_max); if (vmax < best_v) { best_v = vmax; best_x = x_max; }
  if (poly.size() <= 2) return best_x;
  std::vector<double> der; const int deg = (int)poly.size() - 1;
  for (int i = 0; i < deg; ++i) der.push_back(poly[i] * (deg - i));
  std::vector<double> roots; ls_root_real_parts(der, &roots);
  for (double r : roots) { if (r < x_min || r > x_max) continue; const double v = ls_polyval(poly, r); if (v < best_v) { best_v = v; best_x = r; } }
  return best_x;
}

}  // namespace lifcal
