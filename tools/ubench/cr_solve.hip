// Drives bandchol3.hpp (block odd-even reduction of the band + arrow system) on random SPD systems: S = L L^T with L
// block-banded + arrow, rhs = S x_true; reports the error against x_true and the time of the launch sequence.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/cr_solve.out tools/ubench/cr_solve.hip
// Run:   tools/ubench/cr_solve.out F bw NA [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../lifcal_amd/csrc/bandchol3.hpp"

using namespace lifcal;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const uint32_t F = argc > 1 ? atoi(argv[1]) : 334, bw = argc > 2 ? atoi(argv[2]) : 9, NA = argc > 3 ? atoi(argv[3]) : 17;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  const uint32_t n = 6 * F + NA, ld = 6 * F + NA + 1;
  std::mt19937_64 rng(1234 + F * 31 + bw);
  std::normal_distribution<double> nd(0.0, 1.0);
  // L: pose row i (frame fi) has entries in the columns of frames [fi - bw, fi]; arrow rows are dense
  std::vector<double> L((size_t)n * n, 0.0);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t k0 = i < 6 * F ? 6 * (i / 6 >= bw ? i / 6 - bw : 0) : 0;
    const double sc = 0.5 / std::sqrt((double)(i - k0 + 1));   // keeps L (a random triangular matrix) well conditioned
    for (uint32_t k = k0; k < i; ++k) L[(size_t)i * n + k] = sc * nd(rng);
    L[(size_t)i * n + i] = 1.0 + std::fabs(nd(rng));
  }
  auto S = [&](uint32_t i, uint32_t j) {   // i >= j
    double s = 0.0;
    const uint32_t k0 = i < 6 * F ? 6 * (i / 6 >= bw ? i / 6 - bw : 0) : 0;
    for (uint32_t k = k0; k <= j; ++k) s += L[(size_t)i * n + k] * L[(size_t)j * n + k];
    return s;
  };
  std::vector<double> Sband((size_t)F * (bw + 1) * 36, 0.0), Sarrow((size_t)(NA + 1) * ld, 0.0), xt(n), rhs(n, 0.0);
  for (uint32_t f = 0; f < F; ++f)
    for (uint32_t dd = 0; dd <= std::min(bw, f); ++dd)
      for (uint32_t a = 0; a < 6; ++a)
        for (uint32_t b = 0; b < 6; ++b) {
          if (dd == 0 && b > a) continue;
          Sband[((size_t)f * (bw + 1) + dd) * 36 + a * 6 + b] = S(6 * f + a, 6 * (f - dd) + b);
        }
  for (uint32_t a = 0; a < NA; ++a)
    for (uint32_t j = 0; j <= 6 * F + a; ++j) Sarrow[(size_t)a * ld + j] = S(6 * F + a, j);
  for (uint32_t i = 0; i < n; ++i) xt[i] = nd(rng);
  {  // rhs = L (L^T x)
    std::vector<double> y(n, 0.0);
    for (uint32_t k = 0; k < n; ++k) { double s = 0.0; for (uint32_t i = k; i < n; ++i) s += L[(size_t)i * n + k] * xt[i]; y[k] = s; }
    for (uint32_t i = 0; i < n; ++i) { double s = 0.0; for (uint32_t k = 0; k <= i; ++k) s += L[(size_t)i * n + k] * y[k]; rhs[i] = s; }
  }
  for (uint32_t j = 0; j < n; ++j) Sarrow[(size_t)NA * ld + j] = rhs[j];

  CrPlan plan;
  if (!cr_plan(plan, F, bw, NA)) { printf("F=%u bw=%u NA=%u: not eligible\n", F, bw, NA); return 0; }
  double *dSb, *dSa, *dx, *dfail;
  CK(hipMalloc(&dSb, Sband.size() * 8)); CK(hipMalloc(&dSa, Sarrow.size() * 8)); CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dfail, 8));
  CK(hipMemcpy(dSb, Sband.data(), Sband.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dSa, Sarrow.data(), Sarrow.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dx, 0, n * 8)); CK(hipMemset(dfail, 0, 8));
  CK(hipMalloc(&plan.ws.P, cr_ws_doubles_P(plan.ws) * 8)); CK(hipMalloc(&plan.ws.U, cr_ws_doubles_U(plan.ws) * 8));
  CK(hipMalloc(&plan.ws.D, cr_ws_doubles_D(plan.ws) * 8)); CK(hipMalloc(&plan.ws.A, cr_ws_doubles_A(plan.ws) * 8));
  CK(hipMemset(plan.ws.D, 0xFF, cr_ws_doubles_D(plan.ws) * 8)); CK(hipMemset(plan.ws.A, 0xFF, cr_ws_doubles_A(plan.ws) * 8)); CK(hipMalloc(&plan.ws.x, (cr_ws_doubles_x(plan.ws) + 8 * 64 * 16) * 8));
  CK(hipMemset(plan.ws.P, 0xFF, cr_ws_doubles_P(plan.ws) * 8)); CK(hipMemset(plan.ws.U, 0xFF, cr_ws_doubles_U(plan.ws) * 8)); CK(hipMemset(plan.ws.x, 0xFF, cr_ws_doubles_x(plan.ws) * 8));
  CrSys sys{dSb, dSa, dx, dfail, F, bw, NA, ld};
  hipStream_t st; CK(hipStreamCreate(&st));
  cr_solve_launch(plan, sys, st);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(st));
  std::vector<double> x(n); double fail = 0.0;
  CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fail, dfail, 8, hipMemcpyDeviceToHost));
  double emax = 0.0, xmax = 0.0; bool bad = false;
  for (uint32_t i = 0; i < n; ++i) { if (!(x[i] == x[i])) bad = true; emax = std::max(emax, std::fabs(x[i] - xt[i])); xmax = std::max(xmax, std::fabs(xt[i])); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) cr_solve_launch(plan, sys, st);
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("F=%u bw=%u NA=%u n=%u nb=%u m=%u levels=%u nq=%d threads=%u lds=%zu/%zu/%zu: max|x-xt|/max|xt| = %.3e nan=%d fail=%g  %.1f us per solve (%d reps)\n",
         F, bw, NA, n, plan.ws.nb, plan.ws.m, plan.ws.levels, plan.nq, plan.fac_threads, plan.fac_lds, plan.fin_lds, plan.prod_lds, emax / xmax, (int)bad, fail, 1e3 * ms / reps, reps);
#ifdef CR_STAMPS
  {
    std::vector<unsigned long long> stp(8 * 64 * 16);
    CK(hipMemcpy(stp.data(), plan.ws.x + cr_ws_doubles_x(plan.ws), stp.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t l = 0; l < plan.ws.levels; ++l) {
      const unsigned long long* q = &stp[8 * (l * 64 + 0)];
      printf("  level %u block 0 ticks: gather %llu load %llu factor %llu store %llu\n", l, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3]);
      const unsigned long long* r = &stp[8 * (l * 64 + 32)];
      printf("     products kept block 1: own loads %llu zero %llu stage %llu gemm %llu wait %llu write %llu\n", r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5]);
    }
  }
#endif
  return (bad || emax / xmax > 1e-8) ? 1 : 0;
}
