// Microbenchmark: cycles per v_mfma_f64_16x16x4_f64 (1..8 independent accumulators, operands in registers), and
// the f64 FMA VALU rate, one wave per SIMD (256-thread workgroups, 1 per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NA>
__global__ __launch_bounds__(256) void k_mfma(int iters, double* out, unsigned long long* cyc) {
  v4f64 acc[NA];
  for (int q = 0; q < NA; ++q) acc[q] = v4f64{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
  unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NA; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  unsigned long long t1 = clock64();
  double s = 0; for (int q = 0; q < NA; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_fma(int iters, double* out, unsigned long long* cyc) {
  double x[16];
  for (int q = 0; q < 16; ++q) x[q] = threadIdx.x * 1e-3 + q;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = fma(x[q], m, c);
  }
  unsigned long long t1 = clock64();
  double s = 0; for (int q = 0; q < 16; ++q) s += x[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K> double run(K kern, int iters, double* out, unsigned long long* cyc) {
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, iters, out, cyc); (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, iters, out, cyc); (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256); (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto c : h) m += c; return m / 256;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  printf("mfma_f64_16x16x4, 1 acc: %.1f cycles per MFMA\n", run(k_mfma<1>, iters, out, cyc) / iters / 1);
  printf("mfma_f64_16x16x4, 2 acc: %.1f cycles per MFMA\n", run(k_mfma<2>, iters, out, cyc) / iters / 2);
  printf("mfma_f64_16x16x4, 4 acc: %.1f cycles per MFMA\n", run(k_mfma<4>, iters, out, cyc) / iters / 4);
  printf("mfma_f64_16x16x4, 8 acc: %.1f cycles per MFMA\n", run(k_mfma<8>, iters, out, cyc) / iters / 8);
  printf("v_fma_f64 (16 independent chains): %.2f cycles per wave-instruction\n", run(k_fma, iters, out, cyc) / iters / 16);
  return 0;
}
