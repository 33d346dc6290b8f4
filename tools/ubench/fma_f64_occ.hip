// v_fma_f64 throughput vs occupancy: cycles per wave-instruction per SIMD at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fma(int iters, double* out, unsigned long long* cyc) {
  double x[16];
  for (int q = 0; q < 16; ++q) x[q] = threadIdx.x * 1e-3 + q;
  const double m = 1.0000001, c = 1e-9;
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = fma(x[q], m, c);
  }
  __syncthreads();
  unsigned long long t1 = clock64();
  double s = 0; for (int q = 0; q < 16; ++q) s += x[q];
  out[blockIdx.x * BLOCK + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int BLOCK> void run(const char* name) {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * BLOCK * 8); (void)hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_fma<BLOCK>, dim3(256), dim3(BLOCK), 0, 0, iters, out, cyc); (void)hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(256); (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto c : h) m += c; m /= 256;
  const double waves_per_simd = BLOCK / 256.0;
  printf("%s: %.2f cycles per wave-instruction per wave, %.2f cycles per wave-instruction per SIMD\n", name, m / iters / 16, m / iters / 16 / waves_per_simd);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() { run<256>("1 wave/SIMD "); run<512>("2 waves/SIMD"); run<1024>("4 waves/SIMD"); return 0; }
