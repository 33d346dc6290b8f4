// Microbenchmark: cost of LDS ds_add_f64 per wave-instruction under different address patterns (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(int pattern, int iters, double* out, unsigned long long* cyc) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 8192; i += 256) sm[i] = 0.0;
  __syncthreads();
  int addr;
  switch (pattern) {
    case 0: addr = tid; break;                          // distinct, contiguous (conflict-free)
    case 1: addr = (lane / 6) * 97 + w * 1024; break;   // 6-way same address inside a wave
    case 2: addr = (lane % 11) * 36 + w * 7; break;     // ~6 lanes per address, stride 36 doubles (current layout)
    case 3: addr = lane * 36 + w; break;                // distinct addresses, stride 36 doubles (bank pattern of blocks)
    case 4: addr = (lane / 16) * 131 + w * 1024; break; // 16-way same address
    case 5: addr = lane * 33 + w * 2100; break;         // distinct, odd stride (conflict-free banks)
    default: addr = 0; break;                           // 64-way same address
  }
  unsigned long long t0 = clock64();
  double v = 1.0 + lane;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) atomicAdd(&sm[addr + u * 0], v);
  }
  __syncthreads();
  unsigned long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (tid < 64) out[blockIdx.x * 64 + tid] = sm[tid];
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 64 * 8); hipMalloc(&cyc, 256 * 8);
  const char* names[] = {"distinct contiguous", "6-way same addr", "11 addrs stride36 (~6-way)", "distinct stride 36", "16-way same addr", "distinct stride 33", "64-way same addr"};
  for (int waves = 1; waves <= 4; waves *= 4)
    for (int p = 0; p < 7; ++p) {
      const int iters = 200;
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 8192 * 8, 0, p, iters, out, cyc);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 8192 * 8, 0, p, iters, out, cyc);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double mean = 0; for (auto c : h) mean += c; mean /= 256;
      printf("waves/WG %d  %-28s  %8.1f cycles per wave-instruction (per wave), %8.1f per CU-instruction\n", waves, names[p], mean / (iters * 8.0), mean / (iters * 8.0 * waves));
    }
  return 0;
}
