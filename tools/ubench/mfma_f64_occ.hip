// Microbenchmark (round 2, VERDICT "What's weak" 7): v_mfma_f64_16x16x4_f64 at 1 / 2 / 4 waves per SIMD, and next to a
// VALU-only wave on the same SIMD (do the matrix pipe and the vector ALU overlap?).  One workgroup per CU (256 blocks),
// BLOCK / 256 waves per SIMD; cycles from s_memtime (shader clock) around the loop of wave 0 of each role.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_occ.out mfma_f64_occ.hip && ./mfma_f64_occ.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// mode 0: every wave MFMA (4 independent accumulators); mode 1: every wave FMA (16 chains);
// mode 2: waves alternate per SIMD: wave index / 4 even -> MFMA, odd -> FMA (needs >= 2 waves per SIMD)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_mix(int mode, int iters, double* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && ((wave >> 2) & 1) == 0);
  v4f64 acc[4];
  for (int q = 0; q < 4; ++q) acc[q] = v4f64{0, 0, 0, 0};
  double x[16];
  for (int q = 0; q < 16; ++q) x[q] = threadIdx.x * 1e-3 + q;
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4, m = 1.0000001, c = 1e-9;
  __syncthreads();
  const unsigned long long t0 = clock64();
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = fma(x[q], m, c);
    }
  }
  const unsigned long long t1 = clock64();
  double s = 0;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  for (int q = 0; q < 16; ++q) s += x[q];
  out[(size_t)blockIdx.x * BLOCK + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 16 + wave] = t1 - t0;   // per wave: its own loop time
}

template <int BLOCK> void run(int mode, const char* name) {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)256 * BLOCK * 8); (void)hipMalloc(&cyc, 256 * 16 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_mix<BLOCK>, dim3(256), dim3(BLOCK), 0, 0, mode, iters, out, cyc); (void)hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(256 * 16); (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  const int nw = BLOCK / 64, wps = nw / 4;
  double tm = 0, tf = 0; int nm = 0, nf = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) {
    const bool mf = mode == 0 || (mode == 2 && ((w >> 2) & 1) == 0);
    if (mf) { tm += h[b * 16 + w]; ++nm; } else { tf += h[b * 16 + w]; ++nf; }
  }
  printf("%-34s %d wave(s)/SIMD:", name, wps);
  if (nm) { const int mw = (mode == 2) ? wps / 2 : wps; printf("  MFMA %.1f cycles per instruction per wave = %.1f per SIMD (%d MFMA wave(s)/SIMD)", tm / nm / iters / 4, tm / nm / iters / 4 / mw, mw); }
  if (nf) { const int fw = (mode == 2) ? wps / 2 : wps; printf("  FMA %.2f cycles per wave-instruction per wave = %.2f per SIMD (%d FMA wave(s)/SIMD)", tf / nf / iters / 16, tf / nf / iters / 16 / fw, fw); }
  printf("\n");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<256>(0, "mfma only"); run<512>(0, "mfma only"); run<1024>(0, "mfma only");
  run<256>(1, "fma only"); run<512>(1, "fma only"); run<1024>(1, "fma only");
  run<512>(2, "mfma wave + fma wave per SIMD"); run<1024>(2, "2 mfma waves + 2 fma waves per SIMD");
  return 0;
}
