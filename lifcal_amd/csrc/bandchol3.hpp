// bandchol3.hpp — the band + arrow Cholesky as a BLOCK ODD-EVEN REDUCTION (nested-dissection order) over many workgroups.
//
// bandchol.hpp / bandchol2.hpp walk the F pose blocks as a chain: F steps (twisted: F/2) of ~5 k cycles on one or two
// workgroups, 0.36 ms of factorisation + 0.14 ms of back-substitution per LM iteration at F = 334, 254 CUs idle.  A matrix whose
// frame band width is bw is block TRIDIAGONAL in super-blocks of bw frames (m = 6 bw columns).  Eliminating every second
// super-block is independent work — one workgroup each — and leaves a block-tridiagonal system of half the size: log2(F / bw)
// levels instead of F steps.  It is the same Cholesky factorisation under another (odd-even) elimination order, so it is as
// stable as the chain; the price is fill (the couplings between kept blocks become dense m x m blocks): ~8x the flops of the
// chain, spread over up to F / (2 bw) workgroups.
//
// Per level l (s = 2^l; active blocks are the multiples of s, the odd multiples are eliminated), two launches:
//   k_cr_factor    one workgroup per eliminated block e: gathers its current diagonal block D_e, its couplings to the new
//                  neighbours l = e - s and r = e + s (original band at level 0, the previous level's product otherwise) and its
//                  arrow rows into one panel [D; C_l^T; C_r; A | rhs; I], and runs a right-looking Cholesky on it: D = L L^T,
//                  P = X L^-T for all rows below, the identity rows turn into L^-T (back-substitution becomes a matrix-vector
//                  product).  The panel lives in REGISTERS (4 rows x NQ column quads per thread, columns dealt cyclically to the
//                  four waves); only the pivot column travels through LDS, double-buffered: one barrier per column.  The step
//                  code is fully unrolled over the column index with static register indices and NO guards (the diagonal block
//                  is padded with identity columns to 4 NQ): ~60 instructions per column instead of ~450 with run-time guards.
//   k_cr_products  (fp64 MFMA) per KEPT block b one workgroup: D_b -= P_r(b-s) P_r(b-s)^T + P_l(b+s) P_l(b+s)^T and the same for its arrow
//                  rows (every kept block has ONE writer and a fixed summation order: the factorisation is bitwise
//                  reproducible — all ranks of a multi-GPU run factor the same replicated system and must agree); per
//                  ELIMINATED block e one workgroup: the coupling of its neighbours U_rl = P_r P_l^T (+ transpose) and its share
//                  P_a P_a^T of the arrow block.
//   k_cr_final     block 0 + arrow block (dense, m + NA <= 72 columns) with the same register panel; solves for x_0 and x_arrow.
//   k_cr_backsub   per level, top down: x_e = L^-T (y_e - P_l^T x_l - P_r^T x_r - P_a^T x_arrow), one workgroup per block.
// Frames beyond F (the last super-block is padded to bw frames) are identity rows.  The rhs travels as the last arrow row, so the
// forward substitution is part of the factorisation, as in bandchol.hpp.
// Every gather is written as batches of independent, unconditional loads (16 rows per wave in flight): a panel read with one
// load per loop iteration was ~50 dependent L2 round trips.
// Self-contained (no Dev): tools/ubench/cr_solve.hip drives it on random systems; tools/cr_prototype.py is the same scheme in
// numpy (tests/test_cr_prototype.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace lifcal {

#ifndef LIFCAL_DEV
#define LIFCAL_DEV __device__ __forceinline__
#endif

struct CrSys {            // the damped reduced system as k_finalize leaves it (kernels.hpp: Dev::Sband / Sarrow), and where the step goes
  const double* Sband;    // [F][bw+1][6][6]: block (row frame f, column frame f - dd); dd = 0: lower triangle valid
  const double* Sarrow;   // [NA+1][ld]: arrow rows (promoted | camera), last row = rhs; columns 6F.. = arrow block (lower valid)
  double* delta_red;      // 6F + NA: the solution
  double* fail;           // raised to 1.0 on a non-positive pivot
  uint32_t F, bw, NA, ld;
};

struct CrWs {
  double *P, *D, *A, *U, *x;   // panels [nb][prow][m] (P_l | P_r | P_a | L^-T); current diagonal blocks [nb][m][m] and arrow rows
                               // [nb][nax][m] of the kept blocks; per eliminated block U_rl | U_lr | U_aa [nb][ustride]; solution [nb m + nax]
  uint32_t nb, m, nax, prow, ustride, levels;
  __host__ __device__ uint32_t off_rl() const { return 0; }
  __host__ __device__ uint32_t off_lr() const { return m * m; }
  __host__ __device__ uint32_t off_aa() const { return 2 * m * m; }
};

constexpr uint32_t CR_PROD_THREADS = 768;

inline CrWs cr_geometry(uint32_t F, uint32_t bw, uint32_t NA) {
  CrWs w{};
  w.m = 6 * bw; w.nax = NA + 1; w.nb = bw ? (F + bw - 1) / bw : 0;
  w.prow = 3 * w.m + w.nax; w.ustride = 2 * w.m * w.m + w.nax * w.nax;
  w.levels = 0; while ((1u << w.levels) < w.nb) ++w.levels;
  return w;
}
// column quads per thread of the factor kernel: the smallest instantiated size that holds m columns (the rest is identity padding)
inline int cr_factor_nq(uint32_t m) {
  const int need = (int)((m + 3) / 4);
  for (int nq : {3, 6, 9, 12, 14, 15}) if (nq >= need) return nq;
  return 0;
}
inline uint32_t cr_threads(uint32_t nrows) { return nrows <= 256 ? 512u : 1024u; }   // 512 threads: two rows per thread (nrows <= 256), four column parts
inline size_t cr_factor_lds(uint32_t nrows, int NQ) {
  return ((size_t)nrows * (4 * NQ + 1) + 2 + 2 * (256 + 4 * (size_t)NQ) + 8) * sizeof(double);   // panel | two column buffers (256 rows + the permuted block part)
}
inline size_t cr_products_lds(const CrWs& w) {   // k-major operand copies, row strides 80 (<= 64 rows) and 48 (<= 32 rows) doubles
  const size_t mk = (w.m + 3u) & ~3u;
  return std::max((size_t)2 * w.m * (80 + 48), mk * (2 * 80 + 48)) * sizeof(double);
}
inline bool cr_eligible(uint32_t F, uint32_t bw, uint32_t NA) {
  if (bw < 1 || NA > 31 || NA < 1 || 6 * bw > 64) return false;
  const CrWs w = cr_geometry(F, bw, NA);
  const int nq = cr_factor_nq(w.m);
  if (w.nb < 4 || nq == 0) return false;
  const uint32_t nrows = 4 * nq + w.prow;
  return cr_factor_lds(nrows, nq) <= 160 * 1024 && nrows <= 256 && cr_products_lds(w) <= 160 * 1024;
}
__host__ __device__ inline uint32_t cr_n_active(uint32_t nb, uint32_t level) { return ((nb - 1) >> level) + 1; }   // multiples of 2^level below nb
inline uint32_t cr_n_elim(uint32_t nb, uint32_t level) { return cr_n_active(nb, level) / 2; }                        // the odd ones

LIFCAL_DEV void cr_pin(double& v) { asm volatile("" : "+v"(v)); }   // the value exists in a register HERE: no sinking / hoisting across
LIFCAL_DEV void cr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int I, int N, class Fn>
LIFCAL_DEV void cr_static_for(Fn&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); cr_static_for<I + 1, N>(f); }
}

// value = *p * mul + add with p always a valid address (branch-free gathers: the loads of a batch are all in flight together)
struct CrSrc { const double* p; double mul, add; };
// element (6 fi + a, 6 fj + b) of the pose x pose part, either triangle; frames >= F are padding (identity)
LIFCAL_DEV CrSrc cr_src_S(const CrSys& s, uint32_t fi, uint32_t a, uint32_t fj, uint32_t b) {
  const bool pad = fi >= s.F || fj >= s.F;
  const bool sw = fi < fj || (fi == fj && a < b);
  const uint32_t gi = sw ? fj : fi, gj = sw ? fi : fj, ga = sw ? b : a, gb = sw ? a : b;
  const uint32_t dd = gi - gj;
  const bool in = !pad && dd <= s.bw;
  CrSrc r;
  r.p = s.Sband + (in ? ((size_t)gi * (s.bw + 1) + dd) * 36 + ga * 6 + gb : 0);
  r.mul = in ? 1.0 : 0.0;
  r.add = (pad && fi == fj && a == b) ? 1.0 : 0.0;
  return r;
}

// Right-looking Cholesky of a register panel of 4 NQ columns.  Thread (rg, cp) of NRG x 4 owns rows rg + i NRG (i < 4) and
// columns 4 cl + cp (cl < NQ); cp is wave-uniform.  Rows 0..4NQ-1 are the symmetric diagonal block (lower triangle used), every
// other row just follows: afterwards row r holds (X L^-T)[r][:], the diagonal block's rows hold L.
// cb: two column buffers of BS = 4 NRG + 4 NQ doubles: [0, 4 NRG) the column by row, then the diagonal block's part of it
// permuted to (r & 3) NQ + (r >> 2), so that the values a thread needs for ITS columns are contiguous.
// The step code is fully unrolled over the column index (static register indices, no guards: the diagonal block is padded to
// 4 NQ columns) and IDENTICAL for the four waves: what depends on the wave's column part cp is data — the segment of the column
// buffer it reads, a 0 / 1 factor on the one quad that is partly final — not control flow, so the waves, which run in step
// (a barrier per column), share every instruction-cache line of the ~30 KB of straight-line code.
// Measured alternatives: cp as a template parameter (four copies, 380 KB at NQ = 18: the waves stream four different copies
// through a 64 KB instruction cache, 81 us for 72 columns); cp in uniform branches around every block (the compiler moves the
// blocks out of line and spills the column values: 2.4 KB of scratch per lane); a rolled loop over the quads with the registers
// shifted down after every quad and a fall-through switch over the live quads (AGPR traffic: 97 us per panel).
// (Measured and dropped: deferring the updates nobody waits for behind the next barrier, so that they run under the next
// column's LDS reads — same time, 36 more registers: the step is bound by instruction issue of the single wave per SIMD.)
template <int NQ, int RT>
LIFCAL_DEV bool cr_panel_factor(double (&x)[RT][NQ], uint32_t rg, uint32_t cp, uint32_t NRG, double* cb) {
  constexpr uint32_t NC = 4 * NQ;
  const uint32_t NAT = RT * NRG, BS = NAT + NC;
  bool fail = false;
  auto write_col = [&](double* buf, int k) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const uint32_t r = rg + (uint32_t)i * NRG;
      buf[r] = x[i][k];
      if (r < NC) buf[NAT + (r & 3u) * NQ + (r >> 2)] = x[i][k];
    }
  };
  if (cp == 0) write_col(cb, 0);
  const uint32_t segoff = NAT + cp * NQ;
  cr_static_for<0, NQ>([&](auto clc) {
    constexpr int cl = decltype(clc)::value;
    cr_static_for<0, 4>([&](auto cpc) {
      constexpr int cpj = decltype(cpc)::value;
      constexpr uint32_t j = 4u * cl + cpj;
      double* buf = cb + (j & 1u) * BS;
      cr_lds_barrier();
      double piv = buf[NAT + cpj * NQ + cl];
      double rj[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) rj[i] = buf[rg + (uint32_t)i * NRG];
      const double* seg = buf + segoff;
      double cv[NQ];
      cr_static_for<cl, NQ>([&](auto kc) { constexpr int k = decltype(kc)::value; cv[k] = seg[k]; });
      if (!(piv > 0.0)) { piv = 1.0; fail = true; }
      const double is = rsqrt(piv), inv = is * is;
      // quad cl: the columns left of the pivot (and the pivot column) are final — their waves multiply by 0
      const double own = cp > (uint32_t)cpj ? 1.0 : 0.0;
      double a[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) a[i] = rj[i] * inv;
      constexpr int cl1 = (cpj == 3) ? cl + 1 : cl;       // the next pivot column: quad cl1 of column part cp1
      constexpr uint32_t cp1 = (cpj + 1) & 3;
      if constexpr (cl1 < NQ) {   // its quad goes first: it is what every wave waits for
        const double f1 = (cl1 == cl) ? own : 1.0;
#pragma unroll
        for (int i = 0; i < RT; ++i) x[i][cl1] -= (a[i] * f1) * cv[cl1];
        if (cp == cp1) write_col(cb + ((j + 1u) & 1u) * BS, cl1);
      }
      cr_static_for<cl, NQ>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k != cl1) {
          const double fk = (k == cl) ? own : 1.0;
#pragma unroll
          for (int i = 0; i < RT; ++i) x[i][k] -= (a[i] * fk) * cv[k];
        }
      });
      if (cp == (uint32_t)cpj) {
#pragma unroll
        for (int i = 0; i < RT; ++i) x[i][cl] = rj[i] * is;
      }
      // the updated entries are pinned here (an empty asm that "modifies" them): left alone, the optimiser sinks the updates
      // of a column down to their first use, several columns later, and keeps every (a, column value) pair alive until then —
      // kilobytes of spills
      cr_static_for<cl, NQ>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int i = 0; i < RT; ++i) cr_pin(x[i][k]);
      });
    });
  });
  cr_lds_barrier();
  return fail;
}

// registers <-> LDS panel pan[r * ldp + c]
template <int NQ, int RT>
LIFCAL_DEV void cr_panel_load(double (&x)[RT][NQ], const double* pan, uint32_t ldp, uint32_t rg, uint32_t cp, uint32_t NRG, uint32_t nrows) {
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const uint32_t r = rg + (uint32_t)i * NRG;
#pragma unroll
    for (int cl = 0; cl < NQ; ++cl) x[i][cl] = (r < nrows) ? pan[(size_t)r * ldp + 4u * cl + cp] : 0.0;
  }
}
template <int NQ, int RT>
LIFCAL_DEV void cr_panel_store(const double (&x)[RT][NQ], double* pan, uint32_t ldp, uint32_t rg, uint32_t cp, uint32_t NRG, uint32_t nrows) {
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const uint32_t r = rg + (uint32_t)i * NRG;
#pragma unroll
    for (int cl = 0; cl < NQ; ++cl) if (r < nrows) pan[(size_t)r * ldp + 4u * cl + cp] = x[i][cl];
  }
}

// LDS panel rows: [0, MP) D_e padded with identity to MP = 4 NQ | [MP, MP + m) C_l^T (rows: variables of l) | C_r (rows: variables
// of r) | the nax arrow rows of e's columns (last = rhs) | identity (m rows); columns 0..MP-1.  What goes to P is rows MP.. x columns 0..m-1.
template <int NQ, int TPB>
__global__ __launch_bounds__(TPB) void k_cr_factor(CrSys s, CrWs w, uint32_t level) {
  extern __shared__ __attribute__((aligned(16))) double crl[];
  constexpr uint32_t MP = 4 * NQ, ldp = MP + 1;
  const uint32_t T = blockDim.x, NRG = T / 4, tid = threadIdx.x, rg = tid % NRG;
  const uint32_t cp = __builtin_amdgcn_readfirstlane(tid / NRG);   // wave-uniform: NRG is a multiple of 64
  const uint32_t m = w.m, nax = w.nax, nrows = MP + w.prow, bw = s.bw;
  // level == w.levels: the last block, 0, after every level's updates — an elimination without neighbours
  const bool last = level >= w.levels;
  const uint32_t st = 1u << level, e = last ? 0u : (2 * blockIdx.x + 1) << level;
  const bool has_l = !last, has_r = !last && e + st < w.nb;
  double* pan = crl; double* cb = crl + (((size_t)nrows * ldp + 1) & ~(size_t)1);
#ifdef CR_STAMPS
  unsigned long long cst[6];
#define CRSTAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cst[i]) :: "memory")
#else
#define CRSTAMP(i) do { } while (0)
#endif
  CRSTAMP(0);
  const uint32_t wv = tid >> 6, ln = tid & 63u, nwv = T >> 6;
  const double* dummy = s.Sband;
  const double* Dc = w.D + (size_t)e * m * m; const double* Ac = w.A + (size_t)e * nax * m;
  const double* Ulr = w.U + (size_t)(level && has_l ? e - st / 2 : e) * w.ustride + w.off_lr();
  const double* Url = w.U + (size_t)(level && has_r ? e + st / 2 : e) * w.ustride + w.off_rl();
  if (level > 0) {
    // every source is a plain row-major array: D_e, -U_lr of the block eliminated between l and e, -U_rl of the one between e and
    // r, A_e.  One wave per row, lanes over the columns (MP <= 64); the loads of all four segments go out before the first store.
    const bool cok = ln < MP, creal = ln < m;
    const uint32_t c = creal ? ln : 0;
    constexpr int RB = 16;
    for (uint32_t rb = wv; rb < max(m, nax); rb += RB * nwv) {
      double vd[RB], vl[RB], vr[RB], va[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const uint32_t r = min(rb + (uint32_t)u * nwv, m - 1), ra = min(rb + (uint32_t)u * nwv, nax - 1);
        vd[u] = Dc[(size_t)r * m + c];
        vl[u] = has_l ? Ulr[(size_t)r * m + c] : 0.0;
        vr[u] = has_r ? Url[(size_t)r * m + c] : 0.0;
        va[u] = Ac[(size_t)ra * m + c];
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const uint32_t r = rb + (uint32_t)u * nwv;
        if (r < m && cok) {
          pan[(size_t)r * ldp + ln] = creal ? vd[u] : 0.0;
          pan[(size_t)(MP + r) * ldp + ln] = creal ? -vl[u] : 0.0;
          pan[(size_t)(MP + m + r) * ldp + ln] = creal ? -vr[u] : 0.0;
          pan[(size_t)(MP + 2 * m + nax + r) * ldp + ln] = (r == ln) ? 1.0 : 0.0;
        }
        if (r < nax && cok) pan[(size_t)(MP + 2 * m + r) * ldp + ln] = creal ? va[u] : 0.0;
      }
    }
    if (cok) for (uint32_t r = m + wv; r < MP; r += nwv) pan[(size_t)r * ldp + ln] = (r == ln) ? 1.0 : 0.0;   // identity padding of D
  } else
  for (uint32_t c0 = 0; c0 < MP; c0 += 64) {
    const bool cok = c0 + ln < MP;
    const uint32_t c = cok ? c0 + ln : 0, cf = c / 6, cm = c % 6;
    const bool creal = c < m;
    for (uint32_t rb = wv; rb < nrows; rb += 16 * nwv) {
      CrSrc src[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const uint32_t r = min(rb + (uint32_t)u * nwv, nrows - 1);
        CrSrc q{dummy, 0.0, 0.0};
        if (r < MP) {
          if (r >= m || !creal) q.add = (r == c) ? 1.0 : 0.0;
          else q = cr_src_S(s, e * bw + r / 6, r % 6, e * bw + cf, cm);
        } else if (r < MP + m) {
          const uint32_t rr = r - MP;
          if (creal) q = cr_src_S(s, (e - 1) * bw + rr / 6, rr % 6, e * bw + cf, cm);
        } else if (r < MP + 2 * m) {
          const uint32_t rr = r - MP - m;
          if (has_r && creal) q = cr_src_S(s, (e + 1) * bw + rr / 6, rr % 6, e * bw + cf, cm);
        } else if (r < MP + 2 * m + nax) {
          const uint32_t a = r - MP - 2 * m, f = e * bw + cf;
          if (creal && f < s.F) q = CrSrc{s.Sarrow + (size_t)a * s.ld + 6 * f + cm, 1.0, 0.0};
        } else {
          q.add = (r - MP - 2 * m - nax == c) ? 1.0 : 0.0;
        }
        src[u] = q;
      }
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *src[u].p;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const uint32_t r = rb + (uint32_t)u * nwv;
        if (r < nrows && cok) pan[(size_t)r * ldp + c] = v[u] * src[u].mul + src[u].add;
      }
    }
  }
  __syncthreads();
  CRSTAMP(1);
  constexpr int RT = TPB == 512 ? 2 : 4;   // rows per thread: 512 threads = two waves per SIMD, half the rows each
  double x[RT][NQ];
  cr_panel_load<NQ, RT>(x, pan, ldp, rg, cp, NRG, nrows);
  CRSTAMP(2);
  const bool fail = cr_panel_factor<NQ, RT>(x, rg, cp, NRG, cb);
  CRSTAMP(3);
  cr_panel_store<NQ, RT>(x, pan, ldp, rg, cp, NRG, nrows);
  if (fail) *s.fail = 1.0;
  __syncthreads();
  double* out = w.P + (size_t)e * w.prow * m;
  for (uint32_t c0 = 0; c0 < m; c0 += 64) {
    const uint32_t c = c0 + ln;
    if (c < m) for (uint32_t r = wv; r < w.prow; r += nwv) out[(size_t)r * m + c] = pan[(size_t)(MP + r) * ldp + c];
  }
#ifdef CR_STAMPS
  CRSTAMP(4);
  if (tid == 0) for (int i = 0; i < 5; ++i) ((unsigned long long*)w.x)[(size_t)w.nb * m + nax + nax * nax + 8 * (level * 64 + blockIdx.x) + i] = cst[i];
#endif
}

// ---- products on the fp64 matrix pipe ----
// X Y^T in 16x16 tiles with v_mfma_f64_16x16x4_f64 from k-major LDS copies of the operands (Ls[k * ld + row]): per MFMA a lane
// reads ONE double of each operand for 16 multiply-adds of its own — a quarter of the LDS traffic of 4x4 register tiles, which
// had made the first version of this kernel LDS-bound (12 waves x 54 k-steps x 4 KB: 20 k cycles of the CU's LDS pipe).
// Operand map (as in the Schur product experiment of sweep3.hpp): A[i][k] = X[k0 + k][16 ti + i] in lane i + 16 k,
// B[k][j] = Y[k0 + k][16 tj + j] in lane j + 16 k, result entry (16 ti + 4 r + (lane >> 4), 16 tj + (lane & 15)) in acc[r].
// Row strides ld = 16 (mod 32) doubles: the four k-rows of an operand fetch fall on disjoint LDS banks per half-wave.
typedef double cr_v4f64 __attribute__((ext_vector_type(4)));
constexpr uint32_t CR_LDY = 80, CR_LDA = 48;   // operands of <= 64 rows (m <= 64) and <= 32 rows (nax <= 32)
LIFCAL_DEV cr_v4f64 cr_mfma_tile(const double* Xs, uint32_t ldx, const double* Ys, uint32_t ldy, uint32_t ti, uint32_t tj, uint32_t K, cr_v4f64 acc, uint32_t lane) {
  const double* xa = Xs + (size_t)(lane >> 4) * ldx + 16 * ti + (lane & 15u);
  const double* yb = Ys + (size_t)(lane >> 4) * ldy + 16 * tj + (lane & 15u);
  for (uint32_t k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[(size_t)k0 * ldx], yb[(size_t)k0 * ldy], acc, 0, 0, 0);
  return acc;
}
// rows wv, wv + nwv, ... of a row-major block G (n rows x m columns, stride m; null: nothing) into Ls[(koff + k) * ldl + r]; lanes = k
// (m <= 64).  The loads of ALL blocks of a kernel are issued before the first store (CR_STAGE_LOAD ... CR_STAGE_STORE).
template <int NU>
LIFCAL_DEV void cr_stage_load(const double* G, uint32_t n, uint32_t m, uint32_t wv, uint32_t nwv, uint32_t ln, const double* valid, double (&v)[NU]) {
#pragma unroll
  for (int u = 0; u < NU; ++u) { const uint32_t r = wv + (uint32_t)u * nwv; v[u] = *((G && r < n && ln < m) ? G + (size_t)r * m + ln : valid); }
}
template <int NU>
LIFCAL_DEV void cr_stage_store(const double* G, uint32_t n, uint32_t m, uint32_t wv, uint32_t nwv, uint32_t ln, double* Ls, uint32_t ldl, uint32_t koff, const double (&v)[NU]) {
#pragma unroll
  for (int u = 0; u < NU; ++u) { const uint32_t r = wv + (uint32_t)u * nwv; if (G && r < n && ln < m) Ls[(size_t)(koff + ln) * ldl + r] = v[u]; }
}

// blockIdx < nk: kept block b = 2 i s: D_b and A_b lose what its eliminated neighbours eL = b - s and eR = b + s take away: one
// accumulation chain per output tile over K = 2 m (eL's share first, then eR's: a fixed order); otherwise eliminated block e:
// U_rl (+ transpose) and U_aa.  12 waves, output tiles dealt round-robin.
__global__ __launch_bounds__(768) void k_cr_products(CrSys s, CrWs w, uint32_t level) {
  extern __shared__ __attribute__((aligned(16))) double crl[];
  const uint32_t tid = threadIdx.x, T = CR_PROD_THREADS, m = w.m, nax = w.nax, st = 1u << level;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63u, nwv = T >> 6;
  const bool last = level >= w.levels;   // block 0 as the last "eliminated" block: only its share of the arrow block
  const uint32_t nk = last ? 0u : (cr_n_active(w.nb, level) + 1) / 2;
  const uint32_t mt = (m + 15u) / 16, nat = (nax + 15u) / 16, mk = (m + 3u) & ~3u;
  constexpr int NUM = 6, NUA = 3;   // rows per wave of an m-row / nax-row block: 12 waves x 6 >= 64, 12 x 3 >= 32
  if (blockIdx.x < nk) {
    const uint32_t b = (2 * blockIdx.x) << level;
    const bool hasL = b >= st, hasR = b + st < w.nb;
    const double* PL = hasL ? w.P + (size_t)(b - st) * w.prow * m : nullptr;   // panel of eL: its P_r rows are b's variables
    const double* PR = hasR ? w.P + (size_t)(b + st) * w.prow * m : nullptr;   // panel of eR: its P_l rows are b's variables
    const double* valid = PL ? PL : PR;
    double* Ys = crl;                              // [2m][CR_LDY]: P_r(eL) | P_l(eR), k-major
    double* Xa = Ys + (size_t)2 * m * CR_LDY;      // [2m][CR_LDA]: P_a(eL) | P_a(eR)
    double v0[NUM], v1[NUM], v2[NUA], v3[NUA];
    cr_stage_load<NUM>(PL ? PL + (size_t)m * m : nullptr, m, m, wv, nwv, ln, valid, v0);
    cr_stage_load<NUM>(PR, m, m, wv, nwv, ln, valid, v1);
    cr_stage_load<NUA>(PL ? PL + 2 * (size_t)m * m : nullptr, nax, m, wv, nwv, ln, valid, v2);
    cr_stage_load<NUA>(PR ? PR + 2 * (size_t)m * m : nullptr, nax, m, wv, nwv, ln, valid, v3);
    // the block's own values: tile task q = wv + 12 t; D tiles first (mt x mt), then the A tiles (nat x mt)
    const uint32_t ntask = mt * mt + nat * mt;
    cr_v4f64 own[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t q = wv + 12u * t;
      if (q >= ntask) continue;
      const bool is_a = q >= mt * mt;
      const uint32_t qq = is_a ? q - mt * mt : q, ti = qq / mt, tj = qq - ti * mt, c = 16 * tj + (ln & 15u);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const uint32_t r = 16 * ti + 4 * r4 + (ln >> 4);
        double o = 0.0;
        if (c < m && r < (is_a ? nax : m)) {
          if (!is_a) {
            if (level == 0) { const CrSrc sq = cr_src_S(s, b * s.bw + r / 6, r % 6, b * s.bw + c / 6, c % 6); o = *sq.p * sq.mul + sq.add; }
            else o = w.D[(size_t)b * m * m + r * m + c];
          } else {
            if (level == 0) { const uint32_t f = b * s.bw + c / 6; o = f < s.F ? s.Sarrow[(size_t)r * s.ld + 6 * f + c % 6] : 0.0; }
            else o = w.A[(size_t)b * nax * m + r * m + c];
          }
        }
        own[t][r4] = o;
      }
    }
    for (uint32_t i = tid; i < 2 * m * (CR_LDY + CR_LDA); i += T) crl[i] = 0.0;
    __syncthreads();
    cr_stage_store<NUM>(PL ? PL + (size_t)m * m : nullptr, m, m, wv, nwv, ln, Ys, CR_LDY, 0, v0);
    cr_stage_store<NUM>(PR, m, m, wv, nwv, ln, Ys, CR_LDY, m, v1);
    cr_stage_store<NUA>(PL ? PL + 2 * (size_t)m * m : nullptr, nax, m, wv, nwv, ln, Xa, CR_LDA, 0, v2);
    cr_stage_store<NUA>(PR ? PR + 2 * (size_t)m * m : nullptr, nax, m, wv, nwv, ln, Xa, CR_LDA, m, v3);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t q = wv + 12u * t;
      if (q >= ntask) continue;   // wave-uniform
      const bool is_a = q >= mt * mt;
      const uint32_t qq = is_a ? q - mt * mt : q, ti = qq / mt, tj = qq - ti * mt, c = 16 * tj + (ln & 15u);
      cr_v4f64 acc = {0, 0, 0, 0};
      acc = cr_mfma_tile(is_a ? Xa : Ys, is_a ? CR_LDA : CR_LDY, Ys, CR_LDY, ti, tj, 2 * m, acc, ln);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const uint32_t r = 16 * ti + 4 * r4 + (ln >> 4);
        if (c < m && r < (is_a ? nax : m)) {
          const double vv = own[t][r4] - acc[r4];
          if (!is_a) w.D[(size_t)b * m * m + r * m + c] = vv; else w.A[(size_t)b * nax * m + r * m + c] = vv;
        }
      }
    }
    return;
  }
  const uint32_t e = last ? 0u : (2 * (blockIdx.x - nk) + 1) << level;
  const bool has_r = !last && e + st < w.nb;
  const double* P = w.P + (size_t)e * w.prow * m;
  // the last launch also adds up the arrow products of the blocks 1..nb-1 (coalesced: a thread per entry, the blocks dealt to
  // G thread groups, each group in ascending block order, the groups combined in group order: reproducible) for k_cr_arrow
  const uint32_t naa = nax * nax, G = last ? (T / naa > 4 ? 4u : T / naa) : 0u;
  double part_aa = 0.0;
  if (last && G > 0 && tid < G * naa) {
    const uint32_t g = tid / naa, ent = tid - g * naa;
    for (uint32_t e0 = 1 + g; e0 < w.nb; e0 += 16 * G) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const uint32_t ee = e0 + (uint32_t)u * G; v[u] = w.U[(size_t)(ee < w.nb ? ee : 1u) * w.ustride + w.off_aa() + ent]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) { const uint32_t ee = e0 + (uint32_t)u * G; if (ee < w.nb) part_aa += v[u]; }
    }
  }
  double* Xs = crl; double* Ys = Xs + (size_t)mk * CR_LDY; double* Za = Ys + (size_t)mk * CR_LDY;   // P_r | P_l | P_a, k-major, K padded to a multiple of 4
  double v0[NUM], v1[NUM], v2[NUA];
  cr_stage_load<NUM>(has_r ? P + (size_t)m * m : nullptr, m, m, wv, nwv, ln, P, v0);
  cr_stage_load<NUM>(P, m, m, wv, nwv, ln, P, v1);
  cr_stage_load<NUA>(P + 2 * (size_t)m * m, nax, m, wv, nwv, ln, P, v2);
  for (uint32_t i = tid; i < mk * (2 * CR_LDY + CR_LDA); i += T) crl[i] = 0.0;
  __syncthreads();
  cr_stage_store<NUM>(has_r ? P + (size_t)m * m : nullptr, m, m, wv, nwv, ln, Xs, CR_LDY, 0, v0);
  cr_stage_store<NUM>(P, m, m, wv, nwv, ln, Ys, CR_LDY, 0, v1);
  cr_stage_store<NUA>(P + 2 * (size_t)m * m, nax, m, wv, nwv, ln, Za, CR_LDA, 0, v2);
  __syncthreads();
  double* U = w.U + (size_t)e * w.ustride;
  const uint32_t nrl = has_r ? mt * mt : 0u, ntask = nrl + nat * nat;
  for (uint32_t q = wv; q < ntask; q += nwv) {
    const bool is_aa = q >= nrl;
    const uint32_t qq = is_aa ? q - nrl : q, tw = is_aa ? nat : mt, ti = qq / tw, tj = qq - ti * tw, c = 16 * tj + (ln & 15u);
    cr_v4f64 acc = {0, 0, 0, 0};
    acc = cr_mfma_tile(is_aa ? Za : Xs, is_aa ? CR_LDA : CR_LDY, is_aa ? Za : Ys, is_aa ? CR_LDA : CR_LDY, ti, tj, mk, acc, ln);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const uint32_t r = 16 * ti + 4 * r4 + (ln >> 4);
      if (is_aa) { if (r < nax && c < nax) U[w.off_aa() + r * nax + c] = acc[r4]; }
      else if (r < m && c < m) { U[w.off_rl() + r * m + c] = acc[r4]; U[w.off_lr() + c * m + r] = acc[r4]; }
    }
  }
  if (last) {
    __syncthreads();
    double* part = crl;   // [G][naa]
    if (G > 0 && tid < G * naa) part[tid] = part_aa;
    __syncthreads();
    if (tid < naa) {
      double sacc = 0.0;
      if (G > 0) { for (uint32_t g = 0; g < G; ++g) sacc += part[g * naa + tid]; }
      else { for (uint32_t ee = 1; ee < w.nb; ++ee) sacc += w.U[(size_t)ee * w.ustride + w.off_aa() + tid]; }
      w.x[(size_t)w.nb * m + nax + tid] = sacc;
    }
    for (uint32_t t2 = tid + T; t2 < naa; t2 += T) {   // (nax^2 <= 1024 > 768 threads: the entries beyond the first pass)
      double sacc = 0.0;
      for (uint32_t ee = 1; ee < w.nb; ++ee) sacc += w.U[(size_t)ee * w.ustride + w.off_aa() + t2];
      w.x[(size_t)w.nb * m + nax + t2] = sacc;
    }
  }
}

// the arrow block: AA' = AA - sum over ALL blocks e of U_aa(e) (block 0 included: it was eliminated last), factored with the same
// register panel (rows: AA' padded to 4 NQA columns | rhs | identity), x_arrow = L^-T y.  The sum over the blocks runs over the
// lanes of a wave and ends in a fixed butterfly (reproducible); 16 entries per wave are in flight.
constexpr uint32_t CR_NA_MAX = 31;
template <int NQA>
__global__ __launch_bounds__(256) void k_cr_arrow(CrSys s, CrWs w) {
  constexpr uint32_t NPA = 4 * NQA, ldp = NPA + 1, NRG = 64;
  __shared__ double pan[(2 * NPA + 1) * (NPA + 1) + 2], cb[2 * (4 * NRG + NPA)];
  const uint32_t tid = threadIdx.x, ln = tid & 63u, NA = s.NA, nax = w.nax, nrows = NPA + 1 + NA;
  const uint32_t cp = __builtin_amdgcn_readfirstlane(tid >> 6), rg = ln;
  for (uint32_t i = tid; i < nrows * ldp; i += 256) { const uint32_t r = i / ldp, c = i - r * ldp; pan[i] = (r < NPA ? r == c : (r > NPA && r - NPA - 1 == c)) ? 1.0 : 0.0; }
  __syncthreads();
  const double* aasum = w.x + (size_t)w.nb * w.m + nax;   // blocks 1..nb-1 (k_cr_products, last launch); block 0's share is in U
  for (uint32_t q = tid; q < nax * nax; q += 256) {
    const uint32_t a = q / nax, b = q - a * nax;
    if (b > a || b >= NA) continue;
    const double val = (s.Sarrow[(size_t)a * s.ld + 6 * s.F + b] - aasum[q]) - w.U[w.off_aa() + q];
    if (a < NA) { pan[a * ldp + b] = val; pan[b * ldp + a] = val; } else pan[NPA * ldp + b] = val;   // a == NA: the rhs row
  }
  __syncthreads();
  double x[4][NQA];
  cr_panel_load<NQA, 4>(x, pan, ldp, rg, cp, NRG, nrows);
  const bool fail = cr_panel_factor<NQA, 4>(x, rg, cp, NRG, cb);
  cr_panel_store<NQA, 4>(x, pan, ldp, rg, cp, NRG, nrows);
  if (fail) *s.fail = 1.0;
  __syncthreads();
  // x = L^-T y: row NPA holds y, rows NPA + 1 + i hold row i of L^-T
  if (tid < NA) {
    double sacc = 0.0;
    for (uint32_t k = tid; k < NA; ++k) sacc += pan[(size_t)(NPA + 1 + tid) * ldp + k] * pan[(size_t)NPA * ldp + k];
    w.x[(size_t)w.nb * w.m + tid] = sacc; s.delta_red[6 * s.F + tid] = sacc;
  }
}

// x_e = L^-T (y_e - [P_l; P_r; P_a]^T [x_l; x_r; x_a]) for the blocks eliminated at `level`; 512 threads: 8 row groups x 64 columns
// (m <= 64), every load of a phase in flight at once; the rows of L^-T are fetched before the first phase needs its result
__global__ __launch_bounds__(512) void k_cr_backsub(CrSys s, CrWs w, uint32_t level) {
  __shared__ double v[160], part[8][64], tt[64];
  const uint32_t tid = threadIdx.x, m = w.m, NA = s.NA, nax = w.nax;
  const bool last = level >= w.levels;   // block 0: no neighbours, only the arrow part
  const uint32_t st = 1u << level, e = last ? 0u : (2 * blockIdx.x + 1) << level;
  const bool has_l = !last, has_r = !last && e + st < w.nb;
  const double* P = w.P + (size_t)e * w.prow * m;
  const uint32_t nv = 2 * m + NA, g = tid >> 6, j = tid & 63u;
  const bool jok = j < m;
  // row j of L^-T (upper triangular), entries k = j + g + 8 q
  double li[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { const uint32_t k = j + g + 8u * q; li[q] = (jok && k < m) ? P[(size_t)(2 * m + nax + j) * m + k] : 0.0; }
  // rows g + 8 q of [P_l; P_r; P_a[0..NA)], column j
  double pr[20];
#pragma unroll
  for (int q = 0; q < 20; ++q) { const uint32_t r = g + 8u * q; pr[q] = (jok && r < nv) ? P[(size_t)r * m + j] : 0.0; }
  const double yj = (tid < m) ? P[(size_t)(2 * m + NA) * m + tid] : 0.0;   // y_e: the rhs row of the panel
  if (tid < nv) {
    double q;
    if (tid < m) q = has_l ? w.x[(size_t)(e - st) * m + tid] : 0.0;
    else if (tid < 2 * m) q = has_r ? w.x[(size_t)(e + st) * m + (tid - m)] : 0.0;
    else q = w.x[(size_t)w.nb * m + (tid - 2 * m)];
    v[tid] = q;
  }
  __syncthreads();
  {
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 20; ++q) { const uint32_t r = g + 8u * q; if (r < nv) sacc += pr[q] * v[r]; }
    part[g][j] = sacc;
  }
  __syncthreads();
  if (tid < m) {
    double sacc = yj;
#pragma unroll
    for (int q = 0; q < 8; ++q) sacc -= part[q][tid];
    tt[tid] = sacc;
  }
  __syncthreads();
  {
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const uint32_t k = j + g + 8u * q; if (k < m) sacc += li[q] * tt[k]; }
    part[g][j] = sacc;
  }
  __syncthreads();
  if (tid < m) {
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sacc += part[q][tid];
    w.x[(size_t)e * m + tid] = sacc;
    const uint32_t f = e * s.bw + tid / 6;
    if (f < s.F) s.delta_red[6 * f + tid % 6] = sacc;
  }
}

// ---- host side: workspace, one-time kernel attributes, the launch sequence of one solve ----
struct CrPlan {
  CrWs ws{}; bool ready = false; int nq = 0;
  uint32_t fac_threads = 0; size_t fac_lds = 0, fin_lds = 0, prod_lds = 0;
};
inline size_t cr_ws_doubles_P(const CrWs& w) { return (size_t)w.nb * w.prow * w.m; }
inline size_t cr_ws_doubles_D(const CrWs& w) { return (size_t)w.nb * w.m * w.m; }
inline size_t cr_ws_doubles_A(const CrWs& w) { return (size_t)w.nb * w.nax * w.m; }
inline size_t cr_ws_doubles_U(const CrWs& w) { return (size_t)w.nb * w.ustride; }
inline size_t cr_ws_doubles_x(const CrWs& w) { return (size_t)w.nb * w.m + w.nax + (size_t)w.nax * w.nax; }   // solution | sum of the arrow products of the blocks 1..nb-1

#define CR_FOR_NQ(X) X(3) X(6) X(9) X(12) X(14) X(15)

// geometry + kernel attributes; the caller allocates ws.P / D / A / U / x (cr_ws_doubles_*) afterwards.  false: not eligible.
inline bool cr_plan(CrPlan& p, uint32_t F, uint32_t bw, uint32_t NA) {
  p.ready = false;
  if (!cr_eligible(F, bw, NA)) return false;
  p.ws = cr_geometry(F, bw, NA);
  const uint32_t m = p.ws.m;
  p.nq = cr_factor_nq(m);
  const uint32_t nrows = 4 * p.nq + p.ws.prow;
  p.fac_threads = cr_threads(nrows);
  if (const char* ev = getenv("LIFCAL_CR_THREADS")) { if (atoi(ev) == 256) p.fac_threads = 256; }   // A/B: one wave per SIMD, four rows per thread
  p.fac_lds = cr_factor_lds(nrows, p.nq); p.fin_lds = 0;
  p.prod_lds = cr_products_lds(p.ws);
  hipError_t rc = hipSuccess;
#define CR_ATTR(N) if (p.nq == N) rc = p.fac_threads == 256 ? hipFuncSetAttribute((const void*)k_cr_factor<N, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.fac_lds) \
                                                        : hipFuncSetAttribute((const void*)k_cr_factor<N, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.fac_lds);
  CR_FOR_NQ(CR_ATTR)
#undef CR_ATTR
  if (rc != hipSuccess) return false;
  if (hipFuncSetAttribute((const void*)k_cr_products, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.prod_lds) != hipSuccess) return false;
  p.ready = true;
  return true;
}

inline void cr_solve_launch(const CrPlan& p, const CrSys& s, hipStream_t stream) {
  const CrWs& w = p.ws;
  for (uint32_t l = 0; l < w.levels; ++l) {
    const uint32_t na = cr_n_active(w.nb, l), ne = na / 2, nk = (na + 1) / 2;
    if (!ne) continue;
#define CR_LAUNCH(N) if (p.nq == N) { if (p.fac_threads == 256) hipLaunchKernelGGL((k_cr_factor<N, 256>), dim3(ne), dim3(256), p.fac_lds, stream, s, w, l); \
                                      else hipLaunchKernelGGL((k_cr_factor<N, 512>), dim3(ne), dim3(512), p.fac_lds, stream, s, w, l); }
    CR_FOR_NQ(CR_LAUNCH)
#undef CR_LAUNCH
    hipLaunchKernelGGL(k_cr_products, dim3(nk + ne), dim3(CR_PROD_THREADS), p.prod_lds, stream, s, w, l);
  }
  {   // block 0 last: an elimination without neighbours, its share of the arrow block, the arrow block itself, x_0
    const uint32_t l = w.levels;
#define CR_LAUNCH(N) if (p.nq == N) { if (p.fac_threads == 256) hipLaunchKernelGGL((k_cr_factor<N, 256>), dim3(1), dim3(256), p.fac_lds, stream, s, w, l); \
                                      else hipLaunchKernelGGL((k_cr_factor<N, 512>), dim3(1), dim3(512), p.fac_lds, stream, s, w, l); }
    CR_FOR_NQ(CR_LAUNCH)
#undef CR_LAUNCH
    hipLaunchKernelGGL(k_cr_products, dim3(1), dim3(CR_PROD_THREADS), p.prod_lds, stream, s, w, l);
    if (s.NA <= 20) hipLaunchKernelGGL(k_cr_arrow<5>, dim3(1), dim3(256), 0, stream, s, w); else hipLaunchKernelGGL(k_cr_arrow<8>, dim3(1), dim3(256), 0, stream, s, w);
    hipLaunchKernelGGL(k_cr_backsub, dim3(1), dim3(512), 0, stream, s, w, l);
  }
  for (int l = (int)w.levels - 1; l >= 0; --l) {
    const uint32_t ne = cr_n_elim(w.nb, (uint32_t)l);
    if (ne) hipLaunchKernelGGL(k_cr_backsub, dim3(ne), dim3(512), 0, stream, s, w, (uint32_t)l);
  }
}

}  // namespace lifcal
