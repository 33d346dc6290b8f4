// bandchol.hpp — latency-oriented block-banded + arrow Cholesky of the reduced system (included by kernels.hpp).
//
// Replaces ceres DenseSchurComplementSolver's Eigen LLT on the dense (17+6F)^2 matrix (reference call site
// src/CameraCalibration.cpp:956, DENSE_SCHUR) by a factorisation that only touches the co-visibility band:
// O(F bw^2 6^3) instead of n^3/3.  The chain over the F pose blocks is inherently sequential, so ONE wave walks it
// and everything it touches per step sits in LDS:
//   window Wd (n_w x n_w, n_w = 6 (bw+1) + NA + 1): the frames j..j+bw as a ring (slot = frame mod (bw+1)) followed by
//   the arrow rows (promoted points | camera | rhs).  Step j: factor the 6x6 diagonal block, scale the panel rows,
//   rank-6 update of the window, slide (frame j+bw+1 enters from HBM).  Carrying the rhs as an arrow row makes the
//   forward substitution part of the factorisation.
// Each column's panel (rows below the diagonal block, 6 doubles each) is also written contiguously to Lpanel so
// that the back-substitution streams it with coalesced loads.  Systems whose window does not fit LDS use the older
// global-memory kernels (k_band_chol / k_band_backsolve).
#pragma once

namespace lifcal {

struct BandLds {
  uint32_t nw, nr4, off_pn, off_d, off_map, off_dummy, total;   // window | panel, component-major (6 x nr4) | L_jj, L_jj^-1, flag | window row of each panel row
  __host__ __device__ BandLds(uint32_t bw, uint32_t NA) {
    nw = (6 * (bw + 1) + NA + 1) | 1u;   // window dimension AND row stride: odd, so that consecutive rows start on different LDS banks
    nr4 = (nw + 3u) & ~3u;
    off_pn = (nw * nw + 1u) & ~1u; off_d = off_pn + nr4 * 6; off_map = off_d + 80; off_dummy = off_map + (nr4 + 1) / 2 + 1; total = off_dummy + 256;   // ... | one scratch double per thread
  }
};

// rows of column j's panel: frames j+1..j+nbel (6 each), then the NA+1 arrow rows
__global__ __launch_bounds__(256) void k_band_chol_w(Dev d, double* Lpanel) {
  extern __shared__ __attribute__((aligned(16))) double bl[];
  const uint32_t F = d.F, bw = d.bw, NAx = d.NA + 1, ld = d.ld, R = bw + 1;
  const BandLds lay(bw, d.NA);
  const uint32_t nw = lay.nw, arow0 = 6 * R, NR = lay.nr4;
  double* Wd = bl; double* Pn = bl + lay.off_pn; double* Ld = bl + lay.off_d; double* Li = Ld + 36; double* failp = Ld + 72;
  const uint32_t lane = threadIdx.x;   // 256 threads
  uint32_t* wmap = (uint32_t*)(bl + lay.off_map);
  auto slot = [&](uint32_t f) { return 6 * (f % R); };
  if (lane == 0) *failp = 0.0;
  // ---- load the initial window: frames 0..min(bw, F-1), all arrow rows ----
  for (uint32_t i = lane; i < nw * nw; i += 256) Wd[i] = 0.0;
  __syncthreads();
  auto load_frame_row = [&](uint32_t f) {   // blocks (f, f-dd), dd = 0..min(bw, f), and the arrow entries of column f
    const uint32_t ndd = min(bw, f) + 1;
    for (uint32_t t = lane; t < ndd * 36; t += 256) {
      const uint32_t dd = t / 36, e = t % 36, a = e / 6, b = e % 6;
      if (dd == 0 && b > a) continue;
      Wd[(size_t)(slot(f) + a) * nw + slot(f - dd) + b] = d.Sband[((size_t)f * (bw + 1) + dd) * 36 + e];
    }
    for (uint32_t t = lane; t < NAx * 6; t += 256) {
      const uint32_t a = t / 6, b = t % 6;
      Wd[(size_t)(arow0 + a) * nw + slot(f) + b] = d.Sarrow[(size_t)a * ld + 6 * f + b];
    }
  };
  for (uint32_t f = 0; f < min(R, F); ++f) load_frame_row(f);
  for (uint32_t t = lane; t < NAx * NAx; t += 256) {
    const uint32_t a = t / NAx, b = t % NAx;
    if (b <= a) Wd[(size_t)(arow0 + a) * nw + arow0 + b] = d.Sarrow[(size_t)a * ld + 6 * F + b];
  }
  __syncthreads();
  // 6x6 Cholesky of pose block jf and the inverse of its factor: ONE lane, a pure dependency chain
  auto factor_block = [&](uint32_t jf, bool subtract_panel) {
    const uint32_t sf = slot(jf);
    // One lane, a pure dependency chain: reciprocal square roots only (v_rsq_f64 + Newton steps) — the sqrt + divide
    // pairs of the textbook form were most of the time of a chain step.  ir[c] = 1 / L[c][c].
    double L[6][6], ir[6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) L[a][b] = Wd[(size_t)(sf + a) * nw + sf + b];
    if (subtract_panel) {   // last contribution to this block: the first six rows of the current column's panel (all loads first)
      double P6[6][6];
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int a = 0; a < 6; ++a) P6[k][a] = Pn[(size_t)k * NR + a];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
#pragma unroll
          for (int k = 0; k < 6; ++k) L[a][b] -= P6[k][a] * P6[k][b];
        }
    }
    bool ok = true;
#pragma unroll
    for (int cI = 0; cI < 6; ++cI) {
      double dg = L[cI][cI];
#pragma unroll
      for (int k = 0; k < cI; ++k) dg -= L[cI][k] * L[cI][k];
      if (!(dg > 0.0)) { ok = false; dg = 1.0; }
      const double idg = rsqrt(dg);
      ir[cI] = idg; L[cI][cI] = dg * idg;
#pragma unroll
      for (int r = cI + 1; r < 6; ++r) { double s = L[r][cI];
#pragma unroll
        for (int k = 0; k < cI; ++k) s -= L[r][k] * L[cI][k];
        L[r][cI] = s * idg; }
    }
    if (!ok) *failp = 1.0;
    double I[6][6];
#pragma unroll
    for (int cI = 0; cI < 6; ++cI) {
#pragma unroll
      for (int r = 0; r < 6; ++r) I[r][cI] = 0.0;
      I[cI][cI] = ir[cI];
#pragma unroll
      for (int r = cI + 1; r < 6; ++r) { double s = 0.0;
#pragma unroll
        for (int k = cI; k < r; ++k) s -= L[r][k] * I[k][cI];
        I[r][cI] = s * ir[r]; }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) { Li[a * 6 + b] = (b <= a) ? I[a][b] : 0.0; }
  };
#ifdef LIFCAL_STAMPS
  unsigned long long cst[6] = {0, 0, 0, 0, 0, 0}, clast = 0;
  if (lane == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clast) :: "memory");
#define CSTAMP(i) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); cst[i] += t_ - clast; clast = t_; } } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
  auto tri_block = [](uint32_t t, uint32_t& bi, uint32_t& bj) {
    bi = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (bi * (bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    bj = t - bi * (bi + 1) / 2;
  };
  uint32_t bi0, bj0;
  tri_block(lane, bi0, bj0);
  const bool pf_ok = (R * 36 <= 512) && (NAx * 6 <= 256);
  double pfn[3] = {0.0, 0.0, 0.0};
  auto fetch_frame = [&](uint32_t f) {
#pragma unroll
    for (int q = 0; q < 2; ++q) { const uint32_t t = lane + 256 * q; if (t < R * 36) pfn[q] = d.Sband[((size_t)f * (bw + 1) + t / 36) * 36 + t % 36]; }
    if (lane < NAx * 6) pfn[2] = d.Sarrow[(size_t)(lane / 6) * ld + 6 * f + lane % 6];
  };
  if (pf_ok && R < F) fetch_frame(R);
  // ---- the chain over the pose blocks ----
  // (barriers inside the chain order LDS only: __syncthreads() would also wait for the panel / L^-1 stores on their way to
  // HBM — a write round trip per barrier, four per pose block — and nothing in the chain reads them back)
  for (uint32_t j = 0; j < F; ++j) {
    const uint32_t sj = slot(j);
    CSTAMP(5);
    if (j == 0 && lane == 192) factor_block(0, false);   // later blocks are factored by wave 3 inside the previous step's update
    CSTAMP(0);
    lds_barrier();
    CSTAMP(1);
    const uint32_t nbel = min(bw, F - 1 - j);
    const uint32_t nrows = 6 * nbel + NAx;
    double* Lp = Lpanel + (size_t)j * (6 * bw + NAx) * 6;
    if (lane >= 192 && lane < 228) d.Linv[(size_t)j * 36 + (lane - 192)] = Li[lane - 192];   // L_jj^-1 to HBM for the back-substitution: 36 lanes, off the factoring lane's path
    for (uint32_t r = lane; r < nrows; r += 256) {
      const uint32_t wrow = (r < 6 * nbel) ? slot(j + 1 + r / 6) + r % 6 : arow0 + (r - 6 * nbel);
      wmap[r] = wrow;
      const double* src = Wd + (size_t)wrow * nw + sj;
      double x[6], y[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = src[k];
#pragma unroll
      for (int cI = 0; cI < 6; ++cI) { double s = 0.0;
#pragma unroll
        for (int k = 0; k <= cI; ++k) s += x[k] * Li[cI * 6 + k];
        y[cI] = s; }
#pragma unroll
      for (int k = 0; k < 6; ++k) { Pn[(size_t)k * NR + r] = y[k]; Lp[(size_t)r * 6 + k] = y[k]; }
    }
    CSTAMP(2);
    lds_barrier();
    // the frame that enters the ring after this step was requested from HBM ONE STEP AGO (pfn); the request for the frame
    // of the next step goes out now — a step is shorter than the HBM round trip
    // (up to 2 band values + 1 arrow value per thread for bw <= 13; wider bands take the plain path below)
    const bool has_next = (j + R < F);
    const bool pf = has_next && pf_ok;
    double pfv[3] = {pfn[0], pfn[1], pfn[2]};
    if (pf_ok && j + 1 + R < F) fetch_frame(j + 1 + R);
    // rank-6 update of the window in 4x4 blocks of (panel row, panel row) pairs over the lower triangle, one block per
    // thread.  The phase is bound by LDS traffic: the panel is stored component-major (Pn[k][row]) so that the four rows of
    // a block are one 32-byte run per component (12 + 12 ds_read_b128 for 96 MACs, no bank-conflicting 48-byte strides),
    // 5.5 LDS operations per pair instead of 9 with a row per thread, and every thread has the same amount of work.
    // The diagonal block of frame j+1 receives its last contribution from this column: lane 192 (wave 3, idle in the
    // blocked update below for the usual band widths) applies it first and factors the block right away, so that the
    // single-lane factorisation of step j+1 runs UNDER this step's update instead of in front of the next one.
    const bool ahead = nbel > 0;
    if (ahead && lane == 192) factor_block(j + 1, true);   // (the block itself is not written back: nothing reads it after its factorisation)
    {
      const uint32_t nb4 = (nrows + 3u) >> 2, nblk = nb4 * (nb4 + 1) / 2;
      const uint32_t first = lane < 192 ? lane : lane - 192 + 192;   // (all four waves take blocks; lane 192 joins after its factorisation)
      for (uint32_t t = first; t < nblk; t += 256) {
        uint32_t bi = bi0, bj = bj0;   // block of t = lane, decoded once before the chain (a shorter panel uses a prefix of the blocks)
        if (t != lane) tri_block(t, bi, bj);
        double2 pr[6][2], pc[6][2];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          pr[k][0] = *reinterpret_cast<const double2*>(Pn + (size_t)k * NR + 4 * bi); pr[k][1] = *reinterpret_cast<const double2*>(Pn + (size_t)k * NR + 4 * bi + 2);
          pc[k][0] = *reinterpret_cast<const double2*>(Pn + (size_t)k * NR + 4 * bj); pc[k][1] = *reinterpret_cast<const double2*>(Pn + (size_t)k * NR + 4 * bj + 2);
        }
        uint32_t wr[4], wc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { wr[i] = wmap[min(4 * bi + i, nrows - 1)] * nw; wc[i] = wmap[min(4 * bj + i, nrows - 1)]; }
        // branch-free: entries that are not this block's to update (upper triangle of a diagonal block, rows past the panel,
        // the six rows lane 192 takes) are pointed at the thread's scratch double; all reads come before all writes (written
        // one by one the compiler has to assume that the entries alias and pays an LDS round trip per entry)
        const uint32_t scratch = lay.off_dummy + lane;
        uint32_t wa[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) {
            const uint32_t r = 4 * bi + i, cI = 4 * bj + jx;
            wa[i][jx] = (r < nrows && cI <= r && !(ahead && r < 6)) ? wr[i] + wc[jx] : scratch;
          }
        double oldv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) oldv[i][jx] = bl[wa[i][jx]];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) sacc += ((i & 1) ? pr[k][i >> 1].y : pr[k][i >> 1].x) * ((jx & 1) ? pc[k][jx >> 1].y : pc[k][jx >> 1].x);
            bl[wa[i][jx]] = oldv[i][jx] - sacc;
          }
      }
    }
    // slide: frame j leaves its slot, frame j + bw + 1 (if any) enters it — in the SAME phase as the update: the update
    // touches rows and columns of the frames j+1..j+bw and of the arrow only, the incoming frame's row and column live in
    // the slot frame j has just vacated (its column was last read by the panel phase, a barrier ago).  No zeroing is
    // needed: every entry of the slot's row that is read later is overwritten here (all bw+1 blocks of the incoming frame),
    // and stale entries of the slot's column are overwritten when the rows that use them enter.
    if (has_next) {
      const uint32_t f = j + R;
      if (pf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t t = lane + 256 * q;
          if (t < R * 36) { const uint32_t dd = t / 36, e = t % 36, a = e / 6, b2 = e % 6; if (!(dd == 0 && b2 > a)) Wd[(size_t)(slot(f) + a) * nw + slot(f - dd) + b2] = pfv[q]; }
        }
        if (lane < NAx * 6) Wd[(size_t)(arow0 + lane / 6) * nw + slot(f) + lane % 6] = pfv[2];
      } else {
        load_frame_row(f);
      }
    }
    CSTAMP(3);
    lds_barrier();
    CSTAMP(4);
  }
#ifdef LIFCAL_STAMPS
  if (lane == 0 && d.dbg) for (int i = 0; i < 6; ++i) d.dbg[i] = cst[i];
#endif
  // ---- dense Cholesky of the arrow block (NA x NA), rhs row carried along ----
  double* Aa = Wd + (size_t)arow0 * nw + arow0;   // Aa[a * nw + b]
  for (uint32_t cI = 0; cI < d.NA; ++cI) {
    if (lane == 0) { double dg = Aa[(size_t)cI * nw + cI]; if (!(dg > 0.0)) { *failp = 1.0; dg = 1.0; } Aa[(size_t)cI * nw + cI] = sqrt(dg); }
    __syncthreads();
    const double dg = Aa[(size_t)cI * nw + cI];
    for (uint32_t r = cI + 1 + lane; r < NAx; r += 256) Aa[(size_t)r * nw + cI] /= dg;
    __syncthreads();
    const uint32_t m = NAx - cI - 1;
    for (uint32_t t = lane; t < m * (m + 1) / 2; t += 256) {
      uint32_t a = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > t) --a;
      while ((a + 1) * (a + 2) / 2 <= t) ++a;
      const uint32_t b = t - a * (a + 1) / 2;
      Aa[(size_t)(cI + 1 + a) * nw + cI + 1 + b] -= Aa[(size_t)(cI + 1 + a) * nw + cI] * Aa[(size_t)(cI + 1 + b) * nw + cI];
    }
    __syncthreads();
  }
  // the factor of the arrow block and y = L^-1 rhs go back to Sarrow (the back-substitution reads them there)
  for (uint32_t t = lane; t < NAx * NAx; t += 256) {
    const uint32_t a = t / NAx, b = t % NAx;
    if (b <= a) d.Sarrow[(size_t)a * ld + 6 * F + b] = Aa[(size_t)a * nw + b];
  }
  if (lane == 0) d.step[ST_CHOL_FAIL] = *failp;
}

// L^T x = y with the packed panels: x_j = L_jj^-T (y_j - sum over the panel rows of column j)
__global__ __launch_bounds__(64) void k_band_backsolve_w(Dev d, const double* Lpanel) {
  extern __shared__ __attribute__((aligned(16))) double xs[];   // n_red doubles
  const uint32_t F = d.F, bw = d.bw, NA = d.NA, NAx = NA + 1, ld = d.ld, lane = threadIdx.x;
  const double* Aa = d.Sarrow + 6 * F;
  // y: pose part was accumulated in the rhs arrow row of every panel (row index nrows-1 of column j's panel),
  // arrow part sits in the factored arrow block's last row
  for (uint32_t a = lane; a < NA; a += 64) xs[6 * F + a] = Aa[(size_t)NA * ld + a];
  __syncthreads();
  for (int a = (int)NA - 1; a >= 0; --a) {   // arrow block: dense back-substitution
    if (lane == 0) xs[6 * F + a] /= Aa[(size_t)a * ld + a];
    __syncthreads();
    const double xa = xs[6 * F + a];
    for (uint32_t b = lane; b < (uint32_t)a; b += 64) xs[6 * F + b] -= Aa[(size_t)a * ld + b] * xa;
    __syncthreads();
  }
  // One wave walks the chain backwards.  The packed panel of the NEXT column (two rows per lane + the y_j entry of lanes 0-5)
  // is requested before the current column is reduced, the reduction runs on DPP (no LDS crossbar shuffles), and the
  // rhs row is read where it is needed instead of being summed over the wave.
  const uint32_t prow = 6 * bw + NAx;   // panel rows reserved per column
  double nx[2][6], ny = 0.0, nli[6];   // nli: column `lane` of L_jj^-1 (lanes 0-5), fetched with the panel
  auto fetch = [&](int jj) {
    const uint32_t nb = min(bw, F - 1 - (uint32_t)jj), nr = 6 * nb + NAx;
    const double* Lp = Lpanel + (size_t)jj * prow * 6;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = lane + 64 * q;
      const double* row = Lp + (size_t)(r < nr - 1 ? r : 0) * 6;   // clamped: always a valid address, masked at use
#pragma unroll
      for (int k = 0; k < 6; ++k) nx[q][k] = row[k];
    }
    ny = Lp[(size_t)(nr - 1) * 6 + (lane < 6 ? lane : 0)];          // y_j: the rhs row of the panel (forward substitution done by the factorisation)
#pragma unroll
    for (int k = 0; k < 6; ++k) nli[k] = d.Linv[(size_t)jj * 36 + k * 6 + (lane < 6 ? lane : 0)];
  };
  if (F > 0) fetch((int)F - 1);
  for (int j = (int)F - 1; j >= 0; --j) {
    const uint32_t nbel = min(bw, F - 1 - (uint32_t)j);
    const uint32_t nrows = 6 * nbel + NAx;
    const double* Lp = Lpanel + (size_t)j * prow * 6;
    double cur[2][6];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int k = 0; k < 6; ++k) cur[q][k] = nx[q][k];
    const double yj = ny;
    double li[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) li[k] = nli[k];
    if (j > 0) fetch(j - 1);
    double acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = lane + 64 * q;
      if (r < nrows - 1) {
        const double xv = (r < 6 * nbel) ? xs[6 * ((uint32_t)j + 1 + r / 6) + r % 6] : xs[6 * F + (r - 6 * nbel)];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += cur[q][k] * xv;
      }
    }
    for (uint32_t r = lane + 128; r < nrows - 1; r += 64) {   // wider panels: the rows beyond the prefetched two per lane
      const double* row = Lp + (size_t)r * 6;
      const double xv = (r < 6 * nbel) ? xs[6 * ((uint32_t)j + 1 + r / 6) + r % 6] : xs[6 * F + (r - 6 * nbel)];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += row[k] * xv;
    }
    double tot[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double sdpp = wave_sum_dpp(acc[k]);   // total in lane 63
      tot[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sdpp), 63), __builtin_amdgcn_readlane(__double2loint(sdpp), 63));
    }
    // t = y_j - acc (lane k holds t_k), x_j = L_jj^-T t: (L^-T t)[lane] = sum_{k >= lane} Li[k][lane] t[k]
    double tk = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) if (lane == (uint32_t)k) tk = yj - tot[k];
    if (lane < 6) {
      double o = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tk), k), __builtin_amdgcn_readlane(__double2loint(tk), k));
        if (k >= (int)lane) o += li[k] * t;
      }
      xs[6 * j + lane] = o;
    }
    lds_barrier();   // LDS ordering only (a two-steps-ahead prefetch was measured: no gain, the step is instruction-bound on one wave)
  }
  for (uint32_t k = lane; k < d.n_red; k += 64) d.delta_red[k] = xs[k];
}

}  // namespace lifcal
