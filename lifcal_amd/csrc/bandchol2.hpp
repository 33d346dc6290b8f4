// bandchol2.hpp — the band + arrow Cholesky of bandchol.hpp as SEGMENT chains, and the twisted (two-ended) factorisation built
// from them (included by kernels.hpp after bandchol.hpp).
//
// The chain over the F pose blocks is sequential: k_band_chol_w walks 334 steps of ~5 k cycles at the metric point, 0.66 ms, and
// the back-substitution another 0.26 ms, on one workgroup.  A block-banded matrix can be eliminated from BOTH ends at once: frames
// 0..m-1 top-down and frames F-1..m+bw bottom-up never touch each other (their blocks are more than bw apart); each chain leaves
// its updates on the bw frames between them and on the arrow rows.  Three launches instead of one:
//   k_band_chol_seg  grid 2: chain A = frames 0..m-1 (forward), chain B = frames F-1..m+bw (backward: logical frame g = F-1-g, the
//                    stored lower blocks read transposed); each dumps its trailing window (the bw middle frames' blocks, their arrow
//                    entries, the arrow block — original minus its own updates)
//   k_band_merge     middle system = dump A + dump B - original, written over the middle frames' blocks of Sband / Sarrow
//   k_band_chol_seg  grid 1: the middle chain (bw frames) + the dense arrow factorisation
// and the back-substitution the other way round (arrow + middle, then both outer chains at once).  The chain length drops from F
// to (F - bw) / 2 + bw.  Every chain is the code of k_band_chol_w with a frame map; a single segment {+1, 0, F, F} IS the old
// kernel, which stays as the fallback for short sequences (F < 3 bw + 8) and for LIFCAL_TWISTED=0.
#pragma once

namespace lifcal {

struct BandSeg {
  int32_t dir;            // +1: logical frame g = physical base + g; -1: physical base - g
  uint32_t base;          // physical frame of logical frame 0
  uint32_t n_total;       // logical frames of the sub-problem: the n_elim eliminated ones, then the trailing ones
  uint32_t n_elim;        // frames this chain eliminates
  uint32_t final_arrow;   // 1: last chain of the factorisation (factors the arrow block; its back-substitution starts the solve)
  double* dump;           // trailing window of a partial chain (n_total > n_elim), see k_band_merge
};

// rows of column j's panel: logical frames j+1..j+nbel (6 each), then the NA+1 arrow rows
__global__ __launch_bounds__(256) void k_band_chol_seg(Dev d, double* Lpanel, BandSeg seg0, BandSeg seg1) {
  extern __shared__ __attribute__((aligned(16))) double bl[];
  const BandSeg sg = blockIdx.x == 0 ? seg0 : seg1;
  // logical frame g of this chain <-> physical frame; F = frames of the sub-problem (the eliminated ones, then the trailing ones)
  auto phys = [&](uint32_t g) -> uint32_t { return sg.dir > 0 ? sg.base + g : sg.base - g; };
  const uint32_t F = sg.n_total, FP = d.F, bw = d.bw, NAx = d.NA + 1, ld = d.ld, R = bw + 1;
  // element (a, b) of the block (row frame g, column frame g - dd) in the stored band: forward chains read it as stored; a backward
  // chain's column frame is the physically LARGER one, i.e. the stored block is the transpose (the diagonal block is symmetric)
  auto band_at = [&](uint32_t g, uint32_t dd, uint32_t a, uint32_t b) -> double {
    if (sg.dir > 0 || dd == 0) return d.Sband[((size_t)phys(g) * (bw + 1) + dd) * 36 + a * 6 + b];
    return d.Sband[((size_t)(phys(g) + dd) * (bw + 1) + dd) * 36 + b * 6 + a];
  };
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
      Wd[(size_t)(slot(f) + a) * nw + slot(f - dd) + b] = band_at(f, dd, a, b);
    }
    for (uint32_t t = lane; t < NAx * 6; t += 256) {
      const uint32_t a = t / 6, b = t % 6;
      Wd[(size_t)(arow0 + a) * nw + slot(f) + b] = d.Sarrow[(size_t)a * ld + 6 * phys(f) + b];
    }
  };
  for (uint32_t f = 0; f < min(R, F); ++f) load_frame_row(f);
  for (uint32_t t = lane; t < NAx * NAx; t += 256) {
    const uint32_t a = t / NAx, b = t % NAx;
    if (b <= a) Wd[(size_t)(arow0 + a) * nw + arow0 + b] = d.Sarrow[(size_t)a * ld + 6 * FP + b];
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
    for (int q = 0; q < 2; ++q) { const uint32_t t = lane + 256 * q; if (t < R * 36) pfn[q] = band_at(f, t / 36, (t % 36) / 6, t % 6); }
    if (lane < NAx * 6) pfn[2] = d.Sarrow[(size_t)(lane / 6) * ld + 6 * phys(f) + lane % 6];
  };
  if (pf_ok && R < F) fetch_frame(R);
  // ---- the chain over the pose blocks ----
  // (barriers inside the chain order LDS only: __syncthreads() would also wait for the panel / L^-1 stores on their way to
  // HBM — a write round trip per barrier, four per pose block — and nothing in the chain reads them back)
  for (uint32_t j = 0; j < sg.n_elim; ++j) {
    const uint32_t sj = slot(j);
    CSTAMP(5);
    if (j == 0 && lane == 192) factor_block(0, false);   // later blocks are factored by wave 3 inside the previous step's update
    CSTAMP(0);
    lds_barrier();
    CSTAMP(1);
    const uint32_t nbel = min(bw, F - 1 - j);
    const uint32_t nrows = 6 * nbel + NAx;
    double* Lp = Lpanel + (size_t)phys(j) * (6 * bw + NAx) * 6;
    if (lane >= 192 && lane < 228) d.Linv[(size_t)phys(j) * 36 + (lane - 192)] = Li[lane - 192];   // L_jj^-1 to HBM for the back-substitution: 36 lanes, off the factoring lane's path
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
    const bool ahead = nbel > 0 && j + 1 < sg.n_elim;   // (the first TRAILING frame's block is not factored here: it takes the plain update)
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
  if (!sg.final_arrow) {
    // ---- partial factorisation: the trailing frames' blocks, their arrow entries and the arrow block — original values minus
    // this chain's updates — go to the dump; k_band_merge combines the two chains' dumps into the middle system ----
    __syncthreads();
    const uint32_t nt = F - sg.n_elim;                       // trailing frames (= bw for a twisted factorisation)
    const uint32_t npair = nt * (nt + 1) / 2;
    for (uint32_t t = lane; t < npair * 36; t += 256) {
      const uint32_t pr = t / 36, e = t % 36, a = e / 6, b = e % 6;
      uint32_t ti = (uint32_t)((sqrtf(8.0f * (float)pr + 1.0f) - 1.0f) * 0.5f);
      while (ti * (ti + 1) / 2 > pr) --ti;
      while ((ti + 1) * (ti + 2) / 2 <= pr) ++ti;
      const uint32_t tj = pr - ti * (ti + 1) / 2;
      sg.dump[t] = (ti == tj && b > a) ? 0.0 : Wd[(size_t)(slot(sg.n_elim + ti) + a) * nw + slot(sg.n_elim + tj) + b];
    }
    double* da = sg.dump + (size_t)npair * 36;
    for (uint32_t t = lane; t < NAx * nt * 6; t += 256) {
      const uint32_t r = t / (nt * 6), ti = (t / 6) % nt, k = t % 6;
      da[t] = Wd[(size_t)(arow0 + r) * nw + slot(sg.n_elim + ti) + k];
    }
    double* daa = da + (size_t)NAx * nt * 6;
    for (uint32_t t = lane; t < NAx * NAx; t += 256) { const uint32_t a = t / NAx, b = t % NAx; daa[t] = b <= a ? Wd[(size_t)(arow0 + a) * nw + arow0 + b] : 0.0; }
    if (lane == 0 && *failp != 0.0) d.step[ST_CHOL_FAIL] = 1.0;   // (zeroed by k_tables; both chains and the final one may only raise it)
    return;
  }
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
    if (b <= a) d.Sarrow[(size_t)a * ld + 6 * FP + b] = Aa[(size_t)a * nw + b];
  }
  if (lane == 0 && *failp != 0.0) d.step[ST_CHOL_FAIL] = 1.0;
}

// L^T x = y with the packed panels: x_j = L_jj^-T (y_j - sum over the panel rows of column j)
// (segment form: xs is indexed by PHYSICAL frame; a chain first takes what it needs from the chains before it — the arrow part
// and its trailing frames — from delta_red, then walks its own columns backwards and writes them to delta_red)
__global__ __launch_bounds__(64) void k_band_backsolve_seg(Dev d, const double* Lpanel, BandSeg seg0, BandSeg seg1) {
  extern __shared__ __attribute__((aligned(16))) double xs[];   // n_red doubles
  const BandSeg sg = blockIdx.x == 0 ? seg0 : seg1;
  auto phys = [&](uint32_t g) -> uint32_t { return sg.dir > 0 ? sg.base + g : sg.base - g; };
  const uint32_t FP = d.F, F = sg.n_total, bw = d.bw, NA = d.NA, NAx = NA + 1, ld = d.ld, lane = threadIdx.x;
  const double* Aa = d.Sarrow + 6 * FP;
  if (sg.final_arrow) {
    // y: pose part was accumulated in the rhs arrow row of every panel (row index nrows-1 of column j's panel),
    // arrow part sits in the factored arrow block's last row
    for (uint32_t a = lane; a < NA; a += 64) xs[6 * FP + a] = Aa[(size_t)NA * ld + a];
    __syncthreads();
    for (int a = (int)NA - 1; a >= 0; --a) {   // arrow block: dense back-substitution
      if (lane == 0) xs[6 * FP + a] /= Aa[(size_t)a * ld + a];
      __syncthreads();
      const double xa = xs[6 * FP + a];
      for (uint32_t b = lane; b < (uint32_t)a; b += 64) xs[6 * FP + b] -= Aa[(size_t)a * ld + b] * xa;
      __syncthreads();
    }
    for (uint32_t a = lane; a < NA; a += 64) d.delta_red[6 * FP + a] = xs[6 * FP + a];
  } else {
    for (uint32_t a = lane; a < NA; a += 64) xs[6 * FP + a] = d.delta_red[6 * FP + a];
    for (uint32_t t = lane; t < 6 * (F - sg.n_elim); t += 64) { const uint32_t p6 = 6 * phys(sg.n_elim + t / 6) + t % 6; xs[p6] = d.delta_red[p6]; }
    __syncthreads();
  }
  // One wave walks the chain backwards.  The packed panel of the NEXT column (two rows per lane + the y_j entry of lanes 0-5)
  // is requested before the current column is reduced, the reduction runs on DPP (no LDS crossbar shuffles), and the
  // rhs row is read where it is needed instead of being summed over the wave.
  const uint32_t prow = 6 * bw + NAx;   // panel rows reserved per column
  double nx[2][6], ny = 0.0, nli[6];   // nli: column `lane` of L_jj^-1 (lanes 0-5), fetched with the panel
  auto fetch = [&](int jj) {
    const uint32_t nb = min(bw, F - 1 - (uint32_t)jj), nr = 6 * nb + NAx;
    const double* Lp = Lpanel + (size_t)phys((uint32_t)jj) * prow * 6;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = lane + 64 * q;
      const double* row = Lp + (size_t)(r < nr - 1 ? r : 0) * 6;   // clamped: always a valid address, masked at use
#pragma unroll
      for (int k = 0; k < 6; ++k) nx[q][k] = row[k];
    }
    ny = Lp[(size_t)(nr - 1) * 6 + (lane < 6 ? lane : 0)];          // y_j: the rhs row of the panel (forward substitution done by the factorisation)
#pragma unroll
    for (int k = 0; k < 6; ++k) nli[k] = d.Linv[(size_t)phys((uint32_t)jj) * 36 + k * 6 + (lane < 6 ? lane : 0)];
  };
  if (sg.n_elim > 0) fetch((int)sg.n_elim - 1);
  for (int j = (int)sg.n_elim - 1; j >= 0; --j) {
    const uint32_t nbel = min(bw, F - 1 - (uint32_t)j);
    const uint32_t nrows = 6 * nbel + NAx;
    const double* Lp = Lpanel + (size_t)phys((uint32_t)j) * prow * 6;
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
        const double xv = (r < 6 * nbel) ? xs[6 * phys((uint32_t)j + 1 + r / 6) + r % 6] : xs[6 * FP + (r - 6 * nbel)];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += cur[q][k] * xv;
      }
    }
    for (uint32_t r = lane + 128; r < nrows - 1; r += 64) {   // wider panels: the rows beyond the prefetched two per lane
      const double* row = Lp + (size_t)r * 6;
      const double xv = (r < 6 * nbel) ? xs[6 * phys((uint32_t)j + 1 + r / 6) + r % 6] : xs[6 * FP + (r - 6 * nbel)];
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
      xs[6 * phys((uint32_t)j) + lane] = o;
    }
    lds_barrier();   // LDS ordering only (a two-steps-ahead prefetch was measured: no gain, the step is instruction-bound on one wave)
  }
  for (uint32_t t = lane; t < 6 * sg.n_elim; t += 64) { const uint32_t p6 = 6 * phys(t / 6) + t % 6; d.delta_red[p6] = xs[p6]; }
}


// middle system of a twisted factorisation: S_mid = dumpA + dumpB - original, over the blocks among the middle frames
// [m, m + nt), their arrow entries and the arrow block (rhs row included).  Chain A's trailing frame t is the physical frame
// m + t; chain B runs backwards, its trailing frame t is the physical frame m + nt - 1 - t and its blocks are transposed.
__global__ void k_band_merge(Dev d, uint32_t m, uint32_t nt, const double* dumpA, const double* dumpB) {
  const uint32_t NAx = d.NA + 1, bw = d.bw, ld = d.ld, F = d.F;
  const uint32_t npair = nt * (nt + 1) / 2;
  const uint32_t n1 = npair * 36, n2 = n1 + NAx * nt * 6, n3 = n2 + NAx * NAx;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n3) return;
  auto pair = [](uint32_t ti, uint32_t tj) { return ti * (ti + 1) / 2 + tj; };   // ti >= tj
  if (t < n1) {
    const uint32_t pr = t / 36, e = t % 36, a = e / 6, b = e % 6;
    uint32_t ti = (uint32_t)((sqrtf(8.0f * (float)pr + 1.0f) - 1.0f) * 0.5f);
    while (ti * (ti + 1) / 2 > pr) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= pr) ++ti;
    const uint32_t tj = pr - ti * (ti + 1) / 2;                 // physical frames fi = m + ti >= fj = m + tj
    if (ti == tj && b > a) return;
    double* dst = d.Sband + ((size_t)(m + ti) * (bw + 1) + (ti - tj)) * 36 + a * 6 + b;
    const double va = dumpA[(size_t)pair(ti, tj) * 36 + a * 6 + b];
    // chain B: frame fj is its trailing frame nt-1-tj >= nt-1-ti; its block (rows of fj, columns of fi) holds S[fj, fi] = S[fi, fj]^T
    const uint32_t ui = nt - 1 - tj, uj = nt - 1 - ti;
    const double vb = (ti == tj) ? dumpB[(size_t)pair(ui, uj) * 36 + a * 6 + b] : dumpB[(size_t)pair(ui, uj) * 36 + b * 6 + a];
    *dst = va + vb - *dst;
  } else if (t < n2) {
    const uint32_t q = t - n1, r = q / (nt * 6), ti = (q / 6) % nt, k = q % 6;
    double* dst = d.Sarrow + (size_t)r * ld + 6 * (m + ti) + k;
    const double* da = dumpA + (size_t)npair * 36; const double* db = dumpB + (size_t)npair * 36;
    *dst = da[((size_t)r * nt + ti) * 6 + k] + db[((size_t)r * nt + (nt - 1 - ti)) * 6 + k] - *dst;
  } else {
    const uint32_t q = t - n2, a = q / NAx, b = q % NAx;
    if (b > a) return;
    double* dst = d.Sarrow + (size_t)a * ld + 6 * F + b;
    const size_t off = (size_t)npair * 36 + (size_t)NAx * nt * 6 + q;
    *dst = dumpA[off] + dumpB[off] - *dst;
  }
}

}  // namespace lifcal
