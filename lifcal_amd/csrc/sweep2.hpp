// sweep2.hpp — k_sweep2: the fused Jacobian + Schur sweep for "regular" points (included by kernels.hpp).
//
// One WORKGROUP owns a contiguous range of points ("block") whose frames fit a window of nf <= 20 frames.
// Everything the block accumulates lives in LDS until one flush at the end:
//   Spp  pose x pose blocks of the window (lower block triangle, 36 doubles each)
//   Scp  camera x pose, Scc camera x camera, vec = gradient / diagonal / rhs of the window + camera
//   slab U(6) g(3) per point of the current pass, Zd = dense (3 np) x (window columns) matrix of the pass
// A pass = <= 256 groups (group g -> wave g%4, lane g/4).  Per pass:
//   (1) lanes walk their observations (residual, analytic Jacobian, Cauchy weight) and add their rotated blocks with
//       LDS f64 atomics; W_pose goes into its cell of Zd, the lane's 3x3 block A to HBM (back-substitution rebuilds W from it),
//   (2) one thread per point damps and factors U = L L^T (ceres LevenbergMarquardtStrategy + InvertPSDMatrix<3>),
//   (3) Zd <- L^-1 W row-wise; rhs column = L^-1 g,
//   (4) window -= Zd^T Zd as a register-blocked fp64 product (4x4 micro-tiles per lane; on gfx950 v_fmac_f64 beats
//       v_mfma_f64_16x16x4_f64, measured in tools/ubench): every output entry has one owner, so the point elimination
//       W^T U^-1 W (and W^T U^-1 g, the last column) needs no atomics at all.
// mode 1 ("diagonal only", iteration 0): just the Hessian diagonal for the Jacobi scaling.
// Replaces, per LM iteration: ceres autodiff evaluation of OurCostFunctionBundle (reference
// src/BundleAdjustment/BundleAdjustment.h:120-222) + SchurEliminator::Eliminate (out of tree).
#pragma once

namespace lifcal {

constexpr uint32_t ZD_DOUBLES = 8192;   // LDS doubles reserved for the dense Z matrix of one pass (64 KiB)

constexpr uint32_t FRV = 27 + 6 * NCMAX;   // frame-level values a group emits: pose x pose lower (21) + pose gradient (6) + camera x pose (6 NC)
constexpr uint32_t LDS_LIMIT_DOUBLES = 160 * 1024 / 8;
constexpr uint32_t MISC_DOUBLES = 10;   // misc block of the LDS window: [0] cost [1] bad-U count [2] max |g_p| bits [3] two u32 counters [4..7] eight u32 hand-off counters [8..9] four u32 (deterministic emission turns)

// doubles per point of the per-pass point slab (U 6 | g 3, padded): an ODD stride — with 12 the slabs of points 8 apart start on the
// same LDS banks and the nine f64 atomics a lane adds to its point's slab ran into 8-way bank conflicts
constexpr uint32_t SLAB_STRIDE = 13;
struct V2Lds {
  uint32_t nfm, nrep, np_max, zd, off_cp, off_cc, off_vec, off_fr, off_bt, off_slab, off_zd, off_misc, total;
  // LDS doubles of the dense Z matrix of one pass (and of the hand-off buffer that lives there during the observation loop)
  __host__ __device__ static constexpr uint32_t zd_for(uint32_t pass_lanes) { return pass_lanes >= 256 ? ZD_DOUBLES : ZD_DOUBLES / 2; }
  // single_replica: k_sweep3 (lanes sorted by frame, frame-level values pre-summed over runs of lanes): one set of frame
  // accumulators; the LDS this frees holds the accumulator threads' partial Schur tiles (16 doubles x 256 threads) when they fit.
  // pass_lanes: 256 (k_sweep2, k_sweep3 with four waves per role) or 128 (k_sweep3 with two waves per role: half the Z matrix,
  // half the per-pass point slab, no partial tiles — the window of a 12-frame block then takes ~75 KiB, two workgroups per CU)
  __host__ __device__ explicit V2Lds(uint32_t nfmax, bool single_replica = false, uint32_t pass_lanes = 256) {
    nfm = nfmax;
    zd = zd_for(pass_lanes); np_max = pass_lanes / 4;
    const uint32_t npp = nfm * (nfm + 1) / 2;
    off_cp = npp * 36; off_cc = off_cp + NCMAX * 6 * nfm; off_vec = off_cc + 48;
    off_fr = off_vec + 3 * (6 * nfm + NCMAX + 3);
    off_fr = (off_fr + 1) & ~1u;
    // replicated frame accumulators [value][rep][frame]: as many replicas (<= 8) as the 160 KiB LDS allows
    const uint32_t fixed = off_fr + np_max * SLAB_STRIDE + zd + MISC_DOUBLES + 32 + 64;
    uint32_t r = (LDS_LIMIT_DOUBLES - fixed) / (FRV * nfm);
    nrep = single_replica ? 1u : (r < 1 ? 1 : (r > 8 ? 8 : r));
    off_bt = off_fr + nrep * FRV * nfm;
    off_bt = (off_bt + 1) & ~1u;
    const bool bt = single_replica && pass_lanes >= 256 && (off_bt + 16 * 256 + np_max * SLAB_STRIDE + zd + MISC_DOUBLES + 32 + 64 <= LDS_LIMIT_DOUBLES);
    off_slab = off_bt + (bt ? 16 * 256 : 0);
    off_zd = off_slab + np_max * SLAB_STRIDE; off_misc = off_zd + zd; total = off_misc + MISC_DOUBLES + 32 + 64;   // misc | point ids (64 u32) | column info (<= 256 u16)
  }
  __host__ __device__ bool has_bt() const { return off_slab != off_bt; }
};

typedef double v4f64 __attribute__((ext_vector_type(4)));

// wave64 sum through DPP (row_shr 1,2,4,8 + row_bcast 15/31); the total ends up in lane 63
template <int CTRL, int ROW_MASK>
LIFCAL_DEV double dpp_add_step(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi2, lo2);
}
LIFCAL_DEV double wave_sum_dpp(double v) {
  v = dpp_add_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
  v = dpp_add_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return v;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for every global load and
// store in flight (prefetched descriptors, the W rows on their way to HBM); nothing in k_sweep2 reads back what another wave
// wrote to global memory, so only the LDS counter has to reach zero before the barrier
LIFCAL_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef LIFCAL_STAMPS
#define STAMP(i) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[i] += t_ - st_last; st_last = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int NR, bool TAN, bool ADJ>
__global__ __launch_bounds__(256) void k_sweep2(Dev d, double radius, int mode) {
  constexpr int NC = 5 + NR + (TAN ? 2 : 0);
  constexpr int NCC = NC * (NC + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const V2Lds lay(d.v2_nfmax);
  const uint32_t NFm = lay.nfm, vlen = 6 * NFm + NCMAX + 3;
  double* Spp = sm; double* Scp = sm + lay.off_cp; double* Scc = sm + lay.off_cc;
  double* vgB = sm + lay.off_vec; double* vhd = vgB + vlen; double* vrhs = vhd + vlen;
  double* Fr = sm + lay.off_fr;       // [rep][value][frame] replicated frame-level accumulators
  double* slab = sm + lay.off_slab;   // per point of the pass: [0..5] U -> L^-1, [6..8] g
  double* Zd = sm + lay.off_zd;
  double* misc = sm + lay.off_misc;   // [0] cost, [1] bad-U count, [2] max |g_p| (as bits)
  uint32_t* pidl = (uint32_t*)(misc + MISC_DOUBLES);                 // global id of each point of the pass
  unsigned short* colinfo = (unsigned short*)(misc + MISC_DOUBLES + 32);   // per window column: pose (lf<<8 | comp), camera 0x8000|j, rhs 0xC000
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t b = blockIdx.x;
  const uint32_t flo = d.blk_flo[b], nf = d.blk_nf[b];
  const uint32_t ncol = 6 * nf + NC + 1, ncolp = (ncol + 15u) & ~15u;   // pose | camera | rhs, padded
  const uint32_t zs = ncolp + 2;   // row stride of Zd: ncolp is a multiple of 16 doubles (= all rows on one LDS bank), +2 spreads the rows
  const CamConsts c = *d.camc;
#ifdef LIFCAL_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
  if (tid == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); }
#endif
  // Pass descriptors and per-lane slot words are fetched ONE PASS AHEAD (the chain descriptor -> slot -> point -> coordinates
  // is three dependent HBM/L2 round trips otherwise, exposed at the top of every pass with one wave per SIMD)
  const uint32_t ps_begin = d.blk_pass0[b], ps_end = d.blk_pass0[b + 1];
  uint32_t nx_np = 0, nx_gid0 = 0, nx_si = 0, nx_pt = 0, nx_row0 = 0, nx_row1 = 0, nx_fp = 0;
  uint32_t vz;   // an opaque per-lane zero: keeps the descriptor loads in VGPRs (as a uniform value the compiler moves them to
                 // SGPRs with v_readfirstlane right behind the load, i.e. waits a full memory round trip at the top of every pass)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  auto fetch_pass = [&](uint32_t q) {
    nx_np = d.pass_np[q + vz]; nx_gid0 = d.pass_gid0[q + vz];
    nx_si = d.v2_slot[(size_t)q * 256 + tid]; nx_pt = d.v2f_pt[(size_t)q * 256 + tid];
    nx_row0 = d.v2_tile_row0[q * 4 + w]; nx_row1 = d.v2_tile_row0[q * 4 + w + 1];
    nx_fp = d.v2_passpt[(size_t)q * 64 + (tid & 63u)];            // point the thread factors in phase 2 (threads < np)
  };
  if (ps_begin < ps_end) fetch_pass(ps_begin);
  // (the first pass's descriptor chain is in flight while the window is zero-filled)
  { double2* z2 = reinterpret_cast<double2*>(sm); for (uint32_t i = tid; i < lay.off_slab / 2; i += 256) z2[i] = double2{0.0, 0.0}; }
  if (tid == 0 && (lay.off_slab & 1u)) sm[lay.off_slab - 1] = 0.0;
  if (tid < MISC_DOUBLES) misc[tid] = 0.0;
  for (uint32_t cI = tid; cI < ncolp; cI += 256)
    colinfo[cI] = (cI < 6 * nf) ? (unsigned short)(((cI / 6) << 8) | (cI % 6)) : (cI < ncol - 1 ? (unsigned short)(0x8000u | (cI - 6 * nf)) : (unsigned short)0xC000u);
  double cc[NCC], gc[NC], cost = 0.0;
  double lmant = 1.0; int lexp = 0;   // product of the Cauchy arguments of this lane's observations
#pragma unroll
  for (int i = 0; i < NCC; ++i) cc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < NC; ++i) gc[i] = 0.0;

  // ---- Schur product helpers: window -= Zd^T Zd in 4x4 micro-tiles over the lower triangle, one owner per output entry ----
  const uint32_t nmt = (ncol + 3u) >> 2;               // micro-tile rows/cols that contain real columns
  const uint32_t ntri = nmt * (nmt + 1) / 2;
  auto tri_decode = [](uint32_t t, uint32_t& mi, uint32_t& mj) {
    mi = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (mi * (mi + 1) / 2 > t) --mi;
    while ((mi + 1) * (mi + 2) / 2 <= t) ++mi;
    mj = t - mi * (mi + 1) / 2;
  };
  auto gemm_tile = [&](uint32_t mi, uint32_t mj, uint32_t krows, double (&acc16)[4][4]) {
    const double* za = Zd + 4 * mi;
    const double* zb = Zd + 4 * mj;
    // four rows of Zd per step (16 ds_read_b128, 64 FMAs), software-pipelined: the reads of step k+1 are issued before
    // the FMAs of step k (one wave per SIMD: nothing else hides the LDS latency); rows beyond 3 np are zero because
    // krows is rounded up to 4 and the pass zero-fills them
    double2 cur[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cur[r][0] = *reinterpret_cast<const double2*>(za + (size_t)r * zs);
      cur[r][1] = *reinterpret_cast<const double2*>(za + (size_t)r * zs + 2);
      cur[r][2] = *reinterpret_cast<const double2*>(zb + (size_t)r * zs);
      cur[r][3] = *reinterpret_cast<const double2*>(zb + (size_t)r * zs + 2);
    }
#pragma unroll 2
    for (uint32_t k0 = 0; k0 < krows; k0 += 4) {
      double2 nxt[4][4];
      const uint32_t kn = (k0 + 4 < krows) ? k0 + 4 : k0;   // the last step re-reads its own rows
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        nxt[r][0] = *reinterpret_cast<const double2*>(za + (size_t)(kn + r) * zs);
        nxt[r][1] = *reinterpret_cast<const double2*>(za + (size_t)(kn + r) * zs + 2);
        nxt[r][2] = *reinterpret_cast<const double2*>(zb + (size_t)(kn + r) * zs);
        nxt[r][3] = *reinterpret_cast<const double2*>(zb + (size_t)(kn + r) * zs + 2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double av[4] = {cur[r][0].x, cur[r][0].y, cur[r][1].x, cur[r][1].y};
        const double bq[4] = {cur[r][2].x, cur[r][2].y, cur[r][3].x, cur[r][3].y};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc16[i][j] += av[i] * bq[j];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[r][q] = nxt[r][q];
    }
  };
  auto emit_tile = [&](uint32_t mi, uint32_t mj, const double (&acc16)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t ci = 4 * mi + i;
      const uint32_t ii = colinfo[ci < ncolp ? ci : 0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t cj = 4 * mj + j;
        const double dv = acc16[i][j];
        if (ci >= ncol || cj >= ncol || ci < cj) continue;
        const uint32_t jj = colinfo[cj];
        if (!(ii & 0x8000u)) {            // pose x pose
          const uint32_t lfi = ii >> 8, lfj = jj >> 8;
          Spp[(size_t)(lfi * (lfi + 1) / 2 + lfj) * 36 + (ii & 0xFFu) * 6 + (jj & 0xFFu)] -= dv;
        } else if (!(ii & 0x4000u)) {     // camera row
          const uint32_t jc = ii & 0xFFu;
          if (!(jj & 0x8000u)) Scp[(size_t)jc * 6 * NFm + cj] -= dv;
          else Scc[jc * (jc + 1) / 2 + (jj & 0xFFu)] -= dv;
        } else if (cj < ncol - 1) {       // rhs row: W^T U^-1 g
          if (!(jj & 0x8000u)) vrhs[cj] += dv; else vrhs[6 * NFm + (jj & 0xFFu)] += dv;
        }
      }
    }
  };
  // when the triangle has at most one tile per thread, the tile is accumulated over all passes of the block in registers
  const bool keep_tiles = (mode == 0) && ntri <= 256;
  uint32_t mi0 = 0, mj0 = 0;
  if (keep_tiles && tid < ntri) tri_decode(tid, mi0, mj0);
  double tacc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) tacc[i][j] = 0.0;

  STAMP(14);
  for (uint32_t ps = ps_begin; ps < ps_end; ++ps) {
    const uint32_t np = __builtin_amdgcn_readfirstlane(nx_np);
    const uint32_t si = nx_si, pt = nx_pt, row0 = nx_row0, kmax = nx_row1 - nx_row0;
    const uint32_t fp = nx_fp;
    // Jacobi scales for the factor phase: requested now, consumed after the observation loop
    double fsg0 = 1.0, fsg1 = 1.0, fsg2 = 1.0;
    if (mode == 0 && tid < np) { fsg0 = d.sigP[3 * (size_t)fp]; fsg1 = d.sigP[3 * (size_t)fp + 1]; fsg2 = d.sigP[3 * (size_t)fp + 2]; }
    if (ps + 1 < ps_end) fetch_pass(ps + 1);
    const uint32_t krows = (3 * np + 3u) & ~3u;   // K of the product, multiple of 4
    STAMP(12);
    for (uint32_t i = tid; i < 64 * SLAB_STRIDE; i += 256) slab[i] = 0.0;
    if (mode == 0) { double2* z2 = reinterpret_cast<double2*>(Zd); for (uint32_t i = tid; i < (krows * zs) / 2; i += 256) z2[i] = double2{0.0, 0.0}; }
    STAMP(13);
    lds_barrier();
    STAMP(0);
    // ---------------- phase 1: observations -> LDS blocks ----------------
    {
      const uint32_t cnt = si & 0xFFu, lf = (si >> 8) & 0xFFu, lp = (si >> 16) & 0xFFu, rep = (si >> 24) % lay.nrep;
      const uint32_t fr = flo + lf;
      double R[9], Y[3], c0, s0;
      GroupConsts2 gcn;
      {
        const double* ft = d.ft + (size_t)fr * FRAME_STRIDE;
        const double* P = d.pts + 3 * (size_t)pt;
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = ft[k];
        const double P0 = P[0], P1 = P[1], P2 = P[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) Y[k] = R[3 * k] * P0 + R[3 * k + 1] * P1 + R[3 * k + 2] * P2;
        c0 = ft[12]; s0 = ft[13];
        group_prepare2<ADJ>(c, Y[0] + ft[9], Y[1] + ft[10], Y[2] + ft[11], gcn);
      }
      double A[6] = {0, 0, 0, 0, 0, 0}, bv[3] = {0, 0, 0}, C[3][NC];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) C[i][j] = 0.0;
      // (a 3-stage software prefetch of the next observations was measured: no gain, the loop is issue-bound)
      for (uint32_t k = 0; k < kmax; ++k) {
        if (k < cnt) {
          const size_t at = ((size_t)row0 + k) * 64 + lane;
          const double u = d.v2_u[at], v = d.v2_v[at];
          const double* L = d.lt + (size_t)d.v2_lens[at] * LENS_STRIDE;
          double r[2], Jq[2][3], Jc[2][NC];
          double arg;
          obs_eval2<NR, TAN, ADJ>(c, gcn, L, u, v, d.robust != 0, r, Jq, Jc, arg);
          if (d.robust) {
            // rho = b log(1 + s/b): the lane keeps the running PRODUCT of the arguments as mantissa x 2^exponent and takes
            // one log at the end
            int ex; lmant = frexp(lmant * arg, &ex); lexp += ex;
          } else {
            cost += 0.5 * arg;
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            A[0] += Jq[a][0] * Jq[a][0]; A[1] += Jq[a][0] * Jq[a][1]; A[2] += Jq[a][0] * Jq[a][2];
            A[3] += Jq[a][1] * Jq[a][1]; A[4] += Jq[a][1] * Jq[a][2]; A[5] += Jq[a][2] * Jq[a][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) bv[i] += Jq[a][i] * r[a];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < NC; ++j) C[i][j] += Jq[a][i] * Jc[a][j];
            int t = 0;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
              gc[i] += Jc[a][i] * r[a];
#pragma unroll
              for (int j = 0; j <= i; ++j) cc[t++] += Jc[a][i] * Jc[a][j];
            }
          }
        }
      }
      // The prefetched words are consumed (for the compiler's wait-count bookkeeping) HERE, where the in-order vmcnt is already
      // past them: otherwise their first real use sits behind the W / U^-1 stores of this pass and turns into s_waitcnt vmcnt(0),
      // i.e. a wait for HBM write completion at the top of the next pass and in the factor phase.
      asm volatile("" :: "v"(nx_si), "v"(nx_pt), "v"(nx_fp), "v"(nx_row0), "v"(nx_row1), "v"(nx_np), "v"(nx_gid0));
      asm volatile("" :: "v"(fsg0), "v"(fsg1), "v"(fsg2));
      STAMP(7);
      if (cnt > 0) {
        // Jc came back in model-parameter columns: sign/scale folding and the free-column mask, once per group
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < NC; ++j) C[i][j] *= c.chm[j];
        const double Am[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
        // Gr = [e_x x Y, (0,c0,s0) x Y, R[:,2] x Y]: d(R P)/d(a0,a1,a2)
        const double n0 = R[2], n1 = R[5], n2 = R[8];
        const double Gr[3][3] = {{0.0, c0 * Y[2] - s0 * Y[1], n1 * Y[2] - n2 * Y[1]},
                                 {-Y[2], s0 * Y[0], n2 * Y[0] - n0 * Y[2]},
                                 {Y[1], -c0 * Y[0], n0 * Y[1] - n1 * Y[0]}};
        double AG[3][3], GAG[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) AG[i][j] = Am[i][0] * Gr[0][j] + Am[i][1] * Gr[1][j] + Am[i][2] * Gr[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) GAG[i][j] = Gr[0][i] * AG[0][j] + Gr[1][i] * AG[1][j] + Gr[2][i] * AG[2][j];
        // frame-level values go to this lane's replica, value-major: every lane of the wave instruction hits its own address
        // slot = frame * R + replica: consecutive doubles across the lanes of one instruction (bank-ideal, no address clash)
        const uint32_t frs = NFm * lay.nrep;
        double* fr_acc = Fr + (size_t)rep * NFm + lf;   // replica-major: the lanes of a wave instruction land on one compact run of doubles
        {
          int vi = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int bb = 0; bb <= a; ++bb) {
              double v;
              if (a < 3) v = GAG[a][bb]; else if (bb < 3) v = AG[a - 3][bb]; else v = Am[a - 3][bb - 3];
              atomicAdd(fr_acc + (size_t)vi * frs, v);
              ++vi;
            }
#pragma unroll
          for (int a = 0; a < 3; ++a) atomicAdd(fr_acc + (size_t)(21 + a) * frs, Gr[0][a] * bv[0] + Gr[1][a] * bv[1] + Gr[2][a] * bv[2]);
#pragma unroll
          for (int a = 0; a < 3; ++a) atomicAdd(fr_acc + (size_t)(24 + a) * frs, bv[a]);
          if (mode == 0) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
#pragma unroll
              for (int ci = 0; ci < 3; ++ci) atomicAdd(fr_acc + (size_t)(27 + j * 6 + ci) * frs, C[0][j] * Gr[0][ci] + C[1][j] * Gr[1][ci] + C[2][j] * Gr[2][ci]);
#pragma unroll
              for (int ci = 0; ci < 3; ++ci) atomicAdd(fr_acc + (size_t)(27 + j * 6 + 3 + ci) * frs, C[ci][j]);
            }
          }
        }
        double AR[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) AR[i][j] = Am[i][0] * R[j] + Am[i][1] * R[3 + j] + Am[i][2] * R[6 + j];
        double* acc = slab + lp * SLAB_STRIDE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            const double u_ij = R[i] * AR[0][j] + R[3 + i] * AR[1][j] + R[6 + i] * AR[2][j];
            const int pos = (i == 0) ? 0 : (i == 1 ? (j == 0 ? 1 : 3) : (j == 0 ? 2 : (j == 1 ? 4 : 5)));
            atomicAdd(acc + pos, u_ij);
          }
        }
        if (mode == 0) {
          // only A goes to HBM for the back-substitution (48 B per lane): k_backsub rebuilds W_pose = R^T A [Gr | I] from it
          { double* ga = d.Av + (size_t)d.v2_gidx[(size_t)ps * 256 + tid] * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) ga[k] = A[k]; }
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            atomicAdd(acc + 6 + i, R[i] * bv[0] + R[3 + i] * bv[1] + R[6 + i] * bv[2]);
            double* zrow = Zd + (size_t)(3 * lp + i) * zs;
#pragma unroll
            for (int j = 0; j < NC; ++j) atomicAdd(zrow + 6 * nf + j, R[i] * C[0][j] + R[3 + i] * C[1][j] + R[6 + i] * C[2][j]);
#pragma unroll
            for (int j = 0; j < 3; ++j) { const double wij = R[i] * AG[0][j] + R[3 + i] * AG[1][j] + R[6 + i] * AG[2][j]; atomicAdd(zrow + 6 * lf + j, wij); }
#pragma unroll
            for (int j = 0; j < 3; ++j) { const double wij = R[i] * Am[0][j] + R[3 + i] * Am[1][j] + R[6 + i] * Am[2][j]; atomicAdd(zrow + 6 * lf + 3 + j, wij); }
          }
        }
      }
    }
    STAMP(6);
    lds_barrier();
    STAMP(1);
    // ---------------- phase 2: one thread per point: damp, factor U = L L^T ----------------
    if (tid < np) {
      const uint32_t p = fp;
      pidl[tid] = p;
      double* acc = slab + tid * SLAB_STRIDE;
      double U0 = acc[0], U1 = acc[1], U2 = acc[2], U3 = acc[3], U4 = acc[4], U5 = acc[5];
      if (mode == 1) {
        double* ga = d.ptacc + (size_t)p * 36;
        ga[0] = U0; ga[3] = U3; ga[5] = U5;
      } else {
        const double g0 = acc[6], g1 = acc[7], g2 = acc[8];
        double lam[3];
        {
          const double h[3] = {U0, U3, U5}, sgv[3] = {fsg0, fsg1, fsg2};
#pragma unroll
          for (int k = 0; k < 3; ++k) { const double sg = sgv[k]; lam[k] = fmin(fmax(h[k] * sg * sg, d.lm_min), d.lm_max) / (lm_radius(d, radius) * sg * sg); }
        }
        U0 += lam[0]; U3 += lam[1]; U5 += lam[2];
        // L^-1 of the 3x3 Cholesky factor through reciprocal square roots (one wave, a pure latency chain: no sqrt + divide pairs)
        bool ok = U0 > 0.0;
        double i00 = rsqrt(U0);
        const double l10 = U1 * i00, l20 = U2 * i00;
        const double d11 = U3 - l10 * l10; ok = ok && (d11 > 0.0);
        double i11 = rsqrt(d11);
        const double l21 = (U4 - l20 * l10) * i11;
        const double d22 = U5 - l20 * l20 - l21 * l21; ok = ok && (d22 > 0.0);
        double i22 = rsqrt(d22);
        double m10 = -l10 * i00 * i11, m21 = -l21 * i11 * i22, m20 = -(l20 * i00 + l21 * m10) * i22;
        if (!ok) { i00 = i11 = i22 = m10 = m21 = m20 = 0.0; atomicAdd(misc + 1, 1.0); }
        double* gu = d.Uinv + 9 * (size_t)p;
        const double v00 = i00 * i00 + m10 * m10 + m20 * m20, v01 = m10 * i11 + m20 * m21, v02 = m20 * i22;
        const double v11 = i11 * i11 + m21 * m21, v12 = m21 * i22, v22 = i22 * i22;
        gu[0] = v00; gu[1] = v01; gu[2] = v02; gu[3] = v01; gu[4] = v11; gu[5] = v12; gu[6] = v02; gu[7] = v12; gu[8] = v22;
        double* gl = d.lamP + 3 * (size_t)p; gl[0] = lam[0]; gl[1] = lam[1]; gl[2] = lam[2];
        double* ga = d.ptacc + (size_t)p * 36;
        ga[6] = g0; ga[7] = g1; ga[8] = g2;
        const double gm = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
        atomicMax((unsigned long long*)(misc + 2), (unsigned long long)__double_as_longlong(gm));
        acc[0] = i00; acc[1] = m10; acc[2] = m20; acc[3] = i11; acc[4] = m21; acc[5] = i22;
        // rhs column of Zd: L^-1 g
        double* z0 = Zd + (size_t)(3 * tid) * zs + (ncol - 1);
        z0[0] = i00 * g0; z0[zs] = m10 * g0 + i11 * g1; z0[2 * zs] = m20 * g0 + m21 * g1 + i22 * g2;
      }
    }
    lds_barrier();
    STAMP(2);
    if (mode == 0) {
      // ---------------- phase 3: camera part of W -> HBM, then Z = L^-1 W in place (pose + camera columns) ----------------
      // thread = (column, point-phase): no per-item div/mod, point ids come from LDS
      const uint32_t nwc = ncol - 1;
      const uint32_t nth = 256 / nwc > 0 ? 256 / nwc : 1;
      const uint32_t gi = tid / nwc, cidx = tid - gi * nwc;
      if (gi < nth || nwc > 256) {
        for (uint32_t cc0 = cidx; cc0 < nwc; cc0 += (nwc > 256 ? 256 : nwc * nth)) {
#pragma unroll 4
          for (uint32_t lp = (nwc > 256 ? 0 : gi); lp < np; lp += (nwc > 256 ? 1 : nth)) {
            const double* acc = slab + lp * SLAB_STRIDE;
            double* z = Zd + (size_t)(3 * lp) * zs + cc0;
            const double w0 = z[0], w1 = z[zs], w2 = z[2 * zs];
            if (cc0 >= 6 * nf) {
              double* ga = d.ptacc + (size_t)pidl[lp] * 36 + 9 + (cc0 - 6 * nf);
              ga[0] = w0; ga[NCMAX] = w1; ga[2 * NCMAX] = w2;
            }
            z[0] = acc[0] * w0; z[zs] = acc[1] * w0 + acc[3] * w1; z[2 * zs] = acc[2] * w0 + acc[4] * w1 + acc[5] * w2;
          }
        }
      }
      lds_barrier();
      STAMP(3);
      // ---------------- phase 4: window -= Zd^T Zd on the f64 matrix cores, one 16x16 output tile per wave at a time ----------------
      // Register-blocked fp64 product on the vector ALUs: measured on gfx950, v_fmac_f64 (6.3 cycles per wave
      // instruction, 64 MAC) outruns v_mfma_f64_16x16x4_f64 (~140 cycles, 1024 MAC) by ~1.4x, so the product
      // runs as 4x4 micro-tiles per lane over the lower triangle; each output entry has exactly one owner.
      if (keep_tiles) {
        if (tid < ntri) gemm_tile(mi0, mj0, krows, tacc);     // the tile stays in registers until the block's last pass
      } else {
        for (uint32_t t = tid; t < ntri; t += 256) {
          uint32_t mi, mj; tri_decode(t, mi, mj);
          double acc16[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = 0.0;
          gemm_tile(mi, mj, krows, acc16);
          emit_tile(mi, mj, acc16);
        }
      }
    }
    lds_barrier();
    STAMP(4);
  }
  STAMP(8);
  if (keep_tiles && tid < ntri) emit_tile(mi0, mj0, tacc);   // (colinfo and the window are only read/written by owners: no barrier needed before)
  STAMP(9);
  {  // the same folding for the lane's camera x camera block and camera gradient
    int t = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      gc[i] *= c.chm[i];
#pragma unroll
      for (int j = 0; j <= i; ++j) cc[t++] *= c.chm[i] * c.chm[j];
    }
  }
  if (d.robust) cost += 0.5 * c.loss_b * (log(lmant) + (double)lexp * 0.6931471805599453);
  // ---------------- camera x camera block, camera gradient, cost: 256-way reduction through LDS ----------------
  // every thread parks its partial sums as [value][thread] in the (now free) Zd region, 8 threads per value add them up
  // (row stride 264 and interleaved parts: the 32 lanes of one ds_read_b64 group touch 32 distinct bank pairs)
  {
    constexpr int NV = NCC + NC + 1, RV = 28;   // 28 values x 256 threads x 8 B = 56 KiB per round
#pragma unroll
    for (int round = 0; round * RV < NV; ++round) {
#pragma unroll
      for (int v = 0; v < RV; ++v) {
        const int idx = round * RV + v;
        if (idx < NV) Zd[v * 264 + tid] = (idx < NCC) ? cc[idx < NCC ? idx : 0] : (idx < NCC + NC ? gc[(idx - NCC) >= 0 && (idx - NCC) < NC ? idx - NCC : 0] : cost);
      }
      __syncthreads();
      if (tid < RV * 8) {
        const int v = tid >> 3, part = tid & 7, idx = round * RV + v;
        if (idx < NV) {
          double sacc = 0.0;
#pragma unroll 8
          for (int k = 0; k < 32; ++k) sacc += Zd[v * 264 + part + 8 * k];
          if (idx < NCC) {
            int i = 0; while ((i + 1) * (i + 2) / 2 <= idx) ++i;
            if (mode == 0) atomicAdd(Scc + idx, sacc);
            if (idx == i * (i + 1) / 2 + i) atomicAdd(vhd + 6 * NFm + i, sacc);
          } else if (idx < NCC + NC) {
            atomicAdd(vgB + 6 * NFm + (idx - NCC), sacc);
          } else {
            atomicAdd(misc + 0, sacc);
          }
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  STAMP(10);
  // fold the replicas: pose x pose diagonal blocks (+ Hessian diagonal), pose gradient, camera x pose
  for (uint32_t i = tid; i < FRV * nf; i += 256) {
    const uint32_t v = i / nf, lf = i % nf;
    double sacc = 0.0;
    for (uint32_t r = 0; r < lay.nrep; ++r) sacc += Fr[(size_t)v * NFm * lay.nrep + r * NFm + lf];
    if (v < 21) {
      uint32_t a = 0; while ((a + 1) * (a + 2) / 2 <= v) ++a;
      const uint32_t bb = v - a * (a + 1) / 2;
      Spp[(size_t)(lf * (lf + 1) / 2 + lf) * 36 + a * 6 + bb] += sacc;
      if (a == bb) vhd[6 * lf + a] += sacc;
    } else if (v < 27) {
      vgB[6 * lf + (v - 21)] += sacc;
    } else if (mode == 0 && v < 27 + 6 * (uint32_t)NC) {
      const uint32_t j = (v - 27) / 6, ci = (v - 27) % 6;
      Scp[(size_t)j * 6 * NFm + 6 * lf + ci] += sacc;
    }
  }
  __syncthreads();
  STAMP(11);
  // ---------------- flush the window into the global reduced system (contiguous runs) ----------------
  const uint32_t F6 = 6 * d.F, camrow = 3 * d.Q, camcol = F6 + 3 * d.Q;
  for (uint32_t i = tid; i < 6 * nf; i += 256) atomicAdd(d.hdiag + 6 * flo + i, vhd[i]);
  if (tid < (uint32_t)NC) atomicAdd(d.hdiag + camcol + tid, vhd[6 * NFm + tid]);
  if (mode == 0) {
    const uint32_t npp = nf * (nf + 1) / 2;
    for (uint32_t i = tid; i < npp * 36; i += 256) {
      const uint32_t blk = i / 36, e = i % 36;
      uint32_t a = (uint32_t)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > blk) --a;
      while ((a + 1) * (a + 2) / 2 <= blk) ++a;
      const uint32_t bb = blk - a * (a + 1) / 2, dd = a - bb;
      const double v = Spp[i];
      if (dd <= d.bw && v != 0.0 && !(dd == 0 && (e % 6) > (e / 6))) atomicAdd(d.Sband + ((size_t)(flo + a) * (d.bw + 1) + dd) * 36 + e, v);
    }
    for (uint32_t i = tid; i < (uint32_t)NC * 6 * nf; i += 256) {
      const uint32_t j = i / (6 * nf), cidx = i % (6 * nf);
      atomicAdd(d.Sarrow + (size_t)(camrow + j) * d.ld + 6 * flo + cidx, Scp[(size_t)j * 6 * NFm + cidx]);
    }
    if (tid < (uint32_t)NCC) {
      uint32_t i = 0; while ((i + 1) * (i + 2) / 2 <= tid) ++i;
      const uint32_t j = tid - i * (i + 1) / 2;
      atomicAdd(d.Sarrow + (size_t)(camrow + i) * d.ld + camcol + j, Scc[tid]);
    }
    for (uint32_t i = tid; i < 6 * nf; i += 256) { atomicAdd(d.gB + 6 * flo + i, vgB[i]); atomicAdd(d.rhsacc + 6 * flo + i, vrhs[i]); }
    if (tid < (uint32_t)NC) { atomicAdd(d.gB + camcol + tid, vgB[6 * NFm + tid]); atomicAdd(d.rhsacc + camcol + tid, vrhs[6 * NFm + tid]); }
    if (tid == 0) {
      atomicAdd(d.scal + SCAL_COST, misc[0]);
      if (misc[1] != 0.0) atomicAdd(d.scal + SCAL_BAD_U, misc[1]);
      atomicMax((unsigned long long*)(d.scal + SCAL_GMAX0 + d.rank), *(unsigned long long*)(misc + 2));
    }
  }
#ifdef LIFCAL_STAMPS
  __syncthreads();
  STAMP(5);
  if (tid == 0 && d.dbg) for (int i = 0; i < 16; ++i) d.dbg[(size_t)b * 32 + i] = st_acc[i];
#endif
}

}  // namespace lifcal
