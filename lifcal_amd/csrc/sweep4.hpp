// sweep4.hpp — the fused Jacobian + Schur sweep as TWO kernels built for occupancy (included by kernels.hpp).
//
// k_sweep3 (sweep3.hpp) runs one 512-thread workgroup per CU whose phases — observation loop, emission, point factorisation,
// Z = L^-1 W, Schur product, flush — follow each other with the whole CU waiting on the slowest wave: the evaluator waves, whose
// dependent fp64 chain is the floor of the sweep, evaluate ~40 % of the time (profiles/r02_final).  Here the same mathematics is
// cut where its data flow is thin, and the evaluator never waits for anything but its operands:
//
//   k_front4   observations -> blocks.  A workgroup is a STREAM of tiles (64 lanes, lane = (point, frame) group, whole points,
//              lanes sorted by frame: plan.hpp, 64-lane passes) worked by FOUR waves with no workgroup barrier between tiles:
//                E   evaluator     residual, Jq, Jc per observation (device_model.hpp) from a lens row it finds in LDS; hands
//                                  Jq, Jc, r to the accumulators through LDS.  No global memory access on its path.
//                Y   accumulator   camera x camera block and camera gradient (54 running sums per thread)
//                M0  M1            tiles alternate between them.  While a tile is evaluated its M accumulates A = sum Jq^T Jq,
//                                  b, C = sum Jq^T Jc AND feeds the evaluator: the tile's camera-frame points, then one lens row
//                                  + observation words per step (registers -> LDS, one step ahead).  While the NEXT tile is
//                                  evaluated it emits: frame-level values summed over runs of lanes with DPP, run heads add to
//                                  the workgroup's frame accumulators (k_sweep3's scheme); point-level values (U, g, W_cam)
//                                  staged per lane and summed per point by a gather — plain stores, no LDS atomics.
//              Out: A per lane, (U, g, W_cam) per point — what the back-substitution reads anyway — and the frame / camera sums
//              flushed once per workgroup.  <= 168 registers, ~52 KB of LDS: three streams (twelve waves) per CU.
//   k_back4    blocks -> reduced system.  One 1024-thread workgroup (<= 128 registers, four waves per SIMD) per block of points
//              sharing a frame window: thread-per-point damping + 3x3 factorisation for the whole block at once, then chunks of
//              32 points: W rows rebuilt from A and the frame table (as k_backsub does), Z = L^-1 W, window -= Z^T Z as 4x4
//              register tiles with the K dimension split over four thread groups; one flush per block.
//
// Replaces, per LM iteration: ceres autodiff evaluation of OurCostFunctionBundle (reference
// src/BundleAdjustment/BundleAdjustment.h:120-222) + SchurEliminator::Eliminate (out of tree); same outputs as k_sweep3.
#pragma once

namespace lifcal {

#ifndef F4_ROLES
#define F4_ROLES 15   // (diagnostic builds: bit mask of the roles compiled in, to read each role's register needs)
#endif
#ifndef F4_SLEEP
#define F4_SLEEP 1   // s_sleep argument of the counter polls (64 cycles each)
#endif
constexpr uint32_t F4_THREADS = 256;              // k_front4: four waves
constexpr uint32_t F4_HAND = (6 + 2 * NCMAX + 2) * 64u;   // hand-off slot [value][lane]
constexpr uint32_t F4_LENS = 9u * 128u;           // lens slot [16-byte chunk][lane][2]: the lens row (8 chunks) and (u, v) of ONE step
constexpr uint32_t F4_XYZ = 2u * 3u * 64u;        // camera-frame points of the lanes of the next two tiles
constexpr uint32_t F4_STAGE_STRIDE = 65;          // per-point staging [value][65] (odd: the gather reads one column for all values at once)
constexpr uint32_t F4_STAGE = (9 + 3 * NCMAX) * F4_STAGE_STRIDE;
constexpr uint32_t F4_MISC = 16 + 16 + 64;        // doubles: u32 counters (16) | run table of the emitting tile (32 u32) | its point table (64 x 2 u32)
struct V4Front {
  uint32_t nfm, off_lens, off_xyz, off_stage, off_fr, off_misc, total;
  __host__ __device__ explicit V4Front(uint32_t nfmax) {
    nfm = nfmax; off_lens = F4_HAND; off_xyz = off_lens + F4_LENS; off_stage = off_xyz + F4_XYZ; off_fr = off_stage + F4_STAGE;
    off_misc = off_fr + FRV * nfm; off_misc = (off_misc + 1) & ~1u; total = off_misc + F4_MISC;
  }
};
// counters (u32 in the misc block, monotone over the kernel; G = observation steps of the stream so far, t = tile of the stream)
enum { C4_W = 0,      // steps the evaluator has handed over
       C4_RA = 1,     // steps taken by the accumulating M
       C4_RY = 2,     // steps taken by Y
       C4_LW = 3,     // lens rows written
       C4_LR = 4,     // lens rows taken by the evaluator
       C4_TX0 = 5,    // tiles of parity 0 / 1 whose camera-frame points are in LDS
       C4_TX1 = 6,
       C4_GT = 7,     // tiles whose staging has been consumed (emission complete)
       C4_ERR = 8 };  // a wait ran into its bound

constexpr uint32_t B4_THREADS = 1024, B4_CP = 32, B4_ZR = 3 * B4_CP, B4_BATCH = 256, B4_LI = 10;   // chunk of points, Z rows, points factored at once, doubles per factored point
struct V4Back {
  uint32_t ncolp, zs, off_li, off_misc, off_pid, off_col, total;
  __host__ __device__ explicit V4Back(uint32_t nfmax) {
    ncolp = ((6 * nfmax + NCMAX + 1) + 15u) & ~15u; zs = ncolp + 2;
    off_li = B4_ZR * zs; off_misc = off_li + B4_BATCH * B4_LI; off_pid = off_misc + 8; off_col = off_pid + B4_BATCH / 2; total = off_col + (ncolp + 3) / 4 + 2;
  }
  // tiles of the lower triangle a thread group of 256 owns at most this many each
  __host__ __device__ static uint32_t ntri_of(uint32_t nf, uint32_t nc) { const uint32_t nmt = (6 * nf + nc + 1 + 3) >> 2; return nmt * (nmt + 1) / 2; }
};

template <int I, int N, class F>
LIFCAL_DEV void static_for4(F& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for4<I + 1, N>(f); } }

template <int NR, bool TAN, bool ADJ>
__global__ __launch_bounds__(F4_THREADS, 3) void k_front4(Dev d, int mode) {
  constexpr int NC = 5 + NR + (TAN ? 2 : 0);
  constexpr int NCC = NC * (NC + 1) / 2;
  constexpr int HV = 6 + 2 * NC + 2;      // doubles handed over per lane and step: Jq (6) | Jc (2 NC) | r (2)
  constexpr uint32_t NV = 9 + 3 * NC;     // per-lane point-level values: U (6) | g (3) | W_cam (3 NC)
  static_assert(HV * 64 <= F4_HAND, "hand-off slot");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const V4Front lay(d.v2_nfmax);
  const uint32_t NFm = lay.nfm;
  double* hand = sm;                      // [HV][64]
  double* lens = sm + lay.off_lens;       // [chunk][lane][2]
  double* xyz = sm + lay.off_xyz;         // [tile parity][3][64]
  double* stage = sm + lay.off_stage;     // [NV][65]
  double* Fr = sm + lay.off_fr;           // [value][frame]
  uint32_t* ctr = (uint32_t*)(sm + lay.off_misc);
  uint32_t* runtab = ctr + 32;            // [run]: first lane | frame << 8 | pose is free << 16
  uint32_t* pttab = ctr + 64;             // [point of the tile][2]: point id, first lane (gid order) | lanes << 16
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t wg = blockIdx.x;
  // roles rotate with the workgroup index: whatever SIMD the hardware gives wave i of every workgroup, it hosts all four roles
  const uint32_t role = (w + wg) & 3u;    // 0 evaluator, 1 Y, 2 M0, 3 M1
  const uint32_t b = d.fwg_blk[wg];
  const uint32_t ps_begin = d.fwg_pass0[wg], ps_end = d.fwg_pass0[wg + 1];
  const uint32_t flo = d.blk_flo[b], nf = d.blk_nf[b];
  const CamConsts c = *d.camc;
  for (uint32_t i = tid; i < FRV * NFm; i += F4_THREADS) Fr[i] = 0.0;
  if (tid < 32) ctr[tid] = 0u;
  uint32_t vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  lds_barrier();

  // rotation, rotated point and d(R P)/d(angles) of a lane's (point, frame) from its frame row (R | cos a0, sin a0) and point
  struct LaneGeom { double R[9], Y[3], Gr[3][3]; };
  auto lane_geom = [&](const double (&ftv)[11], double P0, double P1, double P2, LaneGeom& q) {
#pragma unroll
    for (int k = 0; k < 9; ++k) q.R[k] = ftv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) q.Y[k] = q.R[3 * k] * P0 + q.R[3 * k + 1] * P1 + q.R[3 * k + 2] * P2;
    const double c0 = ftv[9], s0 = ftv[10], n0 = q.R[2], n1 = q.R[5], n2 = q.R[8];
    // Gr = [e_x x Y, (0,c0,s0) x Y, R[:,2] x Y]
    q.Gr[0][0] = 0.0;     q.Gr[0][1] = c0 * q.Y[2] - s0 * q.Y[1]; q.Gr[0][2] = n1 * q.Y[2] - n2 * q.Y[1];
    q.Gr[1][0] = -q.Y[2]; q.Gr[1][1] = s0 * q.Y[0];               q.Gr[1][2] = n2 * q.Y[0] - n0 * q.Y[2];
    q.Gr[2][0] = q.Y[1];  q.Gr[2][1] = -c0 * q.Y[0];              q.Gr[2][2] = n0 * q.Y[1] - n1 * q.Y[0];
  };
  // (every spin is bounded: a protocol error must show up as a wrong result — C4_ERR, reported through SCAL_BAD_U — not as a hung GPU)
  auto wait_ge = [&](uint32_t which, uint32_t need) {
    uint32_t spins = 0;
    while (__hip_atomic_load(ctr + which, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
      __builtin_amdgcn_s_sleep(F4_SLEEP);
      if (++spins > (1u << 20)) { ctr[C4_ERR] = 1u; break; }
    }
  };
  auto publish = [&](uint32_t which, uint32_t value) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's LDS traffic up to here is done before the counter moves
    __hip_atomic_store(ctr + which, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
#ifdef LIFCAL_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
  if (lane == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); }
#define FSTAMP(i) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[i] += t_ - st_last; st_last = t_; } } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

  uint32_t gbase = 0;   // observation steps of the stream's tiles before the current one
  if (role == 0 && (F4_ROLES & 1)) {
    // =====================================================================================================================
    // evaluator: residual, Jq, Jc of every observation of every tile of the stream.  Operands come through LDS (camera-frame
    // points per tile, lens row + observation words per step); the only global loads are the tile descriptors, a tile ahead.
    // =====================================================================================================================
    double cost = 0.0, lmant = 1.0; int lexp = 0;
    uint32_t nx_si = 0, nx_r0 = 0, nx_r1 = 0;
    auto fetch = [&](uint32_t q) { nx_si = d.v2_slot[(size_t)q * 64 + lane]; nx_r0 = d.v2_tile_row0[q + vz]; nx_r1 = d.v2_tile_row0[q + 1 + vz]; };
    if (ps_begin < ps_end) fetch(ps_begin);
    for (uint32_t ps = ps_begin; ps < ps_end; ++ps) {
      const uint32_t t = ps - ps_begin;
      const uint32_t cnt = nx_si & 0xFFu;
      const uint32_t kmax = __builtin_amdgcn_readfirstlane(nx_r1) - __builtin_amdgcn_readfirstlane(nx_r0);
      if (ps + 1 < ps_end) fetch(ps + 1);
      GroupConsts2 gcn;
      {
        wait_ge(C4_TX0 + (t & 1u), (t >> 1) + 1u);
        const double* xs = xyz + (size_t)(t & 1u) * 192 + lane;
        group_prepare2<ADJ>(c, xs[0], xs[64], xs[128], gcn);
      }
      FSTAMP(0);
      for (uint32_t k = 0; k < kmax; ++k) {
        const uint32_t g = gbase + k;
        wait_ge(C4_LW, g + 1);
        FSTAMP(1);
        double L[LENS_STRIDE], u, v;
        {
          const double* ls = lens + 2 * lane;
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) { const double2 q2 = *reinterpret_cast<const double2*>(ls + ch * 128); L[2 * ch] = q2.x; L[2 * ch + 1] = q2.y; }
          const double2 uv = *reinterpret_cast<const double2*>(ls + 8 * 128); u = uv.x; v = uv.y;
        }
        publish(C4_LR, g + 1);           // the row is in registers: the slot may take the next one
        // the hand-off slot is free once both accumulators have taken step g - 1
        wait_ge(C4_RA, g); wait_ge(C4_RY, g);
        FSTAMP(2);
        if (k < cnt) {
          double r[2], Jq[2][3], Jc[2][NC], arg;
          obs_eval2<NR, TAN, ADJ>(c, gcn, L, u, v, d.robust != 0, r, Jq, Jc, arg);
          double* hs = hand + lane;
#pragma unroll
          for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int j = 0; j < NC; ++j) hs[(6 + a * NC + j) * 64] = Jc[a][j];
#pragma unroll
            for (int i = 0; i < 3; ++i) hs[(a * 3 + i) * 64] = Jq[a][i];
            hs[(6 + 2 * NC + a) * 64] = r[a];
          }
          if (d.robust) { int ex; lmant = frexp(lmant * arg, &ex); lexp += ex; }   // rho = b log(prod (1 + s/b)): one log per lane at the end
          else cost += 0.5 * arg;
        }
        publish(C4_W, g + 1);
        FSTAMP(3);
      }
      gbase += kmax;
    }
    if (d.robust) cost += 0.5 * c.loss_b * (log(lmant) + (double)lexp * 0.6931471805599453);
    cost = wave_sum_dpp(cost);
    if (lane == 63 && mode == 0) atomicAdd(d.scal + SCAL_COST, cost);
  } else if (role == 1 && (F4_ROLES & 2)) {
    // =====================================================================================================================
    // accumulator Y: camera x camera block and camera gradient of the thread over the whole stream
    // =====================================================================================================================
    double cc[NCC], gc[NC];
#pragma unroll
    for (int i = 0; i < NCC; ++i) cc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) gc[i] = 0.0;
    uint32_t nx_si = 0, nx_r0 = 0, nx_r1 = 0;
    auto fetch = [&](uint32_t q) { nx_si = d.v2_slot[(size_t)q * 64 + lane]; nx_r0 = d.v2_tile_row0[q + vz]; nx_r1 = d.v2_tile_row0[q + 1 + vz]; };
    if (ps_begin < ps_end) fetch(ps_begin);
    for (uint32_t ps = ps_begin; ps < ps_end; ++ps) {
      const uint32_t cnt = nx_si & 0xFFu;
      const uint32_t kmax = __builtin_amdgcn_readfirstlane(nx_r1) - __builtin_amdgcn_readfirstlane(nx_r0);
      if (ps + 1 < ps_end) fetch(ps + 1);
      for (uint32_t k = 0; k < kmax; ++k) {
        const uint32_t g = gbase + k;
        wait_ge(C4_W, g + 1);
        FSTAMP(1);
        const bool live = k < cnt;
        const double* hs = hand + lane;
        // one residual row at a time; the step is released after the second row is read
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          double ra = 0.0, Jc[NC];
#pragma unroll
          for (int j = 0; j < NC; ++j) Jc[j] = 0.0;
          if (live) {
#pragma unroll
            for (int j = 0; j < NC; ++j) Jc[j] = hs[(6 + a * NC + j) * 64];
            ra = hs[(6 + 2 * NC + a) * 64];
          }
          if (a == 1) publish(C4_RY, g + 1);
          int tt = 0;
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            gc[i] += Jc[i] * ra;
#pragma unroll
            for (int j = 0; j <= i; ++j) cc[tt++] += Jc[i] * Jc[j];
          }
        }
        FSTAMP(2);
      }
      gbase += kmax;
    }
    const uint32_t camrow = 3 * d.Q, camcol = 6 * d.F + 3 * d.Q;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const double v = wave_sum_dpp(gc[i] * c.chm[i]);
      if (lane == 63 && mode == 0) atomicAdd(d.gB + camcol + i, v);
    }
    int tt = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const double v = wave_sum_dpp(cc[tt] * (c.chm[i] * c.chm[j]));
        ++tt;
        if (lane == 63) {
          if (mode == 0) atomicAdd(d.Sarrow + (size_t)(camrow + i) * d.ld + camcol + j, v);
          if (i == j) atomicAdd(d.hdiag + camcol + i, v);
        }
      }
  } else if (role >= 2 && (F4_ROLES & 4)) {
    // =====================================================================================================================
    // M0 / M1: every second tile of the stream.  Phase 1 (its tile is being evaluated): feed the evaluator — camera-frame
    // points, then one lens row + (u, v) per step, a step ahead — and accumulate A, b, C of the lane.  Phase 2 (the other M's
    // tile is being evaluated): emission of its tile.
    // =====================================================================================================================
    const uint32_t par = role - 2u;
    uint32_t ps = ps_begin;
    uint32_t nx_r0 = 0, nx_r1 = 0;
    // the steps of the other M's tiles count too: walk the row table of every tile (scalar loads)
    for (; ps < ps_end; ++ps) {
      const uint32_t t = ps - ps_begin;
      const uint32_t row0 = d.v2_tile_row0[ps], kmax = d.v2_tile_row0[ps + 1] - row0;
      if ((t & 1u) != par) { gbase += kmax; continue; }
      (void)nx_r0; (void)nx_r1;
      const uint32_t np = d.pass_np[ps], pp0 = d.pass_pt0[ps], gid0 = d.pass_gid0[ps];
      const uint32_t si = d.v2_slot[(size_t)ps * 64 + lane], pt = d.v2f_pt[(size_t)ps * 64 + lane], gidx = d.v2_gidx[(size_t)ps * 64 + lane];
      const uint32_t cnt = si & 0xFFu, lf = (si >> 8) & 0xFFu, fr = flo + lf;
      // ---- phase 1 ----
      double row[LENS_STRIDE], ru, rv;
      uint32_t lnext;
      {
        // first lens row of the tile (requested before anything else: the longest chain), camera-frame points of the tile's lanes
        const size_t at = (size_t)row0 * 64 + lane;
        const uint32_t l0 = d.v2_lens[at];
        const double* ft = d.ft + (size_t)fr * FRAME_STRIDE;
        const double* P = d.pts + 3 * (size_t)pt;
        double f12[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) f12[i] = ft[i];
        const double P0 = P[0], P1 = P[1], P2 = P[2];
        ru = d.v2_u[at]; rv = d.v2_v[at];
        lnext = d.v2_lens[((size_t)row0 + (kmax > 1 ? 1u : 0u)) * 64 + lane];
        const double* Lp = d.lt + (size_t)l0 * LENS_STRIDE;
#pragma unroll
        for (int i = 0; i < LENS_STRIDE; ++i) row[i] = Lp[i];
        double* xs = xyz + (size_t)(t & 1u) * 192 + lane;
        xs[0] = f12[0] * P0 + f12[1] * P1 + f12[2] * P2 + f12[9];
        xs[64] = f12[3] * P0 + f12[4] * P1 + f12[5] * P2 + f12[10];
        xs[128] = f12[6] * P0 + f12[7] * P1 + f12[8] * P2 + f12[11];
        publish(C4_TX0 + (t & 1u), (t >> 1) + 1u);
      }
      double A[6] = {0, 0, 0, 0, 0, 0}, bv[3] = {0, 0, 0}, C[3][NC];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) C[i][j] = 0.0;
      FSTAMP(0);
      for (uint32_t k = 0; k < kmax; ++k) {
        const uint32_t g = gbase + k;
        // the lens slot is free once the evaluator has taken row g - 1
        wait_ge(C4_LR, g);
        FSTAMP(1);
        {
          double* ls = lens + 2 * lane;
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) *reinterpret_cast<double2*>(ls + ch * 128) = double2{row[2 * ch], row[2 * ch + 1]};
          *reinterpret_cast<double2*>(ls + 8 * 128) = double2{ru, rv};
        }
        publish(C4_LW, g + 1);
        if (k + 1 < kmax) {   // next step's row and words into registers (rows are padded: idle lanes read valid memory)
          const double* Lp = d.lt + (size_t)lnext * LENS_STRIDE;
#pragma unroll
          for (int i = 0; i < LENS_STRIDE; ++i) row[i] = Lp[i];
          const size_t at = ((size_t)row0 + k + 1) * 64 + lane;
          ru = d.v2_u[at]; rv = d.v2_v[at];
          lnext = d.v2_lens[((size_t)row0 + (k + 2 < kmax ? k + 2 : k + 1)) * 64 + lane];
        }
        FSTAMP(2);
        wait_ge(C4_W, g + 1);
        FSTAMP(3);
        const bool live = k < cnt;
        const double* hs = hand + lane;
        // one residual row at a time (13 doubles live instead of 26); the step is released after the second row is read
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          double ra = 0.0, Jq[3] = {0.0, 0.0, 0.0}, Jc[NC];
#pragma unroll
          for (int j = 0; j < NC; ++j) Jc[j] = 0.0;
          if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) Jq[i] = hs[(a * 3 + i) * 64];
#pragma unroll
            for (int j = 0; j < NC; ++j) Jc[j] = hs[(6 + a * NC + j) * 64];
            ra = hs[(6 + 2 * NC + a) * 64];
          }
          if (a == 1) publish(C4_RA, g + 1);
          const double q0 = Jq[0], q1 = Jq[1], q2 = Jq[2];
          A[0] += q0 * q0; A[1] += q0 * q1; A[2] += q0 * q2;
          A[3] += q1 * q1; A[4] += q1 * q2; A[5] += q2 * q2;
          bv[0] += q0 * ra; bv[1] += q1 * ra; bv[2] += q2 * ra;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NC; ++j) C[i][j] += Jq[i] * Jc[j];
        }
        FSTAMP(4);
      }
      gbase += kmax;
      // ---- phase 2: emission (the other M feeds the evaluator meanwhile) ----
      // Frame-level values (pose x pose block, pose gradient, camera x pose block of the lane's frame: 27 + 6 NC per lane) are summed
      // over the run of lanes of each frame THROUGH LDS: every lane stores a chunk of 27 values to the staging columns, then one
      // thread per (value, run) adds the run's contiguous entries and adds the sum to the workgroup's frame accumulators — a
      // tenth of the instructions of k_sweep3's DPP run sums (12 per value and lane).  One wave: no barrier, only its own waitcnt.
      LaneGeom q;
      bool fpose;
      {
        const double* ft = d.ft + (size_t)fr * FRAME_STRIDE;
        const double* P = d.pts + 3 * (size_t)pt;
        double gft[11];
#pragma unroll
        for (int k = 0; k < 9; ++k) gft[k] = ft[k];
        gft[9] = ft[12]; gft[10] = ft[13];
        const double P0 = P[0], P1 = P[1], P2 = P[2];
        fpose = d.frame_live[fr] != 0;   // 0: the pose of this frame is held constant (lifcal_ba_set_fixed_frames): no pose columns
        lane_geom(gft, P0, P1, P2, q);
      }
      if (cnt > 0) {
        double* ga = d.Av + (size_t)gidx * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) ga[k] = A[k];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) C[i][j] *= c.chm[j];   // sign/scale folding and free-column mask, once per lane
      const double Am[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
      double AG[3][3], GAG[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) AG[i][j] = Am[i][0] * q.Gr[0][j] + Am[i][1] * q.Gr[1][j] + Am[i][2] * q.Gr[2][j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) GAG[i][j] = q.Gr[0][i] * AG[0][j] + q.Gr[1][i] * AG[1][j] + q.Gr[2][i] * AG[2][j];
      FSTAMP(5);
      // the staging (and the run / point tables) belong to one tile at a time: the previous tile's emission is complete
      wait_ge(C4_GT, t);
      FSTAMP(6);
      // runs of lanes with one frame (lanes are sorted by frame, idle lanes last): heads in lane order
      const uint32_t prev_lf = (uint32_t)__shfl_up((int)lf, 1, 64);
      const bool head = cnt > 0 && (lane == 0 || prev_lf != lf);
      const unsigned long long hmask = __ballot(head), amask = __ballot(cnt > 0);
      const uint32_t nruns = (uint32_t)__popcll(hmask), nact = (uint32_t)__popcll(amask);
      if (head) runtab[__popcll(hmask & ((1ull << lane) - 1ull))] = lane | (lf << 8) | ((fpose ? 1u : 0u) << 16);
      if (lane == 0) runtab[nruns] = nact;
      if (lane < np) { pttab[2 * lane] = d.v2_points[pp0 + lane]; pttab[2 * lane + 1] = d.v2_ptinfo[pp0 + lane]; }
      const uint32_t nfv = (mode == 0) ? (27u + 6u * (uint32_t)NC) : 27u;
      auto frame_value = [&](auto VI) -> double {   // value V of the lane (the order of the frame accumulators)
        constexpr int V = decltype(VI)::value;
        if constexpr (V < 21) {
          constexpr int a = V < 1 ? 0 : (V < 3 ? 1 : (V < 6 ? 2 : (V < 10 ? 3 : (V < 15 ? 4 : 5))));
          constexpr int bb = V - a * (a + 1) / 2;
          if constexpr (a < 3) return GAG[a][bb]; else if constexpr (bb < 3) return AG[a - 3][bb]; else return Am[a - 3][bb - 3];
        } else if constexpr (V < 24) {
          constexpr int a = V - 21;
          return q.Gr[0][a] * bv[0] + q.Gr[1][a] * bv[1] + q.Gr[2][a] * bv[2];
        } else if constexpr (V < 27) {
          return bv[V - 24];
        } else {
          constexpr int j = (V - 27) / 6, ci = (V - 27) % 6;
          if constexpr (ci < 3) return C[0][j] * q.Gr[0][ci] + C[1][j] * q.Gr[1][ci] + C[2][j] * q.Gr[2][ci]; else return C[ci - 3][j];
        }
      };
      auto emit_chunk = [&](auto VB) {
        constexpr int vb = decltype(VB)::value;
        if ((uint32_t)vb >= nfv) return;
        if (cnt > 0) {
          double* st = stage + lane;
          auto store_one = [&](auto I) {
            constexpr int V = vb + decltype(I)::value;
            if constexpr (V < 27 + 6 * NC) st[decltype(I)::value * F4_STAGE_STRIDE] = frame_value(std::integral_constant<int, V>{});
          };
          static_for4<0, 27>(store_one);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own stores are done before its own loads)
        const uint32_t nval = (nfv - (uint32_t)vb < 27u) ? nfv - (uint32_t)vb : 27u;
        for (uint32_t it = lane; it < 27u * nruns; it += 64) {
          const uint32_t r = (it * 2428u) >> 16, v = it - r * 27u;   // it / 27 for it < 27 * 64
          const uint32_t rt = runtab[r], l0 = rt & 0xFFu, l1 = runtab[r + 1] & 0xFFu, rlf = (rt >> 8) & 0xFFu;
          if (v >= nval || !((rt >> 16) & 1u)) continue;
          const double* sp = stage + v * F4_STAGE_STRIDE;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // (four reads in flight: the run is a chain of LDS round trips otherwise)
          uint32_t k = l0;
          for (; k + 4 <= l1; k += 4) { a0 += sp[k]; a1 += sp[k + 1]; a2 += sp[k + 2]; a3 += sp[k + 3]; }
          for (; k < l1; ++k) a0 += sp[k];
          atomicAdd(Fr + (size_t)(vb + v) * NFm + rlf, (a0 + a1) + (a2 + a3));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      emit_chunk(std::integral_constant<int, 0>{});
      emit_chunk(std::integral_constant<int, 27>{});
      emit_chunk(std::integral_constant<int, 54>{});
      FSTAMP(8);
      if (cnt > 0) {   // the lane's share of its point's U, g and W_cam = R^T C, world frame, to the staging columns
        const double (&R)[9] = q.R;
        double AR[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) AR[i][j] = Am[i][0] * R[j] + Am[i][1] * R[3 + j] + Am[i][2] * R[6 + j];
        double* st = stage + (gidx - gid0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            const int pos = (i == 0) ? 0 : (i == 1 ? (j == 0 ? 1 : 3) : (j == 0 ? 2 : (j == 1 ? 4 : 5)));
            st[pos * F4_STAGE_STRIDE] = R[i] * AR[0][j] + R[3 + i] * AR[1][j] + R[6 + i] * AR[2][j];
          }
          st[(6 + i) * F4_STAGE_STRIDE] = R[i] * bv[0] + R[3 + i] * bv[1] + R[6 + i] * bv[2];
#pragma unroll
          for (int j = 0; j < NC; ++j) st[(9 + i * NC + j) * F4_STAGE_STRIDE] = R[i] * C[0][j] + R[3 + i] * C[1][j] + R[6 + i] * C[2][j];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // per-point sums of the staged lane values -> ptacc
      for (uint32_t i = lane; i < np * NV; i += 64) {
        const uint32_t j = i / NV, v = i - j * NV;
        const uint32_t p = pttab[2 * j], info = pttab[2 * j + 1];
        const uint32_t g0 = info & 0xFFFFu, n = info >> 16;
        const double* sp = stage + v * F4_STAGE_STRIDE + g0;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        uint32_t k = 0;
        for (; k + 4 <= n; k += 4) { a0 += sp[k]; a1 += sp[k + 1]; a2 += sp[k + 2]; a3 += sp[k + 3]; }
        for (; k < n; ++k) a0 += sp[k];
        const uint32_t at = v < 9 ? v : 9 + ((v - 9) / NC) * NCMAX + (v - 9) % NC;
        d.ptacc[(size_t)p * 36 + at] = (a0 + a1) + (a2 + a3);
      }
      publish(C4_GT, t + 1);
      FSTAMP(7);
    }
  }
  // =======================================================================================================================
  // flush of the workgroup's frame accumulators (all four waves)
  // =======================================================================================================================
  lds_barrier();
  if (tid == 0 && ctr[C4_ERR] != 0u) atomicAdd(d.scal + SCAL_BAD_U, 1.0e9);   // a wait ran into its bound
  const uint32_t camrow = 3 * d.Q;
  const uint32_t nfv = (mode == 0) ? (27u + 6u * (uint32_t)NC) : 21u;   // (diagonal-only pass: the pose x pose diagonal)
  for (uint32_t i = tid; i < nfv * nf; i += F4_THREADS) {
    const uint32_t v = i / nf, lf = i - v * nf;
    const double s = Fr[(size_t)v * NFm + lf];
    if (s == 0.0) continue;
    const uint32_t fr = flo + lf;
    if (v < 21) {
      uint32_t a = 0; while ((a + 1) * (a + 2) / 2 <= v) ++a;
      const uint32_t bb = v - a * (a + 1) / 2;
      if (mode == 0) atomicAdd(d.Sband + (size_t)fr * (d.bw + 1) * 36 + a * 6 + bb, s);
      if (a == bb) atomicAdd(d.hdiag + 6 * fr + a, s);
    } else if (v < 27) {
      if (mode == 0) atomicAdd(d.gB + 6 * fr + (v - 21), s);
    } else {
      const uint32_t j = (v - 27) / 6, ci = (v - 27) % 6;
      atomicAdd(d.Sarrow + (size_t)(camrow + j) * d.ld + 6 * fr + ci, s);
    }
  }
#ifdef LIFCAL_STAMPS
  FSTAMP(11);
  if (lane == 0 && d.dbg) for (int i = 0; i < 16; ++i) d.dbg[((size_t)wg * 4 + role) * 16 + i] = st_acc[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// k_back4: point elimination of one block of regular points (ceres SchurEliminator::Eliminate for the blocks k_front4 formed)
// ---------------------------------------------------------------------------------------------------------------------
template <int NC, int TPT>
__global__ __launch_bounds__(B4_THREADS) void k_back4(Dev d, double radius) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const V4Back lay(d.v2_nfmax);
  const uint32_t tid = threadIdx.x, b = blockIdx.x;
  const uint32_t flo = d.blk_flo[b], nf = d.blk_nf[b];
  const uint32_t ncol = 6 * nf + NC + 1, zs = lay.zs, ncolp = (ncol + 15u) & ~15u;
  const uint32_t nwc = ncol - 1;
  double* Zd = sm;                                   // [B4_ZR][zs]
  double* Li = sm + lay.off_li;                      // [B4_BATCH][B4_LI]: L^-1 (6) | L^-1 g (3)
  double* misc = sm + lay.off_misc;                  // [0] bad-U count  [1] max |g_p| (as bits)
  uint32_t* pidl = (uint32_t*)(sm + lay.off_pid);    // point ids of the batch
  unsigned short* colinfo = (unsigned short*)(sm + lay.off_col);
  const uint32_t tg = tid >> 8, tt = tid & 255u;     // K group (rows 4 tg + 16 q .. + 3), tile slot
  const uint32_t nmt = (ncol + 3u) >> 2, ntri = nmt * (nmt + 1) / 2;
  auto tri_decode = [](uint32_t t, uint32_t& mi, uint32_t& mj) {
    mi = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (mi * (mi + 1) / 2 > t) --mi;
    while ((mi + 1) * (mi + 2) / 2 <= t) ++mi;
    mj = t - mi * (mi + 1) / 2;
  };
  uint32_t tmi[TPT], tmj[TPT];
  double tacc[TPT][4][4];
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    const uint32_t t = tt + 256u * q;
    tmi[q] = 0; tmj[q] = 0;
    if (t < ntri) tri_decode(t, tmi[q], tmj[q]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) tacc[q][i][j] = 0.0;
  }
  for (uint32_t cI = tid; cI < ncolp; cI += B4_THREADS)
    colinfo[cI] = (cI < 6 * nf) ? (unsigned short)(((cI / 6) << 8) | (cI % 6)) : (cI < ncol - 1 ? (unsigned short)(0x8000u | (cI - 6 * nf)) : (unsigned short)0xC000u);
  if (tid < 2) misc[tid] = 0.0;
  const uint32_t pb = d.blk_pt0[b], pe = d.blk_pt0[b + 1];
  for (uint32_t b0 = pb; b0 < pe; b0 += B4_BATCH) {
    const uint32_t nb = (pe - b0 < B4_BATCH) ? pe - b0 : B4_BATCH;
    __syncthreads();   // (the previous batch's chunks are done with Li; also orders the initialisation above)
    // ---- one thread per point of the batch: damp, factor U = L L^T (ceres LevenbergMarquardtStrategy + InvertPSDMatrix<3>) ----
    if (tid < nb) {
      const uint32_t p = d.v2_points[b0 + tid];
      pidl[tid] = p;
      const double* acc = d.ptacc + (size_t)p * 36;
      double U0 = acc[0], U1 = acc[1], U2 = acc[2], U3 = acc[3], U4 = acc[4], U5 = acc[5];
      const double g0 = acc[6], g1 = acc[7], g2 = acc[8];
      double lam[3];
      {
        const double h[3] = {U0, U3, U5};
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double sg = d.sigP[3 * (size_t)p + k]; lam[k] = fmin(fmax(h[k] * sg * sg, d.lm_min), d.lm_max) / (lm_radius(d, radius) * sg * sg); }
      }
      U0 += lam[0]; U3 += lam[1]; U5 += lam[2];
      bool ok = U0 > 0.0;
      double i00 = rsqrt(U0);
      const double l10 = U1 * i00, l20 = U2 * i00;
      const double d11 = U3 - l10 * l10; ok = ok && (d11 > 0.0);
      double i11 = rsqrt(d11);
      const double l21 = (U4 - l20 * l10) * i11;
      const double d22 = U5 - l20 * l20 - l21 * l21; ok = ok && (d22 > 0.0);
      double i22 = rsqrt(d22);
      double m10 = -l10 * i00 * i11, m21 = -l21 * i11 * i22, m20 = -(l20 * i00 + l21 * m10) * i22;
      if (!ok) { i00 = i11 = i22 = m10 = m21 = m20 = 0.0; atomicAdd(misc + 0, 1.0); }
      double* gu = d.Uinv + 9 * (size_t)p;
      const double v00 = i00 * i00 + m10 * m10 + m20 * m20, v01 = m10 * i11 + m20 * m21, v02 = m20 * i22;
      const double v11 = i11 * i11 + m21 * m21, v12 = m21 * i22, v22 = i22 * i22;
      gu[0] = v00; gu[1] = v01; gu[2] = v02; gu[3] = v01; gu[4] = v11; gu[5] = v12; gu[6] = v02; gu[7] = v12; gu[8] = v22;
      double* gl = d.lamP + 3 * (size_t)p; gl[0] = lam[0]; gl[1] = lam[1]; gl[2] = lam[2];
      const double gm = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
      atomicMax((unsigned long long*)(misc + 1), (unsigned long long)__double_as_longlong(gm));
      double* li = Li + tid * B4_LI;
      li[0] = i00; li[1] = m10; li[2] = m20; li[3] = i11; li[4] = m21; li[5] = i22;
      li[6] = i00 * g0; li[7] = m10 * g0 + i11 * g1; li[8] = m20 * g0 + m21 * g1 + i22 * g2;
    }
    for (uint32_t c0 = 0; c0 < nb; c0 += B4_CP) {
      const uint32_t np = (nb - c0 < B4_CP) ? nb - c0 : B4_CP;
      const uint32_t krows = (3 * np + 15u) & ~15u;
      __syncthreads();                                                                                    // ---- C0: Li of the batch; the previous chunk's product is done with Zd
      { double2* z2 = reinterpret_cast<double2*>(Zd); for (uint32_t i = tid; i < (krows * zs) / 2; i += B4_THREADS) z2[i] = double2{0.0, 0.0}; }
      __syncthreads();                                                                                    // ---- C1
      // ---- W rows of the chunk: 32 sub-threads per point walk its lanes; W_pose = R^T A [Gr | I] (k_backsub's formulas) ----
      {
        const uint32_t lp = tid >> 5, sub = tid & 31u;
        if (lp < np) {
          const uint32_t p = pidl[c0 + lp];
          const uint32_t s0 = d.pt_slot0[p], ns = d.pt_nslots[p];
          const double P0 = d.pts[3 * (size_t)p], P1 = d.pts[3 * (size_t)p + 1], P2 = d.pts[3 * (size_t)p + 2];
          for (uint32_t k = sub; k < ns; k += 32) {
            const uint32_t sidx = s0 + k, f = d.gid_fr[sidx];
            if (!d.frame_live[f]) continue;   // constant pose: no pose columns
            const double* ft = d.ft + (size_t)f * FRAME_STRIDE;
            const double* Ap = d.Av + (size_t)sidx * 6;
            double R[9], A[6];
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = ft[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) A[i] = Ap[i];
            const double c0r = ft[12], sn0 = ft[13];
            const double Y0 = R[0] * P0 + R[1] * P1 + R[2] * P2, Y1 = R[3] * P0 + R[4] * P1 + R[5] * P2, Y2 = R[6] * P0 + R[7] * P1 + R[8] * P2;
            const double n0 = R[2], n1 = R[5], n2 = R[8];
            const double Gr[3][3] = {{0.0, c0r * Y2 - sn0 * Y1, n1 * Y2 - n2 * Y1}, {-Y2, sn0 * Y0, n2 * Y0 - n0 * Y2}, {Y1, -c0r * Y0, n0 * Y1 - n1 * Y0}};
            const double Am[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
            double AG[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j) AG[i][j] = Am[i][0] * Gr[0][j] + Am[i][1] * Gr[1][j] + Am[i][2] * Gr[2][j];
            const uint32_t lf = f - flo;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              double* zrow = Zd + (size_t)(3 * lp + i) * zs + 6 * lf;
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                atomicAdd(zrow + j, R[i] * AG[0][j] + R[3 + i] * AG[1][j] + R[6 + i] * AG[2][j]);       // (atomic: the lanes of a split group share the cell)
                atomicAdd(zrow + 3 + j, R[i] * Am[0][j] + R[3 + i] * Am[1][j] + R[6 + i] * Am[2][j]);
              }
            }
          }
          // camera part of W and the rhs column L^-1 g
          const double* wc = d.ptacc + (size_t)p * 36 + 9;
          for (uint32_t e = sub; e < 3u * NC; e += 32) { const uint32_t i = e / NC, j = e - i * NC; Zd[(size_t)(3 * lp + i) * zs + 6 * nf + j] = wc[i * NCMAX + j]; }
          if (sub < 3) Zd[(size_t)(3 * lp + sub) * zs + (ncol - 1)] = Li[(c0 + lp) * B4_LI + 6 + sub];
        }
      }
      __syncthreads();                                                                                    // ---- C2
      // ---- Z = L^-1 W in place (pose + camera columns) ----
      for (uint32_t i = tid; i < np * nwc; i += B4_THREADS) {
        const uint32_t lp = i / nwc, cI = i - lp * nwc;
        const double* li = Li + (c0 + lp) * B4_LI;
        double* z = Zd + (size_t)(3 * lp) * zs + cI;
        const double w0 = z[0], w1 = z[zs], w2 = z[2 * zs];
        z[0] = li[0] * w0; z[zs] = li[1] * w0 + li[3] * w1; z[2 * zs] = li[2] * w0 + li[4] * w1 + li[5] * w2;
      }
      __syncthreads();                                                                                    // ---- C3
      // ---- window -= Z^T Z: 4x4 register tiles over the lower triangle, K split over the four thread groups ----
#pragma unroll
      for (int q = 0; q < TPT; ++q) {
        if (tt + 256u * q < ntri) {
          const double* za = Zd + 4 * tmi[q];
          const double* zb = Zd + 4 * tmj[q];
          for (uint32_t k0 = 4 * tg; k0 < krows; k0 += 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double2 a01 = *reinterpret_cast<const double2*>(za + (size_t)(k0 + r) * zs), a23 = *reinterpret_cast<const double2*>(za + (size_t)(k0 + r) * zs + 2);
              const double2 b01 = *reinterpret_cast<const double2*>(zb + (size_t)(k0 + r) * zs), b23 = *reinterpret_cast<const double2*>(zb + (size_t)(k0 + r) * zs + 2);
              const double av[4] = {a01.x, a01.y, a23.x, a23.y};
              const double bq[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) tacc[q][i][j] += av[i] * bq[j];
            }
          }
        }
      }
    }
  }
  // ---- the four K groups' partial tiles meet in group 0 (through the Z region), which adds them to the reduced system ----
  double* red = Zd;   // [TPT * 16][256]
  for (uint32_t gsrc = 1; gsrc < 4; ++gsrc) {
    __syncthreads();
    if (tg == gsrc) {
#pragma unroll
      for (int q = 0; q < TPT; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) red[(size_t)((q * 16 + i * 4 + j)) * 256 + tt] = tacc[q][i][j];
    }
    __syncthreads();
    if (tg == 0) {
#pragma unroll
      for (int q = 0; q < TPT; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) tacc[q][i][j] += red[(size_t)((q * 16 + i * 4 + j)) * 256 + tt];
    }
  }
  if (tg == 0) {
    const uint32_t camrow = 3 * d.Q, camcol = 6 * d.F + 3 * d.Q;
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      if (tt + 256u * q >= ntri) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t ci = 4 * tmi[q] + i;
        if (ci >= ncol) continue;
        const uint32_t ii = colinfo[ci];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t cj = 4 * tmj[q] + j;
          const double dv = tacc[q][i][j];
          if (cj >= ncol || ci < cj || dv == 0.0) continue;
          const uint32_t jj = colinfo[cj];
          if (!(ii & 0x8000u)) {            // pose x pose
            const uint32_t lfi = ii >> 8, lfj = jj >> 8, dd = lfi - lfj;
            if (dd <= d.bw) atomicAdd(d.Sband + ((size_t)(flo + lfi) * (d.bw + 1) + dd) * 36 + (ii & 0xFFu) * 6 + (jj & 0xFFu), -dv);
          } else if (!(ii & 0x4000u)) {     // camera row
            const uint32_t jc = ii & 0xFFu;
            if (!(jj & 0x8000u)) atomicAdd(d.Sarrow + (size_t)(camrow + jc) * d.ld + 6 * flo + cj, -dv);
            else atomicAdd(d.Sarrow + (size_t)(camrow + jc) * d.ld + camcol + (jj & 0xFFu), -dv);
          } else if (cj < ncol - 1) {       // rhs row: W^T U^-1 g
            if (!(jj & 0x8000u)) atomicAdd(d.rhsacc + 6 * flo + cj, dv); else atomicAdd(d.rhsacc + camcol + (jj & 0xFFu), dv);
          }
        }
      }
    }
  }
  if (tid == 0) {
    if (misc[0] != 0.0) atomicAdd(d.scal + SCAL_BAD_U, misc[0]);
    atomicMax((unsigned long long*)(d.scal + SCAL_GMAX0 + d.rank), *(unsigned long long*)(misc + 1));
  }
}

}  // namespace lifcal
