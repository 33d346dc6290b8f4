// sweep3.hpp — k_sweep3: k_sweep2's fused Jacobian + Schur sweep with WAVE-SPECIALISED observation loop (included by kernels.hpp).
//
// Why: in k_sweep2 one lane carries every accumulator of its observations — A, b, C (3x9) of the group AND the 45 + 9
// camera x camera / camera-gradient sums of the lane — 476 live registers, one wave per SIMD, ~250 AGPR<->VGPR moves per
// observation.  Here a workgroup has 512 threads = 8 waves, two per SIMD, in two roles over the SAME 256 lanes of a pass:
//   waves 0-3 ("evaluators")    walk the observations: residual, Jq, Jc, weight; keep A and b; hand Jq, Jc, r (26 doubles per
//                               lane and step) to LDS — the dense Z matrix is idle during the observation loop, the hand-off
//                               buffer lives in its space;
//   waves 4-7 ("accumulators")  read them back and keep C, the camera x camera block and the camera gradient.
// Each role fits 256 registers (no spills), the SIMD interleaves the two instruction streams, and every other phase has twice
// the threads.  The hand-off is synchronised PER WAVE PAIR through two LDS counters (steps written / steps read): an evaluator
// only waits for its accumulator to have read the previous step, an accumulator for its evaluator's next step — no workgroup
// barrier inside the observation loop, the pair runs one step apart.
// Everything else — passes, LDS layout, emission targets, factor, Z, Schur product, flush — is k_sweep2's (sweep2.hpp).
// The two roles are two separate code paths on purpose: register allocation is static, a shared path would make every wave
// carry both roles' accumulators.  Both paths execute the same sequence of workgroup barriers.
#pragma once

namespace lifcal {

// WR = waves per role: 4 (512 threads, passes of 256 lanes, one workgroup per CU) or 2 (256 threads, passes of 128 lanes, an LDS
// window under 80 KiB: TWO workgroups per CU, out of step with each other, so that one's latency-bound phases — pass top,
// the single-wave factor phase, LDS-atomic emission — run under the other's observation loop).
// ET = evaluation type: double, or float for options.precision = 1 (residual and Jacobian of an observation in fp32 from fp32
// observation words and the fp32 lens table, hand-off in fp32; every accumulator and everything behind the loop stays fp64).
template <int NR, bool TAN, bool ADJ, int WR, class ET = double>
__global__ __launch_bounds__(128 * WR, 2) void k_sweep3(Dev d, double radius, int mode) {
  constexpr bool F32 = sizeof(ET) == 4;
  constexpr int NC = 5 + NR + (TAN ? 2 : 0);
  constexpr int NCC = NC * (NC + 1) / 2;
  constexpr int HV = 6 + 2 * NC + 2;   // doubles handed over per lane and step: Jq (6) | Jc (2 NC) | r (2)
  constexpr uint32_t LP = 64u * WR;    // lanes of a pass (= threads per role)
  constexpr uint32_t NT = 2u * LP;     // threads of the workgroup
  static_assert(HV * LP <= V2Lds::zd_for(LP), "hand-off buffer must fit the Z matrix");
  // device LM loop (radius < 0: the state lives in HBM): the sweep the host enqueued behind the TERMINATING decision has nothing to do
  if (radius < 0.0 && d.lm[LM_TERMINATION] != 0.0) return;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const V2Lds lay(d.v2_nfmax, true, LP);
  const uint32_t NFm = lay.nfm, vlen = 6 * NFm + NCMAX + 3;
  double* Spp = sm; double* Scp = sm + lay.off_cp; double* Scc = sm + lay.off_cc;
  double* vgB = sm + lay.off_vec; double* vhd = vgB + vlen; double* vrhs = vhd + vlen;
  double* Fr = sm + lay.off_fr;       // [value][frame][replica] replicated frame-level accumulators
  double* slab = sm + lay.off_slab;   // per point of the pass: [0..5] U -> L^-1, [6..8] g
  double* Zd = sm + lay.off_zd;       // dense Z matrix of the pass; during the observation loop: hand-off buffer [HV][256]
  double* misc = sm + lay.off_misc;   // [0] cost, [1] bad-U count, [2] max |g_p| (as bits)
  uint32_t* pidl = (uint32_t*)(misc + MISC_DOUBLES);
  unsigned short* colinfo = (unsigned short*)(misc + MISC_DOUBLES + 32);
  // WR == 2: the four waves of a workgroup sit on the four SIMDs of the CU, one each, so a workgroup alone would put both evaluator
  // waves (300 VALU instructions per step) on SIMDs 0-1 and both accumulators (200) on SIMDs 2-3.  The second workgroup of the CU
  // (its LDS allocation does not start at 0) swaps the roles of its wave pairs, so that every SIMD hosts one evaluator and one
  // accumulator, as with four waves per role.  (LIFCAL_NO_ROLE_FLIP in the build disables it: A/B measurement.)
  uint32_t flip = 0;
#ifndef LIFCAL_NO_ROLE_FLIP
  if (WR == 2) flip = (__builtin_amdgcn_s_getreg(6 | (0 << 6) | (7 << 11)) != 0) ? LP : 0u;   // HW_REG_LDS_ALLOC.LDS_BASE
#endif
  const uint32_t tid = threadIdx.x ^ flip, lane = tid & 63u, w = tid >> 6;
  const bool role_b = w >= (uint32_t)WR;  // accumulator waves
  const uint32_t wl = w & (uint32_t)(WR - 1);   // tile of the pass this wave works on
  const uint32_t t256 = tid & (LP - 1u);  // lane slot of the pass (the same for the evaluator and its accumulator)
  const uint32_t b = blockIdx.x;
  const uint32_t flo = d.blk_flo[b], nf = d.blk_nf[b];
  const uint32_t ncol = 6 * nf + NC + 1, ncolp = (ncol + 15u) & ~15u;
  const uint32_t zs = ncolp + 2;
  const CamConsts c = *d.camc;
#ifdef LIFCAL_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
  if (tid == 0 || tid == LP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); }
#define STAMPB(i) do { if (tid == LP) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[i] += t_ - st_last; st_last = t_; } } while (0)
#else
#define STAMPB(i) do { } while (0)
#endif
  // pass descriptors and per-lane slot words, one pass ahead (see k_sweep2)
  const uint32_t ps_begin = d.blk_pass0[b], ps_end = d.blk_pass0[b + 1];
  uint32_t nx_np = 0, nx_gid0 = 0, nx_si = 0, nx_pt = 0, nx_fp = 0, nx_gid = 0, nx_r[2] = {0, 0};
  uint32_t vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  auto fetch_pass = [&](uint32_t q) {
    nx_np = d.pass_np[q + vz]; nx_gid0 = d.pass_gid0[q + vz];
    nx_si = d.v2_slot[(size_t)q * LP + t256]; nx_pt = d.v2f_pt[(size_t)q * LP + t256]; nx_gid = d.v2_gidx[(size_t)q * LP + t256];
    nx_r[0] = d.v2_tile_row0[q * WR + wl + vz]; nx_r[1] = d.v2_tile_row0[q * WR + wl + 1 + vz];   // observation rows of the wave's tile
    nx_fp = d.v2_passpt[(size_t)q * 64 + (tid & 63u)];
  };
  if (ps_begin < ps_end) fetch_pass(ps_begin);
  { double2* z2 = reinterpret_cast<double2*>(sm); for (uint32_t i = tid; i < lay.off_slab / 2; i += NT) z2[i] = double2{0.0, 0.0}; }
  if (tid == 0 && (lay.off_slab & 1u)) sm[lay.off_slab - 1] = 0.0;
  if (tid < MISC_DOUBLES) misc[tid] = 0.0;
  for (uint32_t cI = tid; cI < ncolp; cI += NT)
    colinfo[cI] = (cI < 6 * nf) ? (unsigned short)(((cI / 6) << 8) | (cI % 6)) : (cI < ncol - 1 ? (unsigned short)(0x8000u | (cI - 6 * nf)) : (unsigned short)0xC000u);

  // ---- phases both roles run with all 512 threads ----
  auto zero_slab = [&]() { for (uint32_t i = tid; i < lay.np_max * SLAB_STRIDE; i += NT) slab[i] = 0.0; };
  auto zero_zd = [&](uint32_t krows) { double2* z2 = reinterpret_cast<double2*>(Zd); for (uint32_t i = tid; i < (krows * zs) / 2; i += NT) z2[i] = double2{0.0, 0.0}; };
  // Z = L^-1 W in place (pose + camera columns), camera part of W to HBM first; thread = (column, point phase)
  auto z_phase = [&](uint32_t np) {
    const uint32_t nwc = ncol - 1;
    const uint32_t nth = NT / nwc > 0 ? NT / nwc : 1;
    const uint32_t gi = tid / nwc, cidx = tid - gi * nwc;
    if (gi < nth || nwc > NT) {
      for (uint32_t cc0 = cidx; cc0 < nwc; cc0 += (nwc > NT ? NT : nwc * nth)) {
#pragma unroll 4
        for (uint32_t lp = (nwc > NT ? 0 : gi); lp < np; lp += (nwc > NT ? 1 : nth)) {
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
  };
  // rotation, rotated point and d(R P)/d(angles) of a lane's (point, frame)
  struct LaneGeom { double R[9], Y[3], Gr[3][3]; };
  auto lane_geom = [&](uint32_t fr, uint32_t pt, LaneGeom& q) {
    const double* ft = d.ft + (size_t)fr * FRAME_STRIDE;
    const double* P = d.pts + 3 * (size_t)pt;
#pragma unroll
    for (int k = 0; k < 9; ++k) q.R[k] = ft[k];
    const double P0 = P[0], P1 = P[1], P2 = P[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) q.Y[k] = q.R[3 * k] * P0 + q.R[3 * k + 1] * P1 + q.R[3 * k + 2] * P2;
    const double c0 = ft[12], s0 = ft[13], n0 = q.R[2], n1 = q.R[5], n2 = q.R[8];
    // Gr = [e_x x Y, (0,c0,s0) x Y, R[:,2] x Y]
    q.Gr[0][0] = 0.0;     q.Gr[0][1] = c0 * q.Y[2] - s0 * q.Y[1]; q.Gr[0][2] = n1 * q.Y[2] - n2 * q.Y[1];
    q.Gr[1][0] = -q.Y[2]; q.Gr[1][1] = s0 * q.Y[0];               q.Gr[1][2] = n2 * q.Y[0] - n0 * q.Y[2];
    q.Gr[2][0] = q.Y[1];  q.Gr[2][1] = -c0 * q.Y[0];              q.Gr[2][2] = n0 * q.Y[1] - n1 * q.Y[0];
  };
  const uint32_t frs = NFm * lay.nrep;   // stride between the values of the frame accumulators (replica 0 is the only one used here)
  // The lanes of a pass are sorted by frame (plan.hpp, frame order): the lanes of one frame form runs inside each row of 16
  // lanes.  Frame-level values are summed over a run with four masked DPP row shifts (lane i += lane i+n if that lane
  // belongs to the same frame), and only the first lane of a run adds the total to LDS: ~7 lanes per wave instruction on
  // distinct addresses instead of 64 lanes with bank and same-address conflicts.
  struct RunMask { double m1, m2, m4, m8; bool head; };
  auto run_masks = [&](uint32_t cnt, uint32_t lf) {
    const int key = cnt > 0 ? (int)lf : (int)(0x7F000000u + lane);   // idle lanes never match
    RunMask r;
    r.m1 = __builtin_amdgcn_update_dpp(-1, key, 0x101, 0xf, 0xf, false) == key ? 1.0 : 0.0;   // row_shl:1 (lane i reads lane i+1 of its row)
    r.m2 = __builtin_amdgcn_update_dpp(-1, key, 0x102, 0xf, 0xf, false) == key ? 1.0 : 0.0;
    r.m4 = __builtin_amdgcn_update_dpp(-1, key, 0x104, 0xf, 0xf, false) == key ? 1.0 : 0.0;
    r.m8 = __builtin_amdgcn_update_dpp(-1, key, 0x108, 0xf, 0xf, false) == key ? 1.0 : 0.0;
    r.head = cnt > 0 && __builtin_amdgcn_update_dpp(-1, key, 0x111, 0xf, 0xf, false) != key;   // row_shr:1: the previous lane of the row
    return r;
  };
  auto run_sum = [](double v, const RunMask& r) {
    auto shl = [](double x, auto ctrl) {
      const int lo = __double2loint(x), hi = __double2hiint(x);
      return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, decltype(ctrl)::value, 0xf, 0xf, true),
                              __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v = fma(shl(v, std::integral_constant<int, 0x101>{}), r.m1, v);
    v = fma(shl(v, std::integral_constant<int, 0x102>{}), r.m2, v);
    v = fma(shl(v, std::integral_constant<int, 0x104>{}), r.m4, v);
    v = fma(shl(v, std::integral_constant<int, 0x108>{}), r.m8, v);
    return v;
  };
  // hand-off counters of the wave pair (evaluator wl, accumulator wl + 4): monotone over the whole kernel, zeroed with misc
  uint32_t* hw_written = (uint32_t*)(misc + 4) + wl;       // steps the evaluator has published
  uint32_t* hw_read = (uint32_t*)(misc + 4) + 4 + wl;      // steps the accumulator has taken
  uint32_t* eval_done = (uint32_t*)(misc + 3);             // evaluator waves that have finished the emission of a pass (monotone)
  // options.deterministic: LDS atomics of the emission are issued in WAVE ORDER (one turn counter per role, monotone over the
  // kernel), so every LDS accumulator receives its addends in the same order on every run; inside one wave instruction the LDS
  // unit serialises lanes that hit one address in lane order
  const bool det = d.deterministic != 0;
  uint32_t* turn = (uint32_t*)(misc + 8) + (role_b ? 1 : 0);
  uint32_t my_turn = wl;                                    // + WR per pass
  // acquire / release at workgroup scope: the consumer's reads of the hand-off buffer may not be hoisted above the wait, the
  // producer's writes may not sink below the publish (on gfx950 both order LDS through lgkmcnt; no cache maintenance is involved)
  auto wait_for = [](uint32_t* flag, uint32_t need) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(2);
  };
  auto publish = [](uint32_t* flag, uint32_t value) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's LDS traffic of this step is done before the counter moves
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  // Schur product: 4x4 micro-tiles over the lower triangle, software-pipelined.  With one tile per thread (ntri <= 256) the
  // evaluator thread keeps its tile in registers across the passes, and the K dimension is split between the roles: evaluator
  // thread t takes rows 0-3 of every 8, accumulator thread 256 + t rows 4-7, its partial tile living in LDS between passes
  // (the accumulators have no registers to spare; the single set of frame accumulators leaves the room).
  const uint32_t nmt = (ncol + 3u) >> 2, ntri = nmt * (nmt + 1) / 2;
  auto tri_decode = [](uint32_t t, uint32_t& mi, uint32_t& mj) {
    mi = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (mi * (mi + 1) / 2 > t) --mi;
    while ((mi + 1) * (mi + 2) / 2 <= t) ++mi;
    mj = t - mi * (mi + 1) / 2;
  };
  auto gemm_tile = [&](uint32_t mi, uint32_t mj, uint32_t kfirst, uint32_t kstep, uint32_t krows, double (&acc16)[4][4]) {
    const double* za = Zd + 4 * mi;
    const double* zb = Zd + 4 * mj;
    double2 cur[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cur[r][0] = *reinterpret_cast<const double2*>(za + (size_t)(kfirst + r) * zs);
      cur[r][1] = *reinterpret_cast<const double2*>(za + (size_t)(kfirst + r) * zs + 2);
      cur[r][2] = *reinterpret_cast<const double2*>(zb + (size_t)(kfirst + r) * zs);
      cur[r][3] = *reinterpret_cast<const double2*>(zb + (size_t)(kfirst + r) * zs + 2);
    }
#pragma unroll 2
    for (uint32_t k0 = kfirst; k0 < krows; k0 += kstep) {
      double2 nxt[4][4];
      const uint32_t kn = (k0 + kstep < krows) ? k0 + kstep : k0;
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
  // WR == 2: 128 threads per role cannot own the ~230 tiles of a 12-frame window one each; all 256 threads share the tiles of
  // every pass (fresh accumulators per pass, Spp -= tile at its end)
  const bool keep_tiles = (WR == 4) && (mode == 0) && ntri <= LP;
  const bool ksplit = keep_tiles && lay.has_bt();
  const bool share_tiles = (WR != 4) && (mode == 0);
  double* bt = sm + lay.off_bt;   // [16][256] partial tiles of the accumulator threads
  if (!role_b) {
    // =====================================================================================================================
    // evaluator waves (threads 0..255)
    // =====================================================================================================================
    double cost = 0.0, lmant = 1.0; int lexp = 0;
    uint32_t hbase = 0;   // steps of the passes before this one (hand-off counters are monotone)
    uint32_t passes_done = 0;
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
      const uint32_t si = nx_si, pt = nx_pt, fp = nx_fp, gidx = nx_gid;
      const uint32_t row0 = __builtin_amdgcn_readfirstlane(nx_r[0]), kmax = __builtin_amdgcn_readfirstlane(nx_r[1]) - row0;
      double fsg0 = 1.0, fsg1 = 1.0, fsg2 = 1.0;
      if (mode == 0 && tid < np) { fsg0 = d.sigP[3 * (size_t)fp]; fsg1 = d.sigP[3 * (size_t)fp + 1]; fsg2 = d.sigP[3 * (size_t)fp + 2]; }
      if (ps + 1 < ps_end) fetch_pass(ps + 1);
      const uint32_t krows = (3 * np + 7u) & ~7u;   // K of the product: multiple of 8 (two row groups of 4, one per role)
      const uint32_t cnt = si & 0xFFu, lf = (si >> 8) & 0xFFu, lp = (si >> 16) & 0xFFu;
      const uint32_t fr = flo + lf;
      // observation words of the evaluation type: (u, v) in fp64, (u - mcx, v - mcy) in fp32; lens rows of 16 ET
      const ET* obs_u = F32 ? reinterpret_cast<const ET*>(d.v2_du) : reinterpret_cast<const ET*>(d.v2_u);
      const ET* obs_v = F32 ? reinterpret_cast<const ET*>(d.v2_dv) : reinterpret_cast<const ET*>(d.v2_v);
      const ET* lens_tab = F32 ? reinterpret_cast<const ET*>(d.ltf) : reinterpret_cast<const ET*>(d.lt);
      // The pass's first dependent loads — frame row and point of the lane, the observation words of steps 0 and 1 — go out
      // HERE, in front of the slab zero-fill and barrier P1 (an asm with a memory clobber: the compiler cannot sink them below
      // it), so that their round trips run under the wait for the other waves instead of behind it.
      double ftv[12], P0, P1, P2;
      {
        const double* ft = d.ft + (size_t)fr * FRAME_STRIDE;
        const double* P = d.pts + 3 * (size_t)pt;
#pragma unroll
        for (int i = 0; i < 12; ++i) ftv[i] = ft[i];
        P0 = P[0]; P1 = P[1]; P2 = P[2];
      }
      ET ua = 0, va = 0, ub = 0, vb = 0; uint32_t lb = 0, la = 0, l0 = 0;
      {
        const size_t at = (size_t)row0 * 64 + lane;                                        // (row0 is a valid row even for kmax == 0: the ELL arrays are padded)
        const size_t at2 = ((size_t)row0 + (kmax > 1 ? 1u : 0u)) * 64 + lane;
        ua = obs_u[at]; va = obs_v[at]; l0 = d.v2_lens[at];
        ub = obs_u[at2]; vb = obs_v[at2]; lb = d.v2_lens[at2];
      }
      STAMP(12);
      zero_slab();
      STAMP(13);
      lds_barrier();                                                                                    // ---- barrier P1
      STAMP(0);
      typename std::conditional<F32, GroupConsts2F, GroupConsts2>::type gcn;
      {
        const double Xc = ftv[0] * P0 + ftv[1] * P1 + ftv[2] * P2 + ftv[9], Yc = ftv[3] * P0 + ftv[4] * P1 + ftv[5] * P2 + ftv[10], Zc = ftv[6] * P0 + ftv[7] * P1 + ftv[8] * P2 + ftv[11];
        if constexpr (F32) group_prepare2f<ADJ>(c, Xc, Yc, Zc, gcn); else group_prepare2<ADJ>(c, Xc, Yc, Zc, gcn);
      }
      double A[6] = {0, 0, 0, 0, 0, 0}, bv[3] = {0, 0, 0};
      ET* hand = reinterpret_cast<ET*>(Zd) + t256;
      constexpr uint32_t HS = LP;   // stride between the values of the hand-off buffer
      // two-stage prefetch: observation words (u, v, lens index) two steps ahead, the 128-byte lens row one step ahead; the
      // row's registers are free at the end of a step (the evaluation consumes the row first), so the peak does not grow.
      // All prefetch loads are UNCONDITIONAL (row index clamped; the ELL padding is valid memory): with loads under a lane
      // condition the compiler's wait-count merge degenerates to vmcnt(0) at the top of the evaluation, i.e. the step would
      // wait for the prefetch it has just issued (HBM latency, every step).
      // The loop is unrolled by two over two sets of observation words (even / odd steps): a rotating set would make the
      // compiler copy freshly loaded registers at the back-edge, i.e. wait for the loads it has just issued.
      ET Ln[LENS_STRIDE];
      double wn0 = 0.0, wn1 = 0.0;   // F32: w of the lens in fp64 (side table), prefetched with the row
      {
        const ET* L = lens_tab + (size_t)l0 * LENS_STRIDE;                                   // lens 0 for idle rows: valid memory, never used
#pragma unroll
        for (int i = 0; i < LENS_STRIDE; ++i) Ln[i] = L[i];
        if constexpr (F32) { wn0 = d.ltw[2 * (size_t)l0]; wn1 = d.ltw[2 * (size_t)l0 + 1]; }
      }
      typename std::conditional<F32, CamF, int>::type cf{};
      if constexpr (F32) cf = cam_to_float(c);
      // one step: evaluate with (uc, vc, Ln), then fetch the lens row of step k+1 (index ln, arrived a step ago) and the
      // words of step k+2 into the set just consumed
      auto step = [&](uint32_t k, ET& uc, ET& vc, uint32_t& lc, const uint32_t ln) {
        // the hand-off buffer is free once the accumulator has taken step k-1 (it did so long ago: checked first, so that
        // the evaluation's results can go to LDS as they are produced instead of staying live)
        wait_for(hw_read, hbase + k);
        if (k < cnt) {
          const ET u = uc, v = vc;
          ET L[LENS_STRIDE];
#pragma unroll
          for (int i = 0; i < LENS_STRIDE; ++i) L[i] = Ln[i];
          ET r[2], Jq[2][3], Jc[2][NC];
          ET arg;
          if constexpr (F32) obs_eval2f_pk<NR, TAN, ADJ>(cf, gcn, L, wn0, wn1, u, v, d.robust != 0, r, Jq, Jc, arg);   // packed fp32 (obs_eval2f: the scalar form)
          else obs_eval2<NR, TAN, ADJ>(c, gcn, L, u, v, d.robust != 0, r, Jq, Jc, arg);
          // hand-off to the accumulator wave, [value][lane]
#pragma unroll
          for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int j = 0; j < NC; ++j) hand[(6 + a * NC + j) * HS] = Jc[a][j];
#pragma unroll
            for (int i = 0; i < 3; ++i) hand[(a * 3 + i) * HS] = Jq[a][i];
            hand[(6 + 2 * NC + a) * HS] = r[a];
          }
          if (d.robust) { int ex; lmant = frexp(lmant * (double)arg, &ex); lexp += ex; }   // rho = b log(prod (1 + s/b)): one log per lane at the end
          else cost += 0.5 * (double)arg;
#pragma unroll
          for (int a = 0; a < 2; ++a) {   // (fp64 accumulation, whatever the evaluation type)
            const double q0 = Jq[a][0], q1 = Jq[a][1], q2 = Jq[a][2], ra = r[a];
            A[0] += q0 * q0; A[1] += q0 * q1; A[2] += q0 * q2;
            A[3] += q1 * q1; A[4] += q1 * q2; A[5] += q2 * q2;
            bv[0] += q0 * ra; bv[1] += q1 * ra; bv[2] += q2 * ra;
          }
        }
        {
          const ET* Lp = lens_tab + (size_t)ln * LENS_STRIDE;
#pragma unroll
          for (int i = 0; i < LENS_STRIDE; ++i) Ln[i] = Lp[i];
          if constexpr (F32) { wn0 = d.ltw[2 * (size_t)ln]; wn1 = d.ltw[2 * (size_t)ln + 1]; }
          const uint32_t kk = (k + 2 < kmax) ? k + 2 : kmax - 1;
          const size_t at = ((size_t)row0 + kk) * 64 + lane;
          uc = obs_u[at]; vc = obs_v[at]; lc = d.v2_lens[at];
        }
        publish(hw_written, hbase + k + 1);
      };
      for (uint32_t k = 0; k < kmax; k += 2) {
        step(k, ua, va, la, lb);
        if (k + 1 < kmax) step(k + 1, ub, vb, lb, la);
      }
      hbase += kmax;
      // rotation, rotated point, d(R P)/d(angles): read again here instead of being carried through the loop (42 registers)
      LaneGeom q;
      { uint32_t fr2 = fr, pt2 = pt; asm volatile("" : "+v"(fr2), "+v"(pt2)); lane_geom(fr2, pt2, q); }
      asm volatile("" :: "v"(nx_si), "v"(nx_pt), "v"(nx_fp), "v"(nx_np), "v"(nx_gid0), "v"(nx_gid), "v"(nx_r[0]), "v"(nx_r[1]));
      asm volatile("" :: "v"(fsg0), "v"(fsg1), "v"(fsg2));
      STAMP(7);
      lds_barrier();                                                                                    // ---- barrier P2a: every accumulator has taken its last step
      if (mode == 0) zero_zd(krows);
      lds_barrier();                                                                                    // ---- barrier P2
      if (det) wait_for(turn, my_turn);
      if (cnt > 0) {
        const double Am[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
        const double (&R)[9] = q.R; const double (&Gr)[3][3] = q.Gr;
        double AG[3][3], GAG[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) AG[i][j] = Am[i][0] * Gr[0][j] + Am[i][1] * Gr[1][j] + Am[i][2] * Gr[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) GAG[i][j] = Gr[0][i] * AG[0][j] + Gr[1][i] * AG[1][j] + Gr[2][i] * AG[2][j];
        RunMask rm = run_masks(cnt, lf);
        const bool fpose = d.frame_live[fr] != 0;   // 0: the pose of this frame is held constant (lifcal_ba_set_fixed_frames): no pose columns
        rm.head = rm.head && fpose;
        double* fr_acc = Fr + lf;
        {
          int vi = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int bb = 0; bb <= a; ++bb) {
              double v;
              if (a < 3) v = GAG[a][bb]; else if (bb < 3) v = AG[a - 3][bb]; else v = Am[a - 3][bb - 3];
              v = run_sum(v, rm);
              if (rm.head) atomicAdd(fr_acc + (size_t)vi * frs, v);
              ++vi;
            }
#pragma unroll
          for (int a = 0; a < 3; ++a) { const double v = run_sum(Gr[0][a] * bv[0] + Gr[1][a] * bv[1] + Gr[2][a] * bv[2], rm); if (rm.head) atomicAdd(fr_acc + (size_t)(21 + a) * frs, v); }
#pragma unroll
          for (int a = 0; a < 3; ++a) { const double v = run_sum(bv[a], rm); if (rm.head) atomicAdd(fr_acc + (size_t)(24 + a) * frs, v); }
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
          { double* ga = d.Av + (size_t)gidx * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) ga[k] = A[k]; }
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            atomicAdd(acc + 6 + i, R[i] * bv[0] + R[3 + i] * bv[1] + R[6 + i] * bv[2]);
            // The row stride of Z is even (16-byte rows for the Schur product's double2 reads) and so is the frame stride 6: with
            // the same column j in every lane an f64 atomic instruction reaches only the EVEN double-banks (4-way conflicts over
            // 64 lanes).  Lanes of odd points therefore take the six columns rotated by one: instruction jj adds column jj + 1.
            double* zrow = Zd + (size_t)(3 * lp + i) * zs + 6 * lf;
            double w6[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) { w6[j] = R[i] * AG[0][j] + R[3 + i] * AG[1][j] + R[6 + i] * AG[2][j]; w6[3 + j] = R[i] * Am[0][j] + R[3 + i] * Am[1][j] + R[6 + i] * Am[2][j]; }
            const bool rot = (lp & 1u) != 0;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
              const double wv = rot ? w6[(jj + 1) % 6] : w6[jj];
              const uint32_t col = rot ? (uint32_t)((jj + 1) % 6) : (uint32_t)jj;
              if (fpose) atomicAdd(zrow + col, wv);
            }
          }
        }
      }
      if (det) { publish(turn, my_turn + 1); my_turn += (uint32_t)WR; }
      STAMP(6);
      // The factor phase reads only what the EVALUATOR waves emitted (U, g of the points) and runs on wave 0 (np <= 64): it
      // waits for the four evaluator waves through an LDS counter instead of a workgroup barrier, so it overlaps the longer
      // emission of the accumulator waves (81 vs 54 LDS atomics per lane); everybody meets again at barrier P4.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(eval_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ++passes_done;
      if (w == 0) wait_for(eval_done, (uint32_t)WR * passes_done);
      STAMP(1);
      // ---- one thread per point: damp, factor U = L L^T (k_sweep2 phase 2) ----
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
          double* z0 = Zd + (size_t)(3 * tid) * zs + (ncol - 1);
          z0[0] = i00 * g0; z0[zs] = m10 * g0 + i11 * g1; z0[2 * zs] = m20 * g0 + m21 * g1 + i22 * g2;
        }
      }
      lds_barrier();                                                                                    // ---- barrier P4
      STAMP(2);
      if (mode == 0) {
        z_phase(np);
        lds_barrier();                                                                                  // ---- barrier P5
        STAMP(3);
        if (keep_tiles) {
          if (tid < ntri) gemm_tile(mi0, mj0, 0, ksplit ? 8u : 4u, krows, tacc);
        } else {
          for (uint32_t t = tid; t < ntri; t += (share_tiles ? NT : LP)) {
            uint32_t mi, mj; tri_decode(t, mi, mj);
            double acc16[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc16[i][j] = 0.0;
            gemm_tile(mi, mj, 0, 4, krows, acc16);
            emit_tile(mi, mj, acc16);
          }
        }
      }
      lds_barrier();                                                                                    // ---- barrier P6
      STAMP(4);
    }
    STAMP(8);
    if (keep_tiles && tid < ntri) {
      if (ksplit) {   // the accumulator thread's partial tile (complete: barrier P6 of the last pass)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) tacc[i][j] += bt[(i * 4 + j) * LP + tid];
      }
      // The tile goes into the window with all its LDS reads in flight together: the eight column descriptors, then the sixteen
      // old values, then the sixteen stores (entries outside the lower triangle / beyond the columns go to a scratch cell of the
      // thread) — entry by entry (emit_tile) it was sixteen dependent LDS round trips, 2 % of a block.
      uint32_t ci4[4], cj4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const uint32_t ci = 4 * mi0 + i, cj = 4 * mj0 + i; ci4[i] = colinfo[ci < ncolp ? ci : 0]; cj4[i] = colinfo[cj < ncolp ? cj : 0]; }
      uint32_t off[4][4]; double sg[4][4];
      const uint32_t dummy = lay.off_zd + 1024u + tid;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t ci = 4 * mi0 + i, cj = 4 * mj0 + j, ii = ci4[i], jj = cj4[j];
          uint32_t o = dummy; double sgn = -1.0;
          if (!(ci >= ncol || cj >= ncol || ci < cj)) {
            if (!(ii & 0x8000u)) {            // pose x pose
              const uint32_t lfi = ii >> 8, lfj = jj >> 8;
              o = (lfi * (lfi + 1) / 2 + lfj) * 36 + (ii & 0xFFu) * 6 + (jj & 0xFFu);
            } else if (!(ii & 0x4000u)) {     // camera row
              const uint32_t jc = ii & 0xFFu;
              o = !(jj & 0x8000u) ? lay.off_cp + jc * 6 * NFm + cj : lay.off_cc + jc * (jc + 1) / 2 + (jj & 0xFFu);
            } else if (cj < ncol - 1) {       // rhs row: W^T U^-1 g
              o = lay.off_vec + 2 * vlen + (!(jj & 0x8000u) ? cj : 6 * NFm + (jj & 0xFFu));
              sgn = 1.0;
            }
          }
          off[i][j] = o; sg[i][j] = sgn;
        }
      double oldv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) oldv[i][j] = sm[off[i][j]];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[off[i][j]] = oldv[i][j] + sg[i][j] * tacc[i][j];
    }
    STAMP(9);
    if (d.robust) cost += 0.5 * c.loss_b * (log(lmant) + (double)lexp * 0.6931471805599453);
    cost = wave_sum_dpp(cost);
    if (det) wait_for(turn, my_turn);
    if (lane == 63) atomicAdd(misc + 0, cost);
    if (det) publish(turn, my_turn + 1);
    // the accumulator waves reduce the camera block meanwhile: same barriers
    lds_barrier(); lds_barrier();                                                                      // ---- barriers T1, T2
  } else {
    // =====================================================================================================================
    // accumulator waves (threads 256..511): C of the lane, camera x camera block and camera gradient of the thread
    // =====================================================================================================================
    double cc[NCC], gc[NC];
    uint32_t hbase = 0;
#pragma unroll
    for (int i = 0; i < NCC; ++i) cc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) gc[i] = 0.0;
    for (uint32_t ps = ps_begin; ps < ps_end; ++ps) {
      const uint32_t np = __builtin_amdgcn_readfirstlane(nx_np);
      const uint32_t si = nx_si, pt = nx_pt;
      const uint32_t kmax = __builtin_amdgcn_readfirstlane(nx_r[1]) - __builtin_amdgcn_readfirstlane(nx_r[0]);
      if (ps + 1 < ps_end) fetch_pass(ps + 1);
      const uint32_t krows = (3 * np + 7u) & ~7u;   // K of the product: multiple of 8 (two row groups of 4, one per role)
      zero_slab();
      lds_barrier();                                                                                    // ---- barrier P1
      STAMPB(0);
      const uint32_t cnt = si & 0xFFu, lf = (si >> 8) & 0xFFu, lp = (si >> 16) & 0xFFu;
      double C[3][NC];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) C[i][j] = 0.0;
      const ET* hand = reinterpret_cast<const ET*>(Zd) + t256;
      constexpr uint32_t HS = LP;
      for (uint32_t k = 0; k < kmax; ++k) {
        wait_for(hw_written, hbase + k + 1);
        const bool live = k < cnt;
        // one residual row at a time (13 doubles live instead of 26); the step is released after the second row is read
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          double r = 0.0, Jq[3] = {0.0, 0.0, 0.0}, Jc[NC];
#pragma unroll
          for (int j = 0; j < NC; ++j) Jc[j] = 0.0;
          if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) Jq[i] = hand[(a * 3 + i) * HS];
#pragma unroll
            for (int j = 0; j < NC; ++j) Jc[j] = hand[(6 + a * NC + j) * HS];
            r = hand[(6 + 2 * NC + a) * HS];
          }
          if (a == 1) publish(hw_read, hbase + k + 1);
          if (live) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < NC; ++j) C[i][j] += Jq[i] * Jc[j];
            int t = 0;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
              gc[i] += Jc[i] * r;
#pragma unroll
              for (int j = 0; j <= i; ++j) cc[t++] += Jc[i] * Jc[j];
            }
          }
        }
      }
      hbase += kmax;
      asm volatile("" :: "v"(nx_si), "v"(nx_pt), "v"(nx_fp), "v"(nx_np), "v"(nx_gid0), "v"(nx_r[0]), "v"(nx_r[1]));
      LaneGeom q;
      STAMPB(1);
      lane_geom(flo + lf, pt, q);                       // in flight while the Z matrix is zero-filled
      lds_barrier();                                                                                    // ---- barrier P2a
      if (mode == 0) zero_zd(krows);
      lds_barrier();                                                                                    // ---- barrier P2
      STAMPB(2);
      if (det) wait_for(turn, my_turn);
      if (mode == 0 && cnt > 0) {
        // camera x pose block of the lane's frame and the camera part of W (k_sweep2's emission, second half)
        const double (&R)[9] = q.R; const double (&Gr)[3][3] = q.Gr;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < NC; ++j) C[i][j] *= c.chm[j];   // sign/scale folding and free-column mask, once per lane
        RunMask rm = run_masks(cnt, lf);
        rm.head = rm.head && d.frame_live[flo + lf] != 0;   // constant pose: no camera x pose block
        double* fr_acc = Fr + lf;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) { const double v = run_sum(C[0][j] * Gr[0][ci] + C[1][j] * Gr[1][ci] + C[2][j] * Gr[2][ci], rm); if (rm.head) atomicAdd(fr_acc + (size_t)(27 + j * 6 + ci) * frs, v); }
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) { const double v = run_sum(C[ci][j], rm); if (rm.head) atomicAdd(fr_acc + (size_t)(27 + j * 6 + 3 + ci) * frs, v); }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {   // (columns rotated by one for the lanes of odd points: see the evaluators' W rows)
          double* zrow = Zd + (size_t)(3 * lp + i) * zs + 6 * nf;
          double wc[NC];
#pragma unroll
          for (int j = 0; j < NC; ++j) wc[j] = R[i] * C[0][j] + R[3 + i] * C[1][j] + R[6 + i] * C[2][j];
          const bool rot = (lp & 1u) != 0;
#pragma unroll
          for (int jj = 0; jj < NC; ++jj) atomicAdd(zrow + (rot ? (uint32_t)((jj + 1) % NC) : (uint32_t)jj), rot ? wc[(jj + 1) % NC] : wc[jj]);
        }
      }
      if (det) { publish(turn, my_turn + 1); my_turn += (uint32_t)WR; }
      STAMPB(3);
      lds_barrier();                                                                                    // ---- barrier P4 (the factor phase ran on evaluator wave 0 meanwhile)
      STAMPB(4);
      if (mode == 0) {
        z_phase(np);
        lds_barrier();                                                                                  // ---- barrier P5
      }
      if (ksplit && t256 < ntri) {   // rows 4-7 of every 8 of the Schur product; the partial tile lives in LDS between passes
        uint32_t mi, mj; tri_decode(t256, mi, mj);
        double acc16[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc16[i][j] = bt[(i * 4 + j) * LP + t256];
        gemm_tile(mi, mj, 4, 8, krows, acc16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) bt[(i * 4 + j) * LP + t256] = acc16[i][j];
      }
      if (share_tiles) {   // the accumulator threads' share of the pass's tiles (see the evaluator branch)
        for (uint32_t t = tid; t < ntri; t += NT) {
          uint32_t mi, mj; tri_decode(t, mi, mj);
          double acc16[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = 0.0;
          gemm_tile(mi, mj, 0, 4, krows, acc16);
          emit_tile(mi, mj, acc16);
        }
      }
      lds_barrier();                                                                                    // ---- barrier P6
      STAMPB(5);
    }
    {  // sign/scale folding for the thread's camera x camera block and camera gradient
      int t = 0;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        gc[i] *= c.chm[i];
#pragma unroll
        for (int j = 0; j <= i; ++j) cc[t++] *= c.chm[i] * c.chm[j];
      }
    }
    // Reduction of the thread's camera x camera block and camera gradient over the accumulator threads: four DPP row shifts leave
    // the sum of every row of 16 lanes in its lane 15, those 4 WR row sums per value go to the free Z region, one thread per value
    // adds them up in a fixed order (bitwise reproducible, no atomics: nobody else touches these window entries after barrier T1).
    // (The 256-way reduction through LDS it replaces — [value][thread], two rounds of 28 values, four barriers — was 4 % of a block.)
    constexpr int NVB = NCC + NC;
    constexpr uint32_t NROWS = 4u * WR;   // rows of 16 lanes among the accumulator threads
    {
      double* red = Zd;   // [row][NVB]
      const uint32_t rowi = t256 >> 4;
      const bool tail = (lane & 15u) == 15u;
#pragma unroll
      for (int idx = 0; idx < NVB; ++idx) {
        double v = (idx < NCC) ? cc[idx < NCC ? idx : 0] : gc[(idx - NCC) >= 0 && (idx - NCC) < NC ? idx - NCC : 0];
        v = dpp_add_step<0x111, 0xf>(v); v = dpp_add_step<0x112, 0xf>(v); v = dpp_add_step<0x114, 0xf>(v); v = dpp_add_step<0x118, 0xf>(v);
        if (tail) red[rowi * NVB + idx] = v;
      }
      lds_barrier();                                                                                    // ---- barrier T1
      if (t256 < (uint32_t)NVB) {
        const int idx = (int)t256;
        double sacc = 0.0;
#pragma unroll
        for (uint32_t k = 0; k < NROWS; ++k) sacc += red[k * NVB + idx];
        if (idx < NCC) {
          int i = 0; while ((i + 1) * (i + 2) / 2 <= idx) ++i;
          if (mode == 0) Scc[idx] += sacc;
          if (idx == i * (i + 1) / 2 + i) vhd[6 * NFm + i] += sacc;
        } else {
          vgB[6 * NFm + (idx - NCC)] += sacc;
        }
      }
      lds_barrier();                                                                                    // ---- barrier T2
    }
  }
  // =======================================================================================================================
  // common tail, 512 threads: fold the replicas, flush the window (k_sweep2's)
  // =======================================================================================================================
  lds_barrier();
  STAMP(10);
  for (uint32_t i = tid; i < FRV * nf; i += NT) {
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
  lds_barrier();
  STAMP(11);
  if (det) {
    // options.deterministic: no cross-block atomics.  The block's window (Spp | Scp | Scc | gB | hdiag | rhs) and its three
    // scalars go to the block's slab in HBM as they are; k_det_reduce adds the slabs up in block order, one owner per entry.
    double* out = d.det_slab + (size_t)b * d.det_stride;
    for (uint32_t i = tid; i < lay.off_fr; i += NT) out[i] = sm[i];
    if (tid < 3) out[lay.off_fr + tid] = misc[tid];
    return;
  }
  const uint32_t F6 = 6 * d.F, camrow = 3 * d.Q, camcol = F6 + 3 * d.Q;
  for (uint32_t i = tid; i < 6 * nf; i += NT) atomicAdd(d.hdiag + 6 * flo + i, vhd[i]);
  if (tid < (uint32_t)NC) atomicAdd(d.hdiag + camcol + tid, vhd[6 * NFm + tid]);
  if (mode == 0) {
    const uint32_t npp = nf * (nf + 1) / 2;
    for (uint32_t i = tid; i < npp * 36; i += NT) {
      const uint32_t blk = i / 36, e = i % 36;
      uint32_t a = (uint32_t)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > blk) --a;
      while ((a + 1) * (a + 2) / 2 <= blk) ++a;
      const uint32_t bb = blk - a * (a + 1) / 2, dd = a - bb;
      const double v = Spp[i];
      if (dd <= d.bw && v != 0.0 && !(dd == 0 && (e % 6) > (e / 6))) atomicAdd(d.Sband + ((size_t)(flo + a) * (d.bw + 1) + dd) * 36 + e, v);
    }
    for (uint32_t i = tid; i < (uint32_t)NC * 6 * nf; i += NT) {
      const uint32_t j = i / (6 * nf), cidx = i % (6 * nf);
      atomicAdd(d.Sarrow + (size_t)(camrow + j) * d.ld + 6 * flo + cidx, Scp[(size_t)j * 6 * NFm + cidx]);
    }
    if (tid < (uint32_t)NCC) {
      uint32_t i = 0; while ((i + 1) * (i + 2) / 2 <= tid) ++i;
      const uint32_t j = tid - i * (i + 1) / 2;
      atomicAdd(d.Sarrow + (size_t)(camrow + i) * d.ld + camcol + j, Scc[tid]);
    }
    for (uint32_t i = tid; i < 6 * nf; i += NT) { atomicAdd(d.gB + 6 * flo + i, vgB[i]); atomicAdd(d.rhsacc + 6 * flo + i, vrhs[i]); }
    if (tid < (uint32_t)NC) { atomicAdd(d.gB + camcol + tid, vgB[6 * NFm + tid]); atomicAdd(d.rhsacc + camcol + tid, vrhs[6 * NFm + tid]); }
    if (tid == 0) {
      atomicAdd(d.scal + SCAL_COST, misc[0]);
      if (misc[1] != 0.0) atomicAdd(d.scal + SCAL_BAD_U, misc[1]);
      atomicMax((unsigned long long*)(d.scal + SCAL_GMAX0 + d.rank), *(unsigned long long*)(misc + 2));
    }
  }
#ifdef LIFCAL_STAMPS
  lds_barrier();
  STAMP(5);
  if ((tid == 0 || tid == LP) && d.dbg) for (int i = 0; i < 16; ++i) d.dbg[(size_t)b * 32 + (tid == 0 ? 0 : 16) + i] = st_acc[i];
#endif
}

}  // namespace lifcal
