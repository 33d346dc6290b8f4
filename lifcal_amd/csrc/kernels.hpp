// kernels.hpp — gfx950 kernels of the bundle-adjustment hot path (wave64, fp64).
//
// Sweep = what ceres does per LM iteration for the reference:
//   k_tables      camera constants, frame table, lens table          (hoisted out of the per-obs functor)
//   k_sweep       residual + analytic Jacobian per observation, robust weights (CauchyLoss(0.5),
//                 reference src/CameraCalibration.cpp:892), and the block accumulation
//                 U_p, g_p, W_p (point blocks) and B, g_B (camera+pose blocks)
//   k_constraints distance constraints (reference BundleAdjustment.h:255-279)
//   k_schur       U_p + D_p -> inverse; S -= W^T U^-1 W ; rhs += W^T U^-1 g   (ceres SchurEliminator)
//   k_finalize    LM diagonal on the reduced system, rhs, identity on fixed columns
//   k_band_chol / k_band_backsolve   block-banded + arrow Cholesky of S (ceres DenseSchurComplementSolver)
//   k_update_reduced / k_backsub     step for camera+poses, back-substitution for points, candidate point
//   k_cost        candidate cost ; k_stats reprojection statistics (reference :1026-1103)
//
// Layout: one LANE owns one (point, frame) group and walks its ~6 observations; a wave is a tile of
// 64 groups whose observation payload is stored [k][lane] so step k is one coalesced 512-B load per
// array.  Everything shared by the group (camera-frame point, 1/(Z+zC0), pose) is computed once; the
// per-observation outer products are accumulated in camera-frame space (A = sum Jq^T Jq, b = sum Jq^T r,
// C = sum Jq^T Jc) and rotated into pose/point blocks once per group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/lifcal_ba.h"
#include "device_model.hpp"

namespace lifcal {

constexpr uint32_t DET_NF_MAX = 20;   // widest frame window of a block (Plan::NF_MAX; lifcal_ba.hip checks the two against each other)
constexpr int SCAL_COST = 0, SCAL_BAD_U = 1, SCAL_GMAX0 = 2, SCAL_N = 2 + 64;
// host-visible scalars of one LM step
constexpr int ST_GTD = 0, ST_DDD = 1, ST_STEP2 = 2, ST_X2 = 3, ST_CAND_COST = 4, ST_CHOL_FAIL = 5, ST_GMAX_RED = 6, ST_DIR = 7, ST_N = 8;

// device-resident state of the Levenberg-Marquardt loop (k_lm_control; the host mirrors it once per iteration)
enum { LM_RADIUS = 0, LM_DECREASE = 1, LM_X_COST = 2, LM_GMAX = 3, LM_ITER = 4, LM_INVALID = 5, LM_STEP_OK = 6, LM_SUCCESSFUL = 7, LM_UNSUCCESSFUL = 8,
       LM_TERMINATION = 9, LM_COMMIT = 10, LM_FRESH = 11, LM_INITIAL_COST = 12, LM_LAST_REL = 13, LM_LAST_STEP = 14, LM_LAST_CHANGE = 15, LM_SWEEPS = 16,
       LM_SEQ = 17 /* round counter of the host mirror */, LM_T0 = 18, LM_TICKS_LINEAR = 19 /* 100 MHz ticks: linear solve + candidate evaluation */, LM_N = 24 };
struct LmOpts { double f_tol, p_tol, g_tol, min_rel_decrease, max_radius, min_radius; int max_iterations; };

// a set of tiles for the value-only kernels (both sweep paths share them)
struct TileSet { uint32_t n_tiles; const uint32_t *tile_row0, *slot_pt, *slot_fr, *slot_cnt, *ell_lens; const double *ell_u, *ell_v; };

struct Dev {
  // sizes
  uint32_t F, P, Q, NA, bw, nc, n_tiles, n_slots, n_lenses, n_red, ld, M_local, n_owned;
  uint32_t n_radial, tangential, adj, robust, use_poses, use_points, rank, world;
  uint32_t fixed_mask;
  double spx, spy, scale, loss_scale;
  double lm_min, lm_max;
  // parameters (current / candidate)
  double *cam, *views, *pts, *cam_c, *views_c, *pts_c;
  const double *lower, *upper;  // 17 each or null
  // tables
  CamConsts *camc, *camc_c;
  double *ft, *ft_c, *lt, *lt_c;
  // observations
  const uint32_t *tile_row0, *slot_pt, *slot_fr, *slot_cnt, *ell_lens;
  const double *ell_u, *ell_v;
  // points
  const int32_t* promoted;       // P
  const uint32_t* promoted_ids;  // Q
  double* panel_g;               // global panel scratch for k_band_chol when it does not fit LDS (else null)
  const uint32_t *pt_slot0, *pt_nslots, *owned;  // P, P, n_owned
  const uint8_t* frame_live;     // F
  double *ptacc;                 // P*36: U(6) g(3) Wc(27)
  double *Uinv, *lamP, *sigP;    // 9P, 3P, 3P
  double *Wv;                    // 18 per slot: W_pose of the groups of SPECIAL points (v1 kernels)
  double *Av;                    // 6 per slot: camera-frame block A = sum Jq^T Jq of the lanes of REGULAR points; W_pose = R^T A [Gr | I] is rebuilt in k_backsub
  const uint8_t* pt_special;     // P: 1 = the point takes the v1 kernels
  const uint32_t *gid_fr, *slot_gid;              // frame of each group id; v1 path: group id of each slot
  // v2 (LDS-window) path
  uint32_t n_blocks, v2_nfmax, n_special;
  const uint32_t *blk_pass0, *blk_flo, *blk_nf, *pass_pt0, *pass_np, *pass_gid0, *pass_ng, *v2_points, *v2_ptinfo, *v2_slot, *v2_tile_row0, *v2_lens, *v2f_pt, *v2_passpt, *v2_gidx;
  const double *v2_u, *v2_v;
  // k_front4 / k_back4 (sweep4.hpp): front workgroup -> block and pass range; block -> range of v2_points
  uint32_t n_fwg;
  const uint32_t *fwg_blk, *fwg_pass0, *blk_pt0;
  const float *v2_du, *v2_dv;    // options.precision = 1: observation relative to its micro-lens centre (u - mcx, v - mcy), fp32
  float* ltf;                    // options.precision = 1: fp32 lens table of the CURRENT point (see lens_row_to_float)
  double* ltw;                   // ... and w = (a) c_u of every lens in fp64 (2 per lens): the one fp64 operand of the fp32 evaluation
  // options.deterministic = 1: per-block slabs of the LDS windows (k_det_reduce sums them in block order), per-workgroup slots
  // of the value-only kernels (k_det_sum)
  uint32_t deterministic, det_stride;
  double *det_slab, *det_slots;
  uint32_t* det_turn;            // ... and the turn counters of the global-atomic kernels (special points): [0] k_sweep, [1] k_schur
  const uint32_t* special_owned;
  // constraints
  const uint32_t *c_i, *c_j, *my_cons, *pt_cons0, *pt_cons_list; const double *c_dist, *c_sigma;
  double* Wpart;                 // 9 per constraint
  // reduced system (one contiguous all-reduced block): Sband | Sarrow | rhsacc | gB | hdiag | scal
  double *Sband, *Sarrow, *rhsacc, *gB, *hdiag, *scal;
  double *sig_red, *lam_red, *delta_red, *Linv;  // n_red, n_red, n_red, 36F
  double* lm;                    // LM_N doubles: the device-resident trust-region state (k_lm_control), radius first
  double* step;                  // ST_N scalars
  double* dP;                    // 3P: unscaled point step of the last back-substitution (line search re-applies it)
  unsigned long long* dbg;       // diagnostic build only (LIFCAL_STAMPS): per-block phase cycle counts
};

// trust-region radius of a kernel: the launch argument, or (negative argument) the device-resident state
LIFCAL_DEV double lm_radius(const Dev& d, double arg) { return arg < 0.0 ? d.lm[LM_RADIUS] : arg; }

LIFCAL_DEV double* s_addr(const Dev& d, uint32_t row, uint32_t col) {  // row >= col, internal ordering
  if (row < 6 * d.F) {
    const uint32_t f = row / 6, i = row % 6, fc = col / 6, j = col % 6;
    return d.Sband + ((size_t)f * (d.bw + 1) + (f - fc)) * 36 + i * 6 + j;
  }
  return d.Sarrow + (size_t)(row - 6 * d.F) * d.ld + col;
}
LIFCAL_DEV void s_add(const Dev& d, uint32_t row, uint32_t col, double v) {
  if (row >= col) atomicAdd(s_addr(d, row, col), v); else atomicAdd(s_addr(d, col, row), v);
}

LIFCAL_DEV double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// options.deterministic = 1 on the kernels that sum with GLOBAL atomics (special points: k_sweep, k_schur): the waves emit one
// after the other, wave `my` after wave `my - 1`, so every accumulator receives its addends in one fixed order (the atomics of one
// wave reach an address in program order, the lanes of one instruction in the memory pipeline's fixed lane order).  The
// hand-over is a release / acquire pair at agent scope: the release waits for the wave's own atomics (s_waitcnt vmcnt(0): performed
// at the coherence point) before the counter moves.  No deadlock for any grid size: a wave waits only for its predecessor,
// workgroups are dispatched in index order, so the lowest unfinished wave is always resident.  The wait is bounded (4 s of the
// 100 MHz clock): a protocol error would show up as SCAL_BAD_U — a failed solve — never as a hung GPU.
LIFCAL_DEV void det_turn_wait(const Dev& d, uint32_t which, uint32_t my) {
  uint32_t* turn = d.det_turn + which;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
    if (v == my) break;
    __builtin_amdgcn_s_sleep(16);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) {
      if ((threadIdx.x & 63u) == 0) atomicAdd(d.scal + SCAL_BAD_U, 1.0);
      break;
    }
  }
}
LIFCAL_DEV void det_turn_pass(const Dev& d, uint32_t which, uint32_t my) {
  if ((threadIdx.x & 63u) == 0) __hip_atomic_store(d.det_turn + which, my + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------------------------
__global__ void k_camc(const double* cam, CamConsts* out, double spx, double spy, double scale, int n_radial,
                       int tangential, unsigned fixed_mask, double loss_scale, int fold) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    CamConsts c;
    cam_prepare(cam, spx, spy, fold ? scale : (double)(float)scale, n_radial, tangential != 0, fixed_mask, loss_scale, fold != 0, c);
    *out = c;
  }
}

__global__ void k_frames(const double* views, double* ft, uint32_t F) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < F) { double o[FRAME_STRIDE]; frame_eval(views + 6 * (size_t)f, o);
#pragma unroll
    for (int k = 0; k < FRAME_STRIDE; ++k) ft[(size_t)f * FRAME_STRIDE + k] = o[k]; }
}

template <int NR, bool TAN>
__global__ void k_lenses(const CamConsts* camc, const double* lens_xy, double* lt, uint32_t n, int want_tangents) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  const CamConsts c = *camc;
  double o[LENS_STRIDE];
#pragma unroll
  for (int k = 0; k < LENS_STRIDE; ++k) o[k] = 0.0;
  lens_eval<NR, TAN>(c, lens_xy[2 * (size_t)l], lens_xy[2 * (size_t)l + 1], want_tangents != 0, o);
#pragma unroll
  for (int k = 0; k < LENS_STRIDE; ++k) lt[(size_t)l * LENS_STRIDE + k] = o[k];
}

// one launch for everything that precedes a sweep: camera constants (recomputed by every thread: ~100 flops, no
// dependency on another kernel), lens table, frame table, and zero-filling of the accumulation buffers
template <int NR, bool TAN>
__global__ __launch_bounds__(256) void k_tables(Dev d, const double* cam, const double* views, CamConsts* camc_out, double* ft, double* lt,
                                                const double* lens_xy, int want_tangents, int fold, double* zero0, uint32_t n_zero0, double* zero1, uint32_t n_zero1,
                                                float* ltf /* fp32 lens table (options.precision = 1) or null */) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
  CamConsts c;
  cam_prepare(cam, d.spx, d.spy, fold ? d.scale : (double)(float)d.scale, (int)d.n_radial, d.tangential != 0, d.fixed_mask, d.loss_scale, fold != 0, c);
  if (t == 0) *camc_out = c;
  if (t < d.n_lenses) {
    double o[LENS_STRIDE];
#pragma unroll
    for (int k = 0; k < LENS_STRIDE; ++k) o[k] = 0.0;
    lens_eval<NR, TAN>(c, lens_xy[2 * (size_t)t], lens_xy[2 * (size_t)t + 1], want_tangents != 0, o);
#pragma unroll
    for (int k = 0; k < LENS_STRIDE; ++k) lt[(size_t)t * LENS_STRIDE + k] = o[k];
    if (ltf) {
      float of[LENS_STRIDE]; double w64[2];
      if (d.adj) lens_row_to_float<true>(c, o, of, w64); else lens_row_to_float<false>(c, o, of, w64);
#pragma unroll
      for (int k = 0; k < LENS_STRIDE; ++k) ltf[(size_t)t * LENS_STRIDE + k] = of[k];
      d.ltw[2 * (size_t)t] = w64[0]; d.ltw[2 * (size_t)t + 1] = w64[1];
    }
  }
  if (t < d.F) {
    double o[FRAME_STRIDE]; frame_eval(views + 6 * (size_t)t, o);
#pragma unroll
    for (int k = 0; k < FRAME_STRIDE; ++k) ft[(size_t)t * FRAME_STRIDE + k] = o[k];
  }
  for (uint32_t i = t; i < n_zero0; i += nthreads) zero0[i] = 0.0;
  for (uint32_t i = t; i < n_zero1; i += nthreads) zero1[i] = 0.0;
}

// point slabs of the special points (the v1 kernels accumulate into them with atomics)
__global__ void k_zero_special(Dev d) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < d.n_special * 36) d.ptacc[(size_t)d.special_owned[t / 36] * 36 + t % 36] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// sweep: one lane per (point, frame) group
// ---------------------------------------------------------------------------------------------
template <int NR, bool TAN, bool ADJ>
__global__ __launch_bounds__(256) void k_sweep(Dev d) {
  constexpr int NC = 5 + NR + (TAN ? 2 : 0);
  constexpr int NCC = NC * (NC + 1) / 2;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
  // options.deterministic: one tile per wave (the host sizes the grid), the waves emit in tile order (det_turn_wait / _pass)
  if (d.deterministic && wave >= d.n_tiles) return;
  const CamConsts c = *d.camc;
  double cc[NCC], gc[NC], cost = 0.0;
#pragma unroll
  for (int i = 0; i < NCC; ++i) cc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < NC; ++i) gc[i] = 0.0;

  for (uint32_t tile = wave; tile < d.n_tiles; tile += n_waves) {
    const uint32_t slot = tile * 64 + lane;
    const uint32_t cnt = d.slot_cnt[slot];
    const uint32_t row0 = d.tile_row0[tile], kmax = d.tile_row0[tile + 1] - row0;
    const uint32_t pt = d.slot_pt[slot], fr = d.slot_fr[slot];
    double R[9], Y[3], c0 = 1.0, s0 = 0.0;
    GroupConsts g;
    {
      const double* ft = d.ft + (size_t)fr * FRAME_STRIDE;
      const double* P = d.pts + 3 * (size_t)pt;
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = ft[k];
      const double P0 = P[0], P1 = P[1], P2 = P[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) Y[k] = R[3 * k] * P0 + R[3 * k + 1] * P1 + R[3 * k + 2] * P2;
      c0 = ft[12]; s0 = ft[13];
      group_prepare(c, Y[0] + ft[9], Y[1] + ft[10], Y[2] + ft[11], g);
    }
    double A[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}, C[3][NC];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < NC; ++j) C[i][j] = 0.0;

    for (uint32_t k = 0; k < kmax; ++k) {
      if (k < cnt) {
        const size_t at = ((size_t)row0 + k) * 64 + lane;
        const double u = d.ell_u[at], v = d.ell_v[at];
        const double* L = d.lt + (size_t)d.ell_lens[at] * LENS_STRIDE;
        double r[2], Jq[2][3], Jc[2][NC];
        obs_eval<NR, TAN, ADJ>(c, g, L, u, v, r, Jq, Jc);
        const double sq = r[0] * r[0] + r[1] * r[1];
        if (d.robust) {  // ceres::CauchyLoss + Corrector with rho'' < 0: scale r and J by sqrt(rho')
          const double sum = 1.0 + sq * c.loss_c;
          const double inv = 1.0 / sum;
          cost += 0.5 * c.loss_b * log(sum);
          const double sc = sqrt(fmax(inv, 2.2250738585072014e-308));
          r[0] *= sc; r[1] *= sc;
#pragma unroll
          for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int j = 0; j < 3; ++j) Jq[a][j] *= sc;
#pragma unroll
            for (int j = 0; j < NC; ++j) Jc[a][j] *= sc;
          }
        } else {
          cost += 0.5 * sq;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          A[0] += Jq[a][0] * Jq[a][0]; A[1] += Jq[a][0] * Jq[a][1]; A[2] += Jq[a][0] * Jq[a][2];
          A[3] += Jq[a][1] * Jq[a][1]; A[4] += Jq[a][1] * Jq[a][2]; A[5] += Jq[a][2] * Jq[a][2];
#pragma unroll
          for (int i = 0; i < 3; ++i) b[i] += Jq[a][i] * r[a];
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

    if (d.deterministic) det_turn_wait(d, 0, tile);   // (wave-uniform: a scalar branch, not a lane mask)
    if (cnt > 0 && d.use_poses) {
      const double Am[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
      // Gr = [e_x x Y, (0,c0,s0) x Y, R[:,2] x Y]
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
      // pose-pose block (lower), pose gradient, pose diagonal.  A frame whose pose is held constant (lifcal_ba_set_fixed_frames: its
      // observations still constrain the points, the pose columns do not exist) contributes zeros: the mask is a FACTOR, not a
      // branch — this kernel keeps its 54 camera accumulators in AGPRs and updates them under the lane masks of the observation
      // loop; a further divergent region around the scatter is what once made those accumulators lose contributions
      const uint32_t pr = 6 * fr;
      double* Sd = d.Sband + (size_t)fr * (d.bw + 1) * 36;
      const double fm = d.frame_live[fr] != 0 ? 1.0 : 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = 0; bb <= a; ++bb) {
          double v;
          if (a < 3) v = GAG[a][bb]; else if (bb < 3) v = AG[a - 3][bb]; else v = Am[a - 3][bb - 3];
          v *= fm;
          atomicAdd(Sd + a * 6 + bb, v);
          if (a == bb) atomicAdd(d.hdiag + pr + a, v);
        }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        atomicAdd(d.gB + pr + a, fm * (Gr[0][a] * b[0] + Gr[1][a] * b[1] + Gr[2][a] * b[2]));
        atomicAdd(d.gB + pr + 3 + a, fm * b[a]);
      }
      // camera x pose block: cp[j][c] = C^T [Gr | I]
      const uint32_t camrow = 3 * d.Q;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        double* row = d.Sarrow + (size_t)(camrow + j) * d.ld + pr;
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) atomicAdd(row + cidx, fm * (C[0][j] * Gr[0][cidx] + C[1][j] * Gr[1][cidx] + C[2][j] * Gr[2][cidx]));
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) atomicAdd(row + 3 + cidx, fm * C[cidx][j]);
      }
      if (d.use_points) {
        double AR[3][3], U[3][3], gP[3], Wc[3][NC], Wv[3][6];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) AR[i][j] = Am[i][0] * R[j] + Am[i][1] * R[3 + j] + Am[i][2] * R[6 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) U[i][j] = R[i] * AR[0][j] + R[3 + i] * AR[1][j] + R[6 + i] * AR[2][j];
          gP[i] = R[i] * b[0] + R[3 + i] * b[1] + R[6 + i] * b[2];
#pragma unroll
          for (int j = 0; j < NC; ++j) Wc[i][j] = R[i] * C[0][j] + R[3 + i] * C[1][j] + R[6 + i] * C[2][j];
#pragma unroll
          for (int j = 0; j < 3; ++j) Wv[i][j] = R[i] * AG[0][j] + R[3 + i] * AG[1][j] + R[6 + i] * AG[2][j];
#pragma unroll
          for (int j = 0; j < 3; ++j) Wv[i][3 + j] = R[i] * Am[0][j] + R[3 + i] * Am[1][j] + R[6 + i] * Am[2][j];
        }
        // every point accumulates into its own slab; promoted points are scattered into the reduced
        // system by k_schur (keeps this kernel free of divergent control flow)
        double* acc = d.ptacc + (size_t)pt * 36;
        atomicAdd(acc + 0, U[0][0]); atomicAdd(acc + 1, U[1][0]); atomicAdd(acc + 2, U[2][0]);
        atomicAdd(acc + 3, U[1][1]); atomicAdd(acc + 4, U[2][1]); atomicAdd(acc + 5, U[2][2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) atomicAdd(acc + 6 + i, gP[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < NC; ++j) atomicAdd(acc + 9 + i * NCMAX + j, Wc[i][j]);
        double* wv = d.Wv + (size_t)d.slot_gid[slot] * 18;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) wv[i * 6 + j] = fm * Wv[i][j];
      }
    }
  }
  // camera x camera block, camera gradient, cost: reduce across the wave, one lane per value
  const uint32_t camrow = 3 * d.Q, camcol = 6 * d.F + 3 * d.Q;
  {
    int t = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const double s = wave_sum(cc[t]);
        if (lane == (uint32_t)(t & 63)) { atomicAdd(d.Sarrow + (size_t)(camrow + i) * d.ld + camcol + j, s); if (i == j) atomicAdd(d.hdiag + camcol + i, s); }
        ++t;
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) { const double s = wave_sum(gc[i]); if (lane == (uint32_t)i) atomicAdd(d.gB + camcol + i, s); }
    const double s = wave_sum(cost);
    if (lane == 63) atomicAdd(d.scal + SCAL_COST, s);
  }
  if (d.deterministic) det_turn_pass(d, 0, wave);
}

// one column of the reduced system: ceres LevenbergMarquardtStrategy diagonal (clamp(sigma^2 h) / (radius sigma^2) in the
// unscaled space), rhs = -g_B + W^T U^-1 g into the rhs arrow row, identity for columns that are not solved for
LIFCAL_DEV double finalize_column(const Dev& d, uint32_t t, double radius) {
  const uint32_t F6 = 6 * d.F;
  bool live;
  if (t < F6) live = d.use_poses && d.frame_live[t / 6];
  else if (t < F6 + 3 * d.Q) live = true;
  else live = d.camc->chm[t - F6 - 3 * d.Q] != 0.0;
  double* diag = s_addr(d, t, t);
  double* rhs = d.Sarrow + (size_t)d.NA * d.ld;  // extra arrow row carries the right-hand side
  if (live) {
    const double s = d.sig_red[t];
    const double lam = fmin(fmax(d.hdiag[t] * s * s, d.lm_min), d.lm_max) / (lm_radius(d, radius) * s * s);
    d.lam_red[t] = lam;
    *diag += lam;
    rhs[t] = -d.gB[t] + d.rhsacc[t];
  } else {
    d.lam_red[t] = 0.0;
    *diag = 1.0;
    rhs[t] = 0.0;
  }
  return fabs(d.gB[t]);
}

}  // namespace lifcal
#include "sweep2.hpp"   // k_sweep2: the LDS-window fused sweep (regular points)
#include "sweep3.hpp"   // k_sweep3: the same with a wave-specialised observation loop (512 threads)
#include "sweep4.hpp"   // k_front4 + k_back4: the sweep cut in two kernels built for occupancy (default)
namespace lifcal {

// ---------------------------------------------------------------------------------------------
// distance constraints: r = (|Pi - Pj| - dist) / (sigma + 1e-6), squared loss; c_j is always promoted
// ---------------------------------------------------------------------------------------------
__global__ void k_constraints(Dev d, int cost_only, const double* pts, double* cost_out) {
  // (options.deterministic: launched as ONE thread, which takes the constraints in order)
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < d.M_local; t += gridDim.x * blockDim.x) {
  const uint32_t c = d.my_cons[t];
  const uint32_t pi = d.c_i[c], pj = d.c_j[c];
  const double* Pi = pts + 3 * (size_t)pi; const double* Pj = pts + 3 * (size_t)pj;
  const double dx = Pi[0] - Pj[0], dy = Pi[1] - Pj[1], dz = Pi[2] - Pj[2];
  const double n = sqrt(dx * dx + dy * dy + dz * dz);
  const double is = 1.0 / (d.c_sigma[c] + 0.000001);
  const double r = (n - d.c_dist[c]) * is;
  atomicAdd(cost_out, 0.5 * r * r);
  if (cost_only) continue;
  const double in = is / n;
  const double Ji[3] = {dx * in, dy * in, dz * in};  // J_j = -J_i
  const uint32_t F6 = 6 * d.F;
  const int32_t qi = d.promoted[pi], qj = d.promoted[pj];
  const uint32_t cj = F6 + 3 * (uint32_t)qj;
  // endpoint j (promoted): block, gradient, diagonal
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b <= a; ++b) s_add(d, cj + a, cj + b, Ji[a] * Ji[b]);
    atomicAdd(d.hdiag + cj + a, Ji[a] * Ji[a]);
    atomicAdd(d.gB + cj + a, -Ji[a] * r);
  }
  if (qi < 0) {
    double* acc = d.ptacc + (size_t)pi * 36;
    atomicAdd(acc + 0, Ji[0] * Ji[0]); atomicAdd(acc + 1, Ji[1] * Ji[0]); atomicAdd(acc + 2, Ji[2] * Ji[0]);
    atomicAdd(acc + 3, Ji[1] * Ji[1]); atomicAdd(acc + 4, Ji[2] * Ji[1]); atomicAdd(acc + 5, Ji[2] * Ji[2]);
    for (int a = 0; a < 3; ++a) atomicAdd(acc + 6 + a, Ji[a] * r);
    double* wp = d.Wpart + (size_t)c * 9;  // W_i partner block: J_i^T J_j
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) wp[a * 3 + b] = -Ji[a] * Ji[b];
  } else {
    const uint32_t ci = F6 + 3 * (uint32_t)qi;
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b <= a; ++b) s_add(d, ci + a, ci + b, Ji[a] * Ji[b]);
      atomicAdd(d.hdiag + ci + a, Ji[a] * Ji[a]);
      atomicAdd(d.gB + ci + a, Ji[a] * r);
      // cross block J_i^T J_j = -Ji Ji^T between two different promoted points
      for (int b = 0; b < 3; ++b) {
        const double v = -Ji[a] * Ji[b];
        if (ci > cj) atomicAdd(s_addr(d, ci + a, cj + b), v); else atomicAdd(s_addr(d, cj + b, ci + a), v);
      }
    }
  }
  }
}

// promoted points: the observation part of their Hessian diagonal sits in the point slab of the owning
// rank; fold it into hdiag (so that it is all-reduced with the rest) before the Jacobi scaling is taken
__global__ void k_promote_diag(Dev d) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= d.Q) return;
  const uint32_t p = d.promoted_ids[q];
  const double* acc = d.ptacc + (size_t)p * 36;
  const uint32_t col = 6 * d.F + 3 * q;
  atomicAdd(d.hdiag + col + 0, acc[0]); atomicAdd(d.hdiag + col + 1, acc[3]); atomicAdd(d.hdiag + col + 2, acc[5]);
}

// ---------------------------------------------------------------------------------------------
// Jacobi scaling (ceres: 1 / (1 + sqrt(column norm^2)), fixed at iteration 0)
// ---------------------------------------------------------------------------------------------
__global__ void k_jacobi(Dev d, int enabled) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < d.n_red) d.sig_red[t] = enabled ? 1.0 / (1.0 + sqrt(d.hdiag[t])) : 1.0;
  if (t < d.P) {
    const double* acc = d.ptacc + (size_t)t * 36;
    d.sigP[3 * (size_t)t + 0] = enabled ? 1.0 / (1.0 + sqrt(acc[0])) : 1.0;
    d.sigP[3 * (size_t)t + 1] = enabled ? 1.0 / (1.0 + sqrt(acc[3])) : 1.0;
    d.sigP[3 * (size_t)t + 2] = enabled ? 1.0 / (1.0 + sqrt(acc[5])) : 1.0;
  }
}

// ---------------------------------------------------------------------------------------------
// per-point Schur complement: one wave per owned, eliminated point
// ---------------------------------------------------------------------------------------------
struct WBlock { const double* W; uint32_t width, ldw, base; };

LIFCAL_DEV WBlock point_block(const Dev& d, uint32_t p, uint32_t bidx, uint32_t ns) {
  WBlock b;
  if (bidx == 0) { b.W = d.ptacc + (size_t)p * 36 + 9; b.width = d.nc; b.ldw = NCMAX; b.base = 6 * d.F + 3 * d.Q; }
  else if (bidx <= ns) { const uint32_t s = d.pt_slot0[p] + bidx - 1; b.W = d.Wv + (size_t)s * 18; b.width = 6; b.ldw = 6; b.base = 6 * d.gid_fr[s]; }
  else { const uint32_t c = d.pt_cons_list[d.pt_cons0[p] + (bidx - ns - 1)]; b.W = d.Wpart + (size_t)c * 9; b.width = 3; b.ldw = 3; b.base = 6 * d.F + 3 * (uint32_t)d.promoted[d.c_j[c]]; }
  return b;
}

__global__ __launch_bounds__(256) void k_schur(Dev d, double radius) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= d.n_special) return;
  const uint32_t p = d.special_owned[w];
  const double* acc = d.ptacc + (size_t)p * 36;
  if (d.promoted[p] >= 0) {
    // promoted point: its three columns belong to the reduced system (arrow rows 3q..3q+2):
    // U -> diagonal block, g -> gradient, Wc -> camera x point block, Wv -> point x pose blocks
    const uint32_t q = (uint32_t)d.promoted[p], F6 = 6 * d.F, col0 = F6 + 3 * q, camcol = F6 + 3 * d.Q;
    const uint32_t ns = d.pt_nslots[p];
    if (d.deterministic) det_turn_wait(d, 1, w);
    if (lane < 6) { const int ii[6] = {0, 1, 2, 1, 2, 2}, jj[6] = {0, 0, 0, 1, 1, 2}; s_add(d, col0 + ii[lane], col0 + jj[lane], acc[lane]); }
    if (lane < 3) atomicAdd(d.gB + col0 + lane, acc[6 + lane]);
    for (uint32_t t = lane; t < 3 * d.nc; t += 64) { const uint32_t i = t / d.nc, j = t % d.nc; s_add(d, camcol + j, col0 + i, acc[9 + i * NCMAX + j]); }
    for (uint32_t t = lane; t < 18 * ns; t += 64) {
      const uint32_t sidx = d.pt_slot0[p] + t / 18, e = t % 18, i = e / 6, j = e % 6;
      s_add(d, col0 + i, 6 * d.gid_fr[sidx] + j, d.Wv[(size_t)sidx * 18 + e]);
    }
    if (d.deterministic) det_turn_pass(d, 1, w);
    return;
  }
  // damped point block (ceres LevenbergMarquardtStrategy: D^2 = clamp(diag(J^T J)) / radius in scaled space)
  double U[6], lam[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) U[k] = acc[k];
  const double g0 = acc[6], g1 = acc[7], g2 = acc[8];
  {
    const double h[3] = {U[0], U[3], U[5]};
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double s = d.sigP[3 * (size_t)p + k]; lam[k] = fmin(fmax(h[k] * s * s, d.lm_min), d.lm_max) / (lm_radius(d, radius) * s * s); }
  }
  U[0] += lam[0]; U[3] += lam[1]; U[5] += lam[2];
  // inverse through the Cholesky factor (ceres InvertPSDMatrix<3>)
  double iv[9]; bool ok = true;
  {
    double l00 = U[0]; ok = ok && (l00 > 0.0); l00 = sqrt(l00);
    const double l10 = U[1] / l00, l20 = U[2] / l00;
    double l11 = U[3] - l10 * l10; ok = ok && (l11 > 0.0); l11 = sqrt(l11);
    const double l21 = (U[4] - l20 * l10) / l11;
    double l22 = U[5] - l20 * l20 - l21 * l21; ok = ok && (l22 > 0.0); l22 = sqrt(l22);
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    const double m10 = -l10 * i00 * i11, m21 = -l21 * i11 * i22, m20 = -(l20 * i00 + l21 * m10) * i22;  // L^-1
    iv[0] = i00 * i00 + m10 * m10 + m20 * m20; iv[1] = m10 * i11 + m20 * m21; iv[2] = m20 * i22;
    iv[4] = i11 * i11 + m21 * m21;            iv[5] = m21 * i22;             iv[8] = i22 * i22;
    iv[3] = iv[1]; iv[6] = iv[2]; iv[7] = iv[5];
    if (!ok) { for (int k = 0; k < 9; ++k) iv[k] = 0.0; }
  }
  if (d.deterministic) det_turn_wait(d, 1, w);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) d.Uinv[9 * (size_t)p + k] = iv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) d.lamP[3 * (size_t)p + k] = lam[k];
    if (!ok) atomicAdd(d.scal + SCAL_BAD_U, 1.0);
    // max |g_p| for the gradient-tolerance test
    const double gm = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
    atomicMax((unsigned long long*)(d.scal + SCAL_GMAX0 + d.rank), (unsigned long long)__double_as_longlong(gm));
  }
  const double ig[3] = {iv[0] * g0 + iv[1] * g1 + iv[2] * g2, iv[3] * g0 + iv[4] * g1 + iv[5] * g2, iv[6] * g0 + iv[7] * g1 + iv[8] * g2};
  const uint32_t ns = d.pt_nslots[p];
  const uint32_t ncon = d.pt_cons0 ? d.pt_cons0[p + 1] - d.pt_cons0[p] : 0;
  const uint32_t nb = 1 + ns + ncon;
  // rhs accumulation: W^T U^-1 g
  for (uint32_t bi = lane; bi < nb; bi += 64) {
    const WBlock B = point_block(d, p, bi, ns);
    for (uint32_t j = 0; j < B.width; ++j)
      atomicAdd(d.rhsacc + B.base + j, B.W[j] * ig[0] + B.W[B.ldw + j] * ig[1] + B.W[2 * B.ldw + j] * ig[2]);
  }
  // S -= W^T U^-1 W over unordered block pairs
  const uint32_t npairs = nb * (nb + 1) / 2;
  for (uint32_t t = lane; t < npairs; t += 64) {
    uint32_t bi = 0, rem = t;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    const uint32_t bj = bi + rem;
    const WBlock Bi = point_block(d, p, bi, ns), Bj = point_block(d, p, bj, ns);
    const bool same = (bi == bj), same_base = (Bi.base == Bj.base);
    for (uint32_t j = 0; j < Bj.width; ++j) {
      const double w0 = Bj.W[j], w1 = Bj.W[Bj.ldw + j], w2 = Bj.W[2 * Bj.ldw + j];
      const double y0 = iv[0] * w0 + iv[1] * w1 + iv[2] * w2, y1 = iv[3] * w0 + iv[4] * w1 + iv[5] * w2, y2 = iv[6] * w0 + iv[7] * w1 + iv[8] * w2;
      for (uint32_t i = same ? j : 0; i < Bi.width; ++i) {
        double m = -(Bi.W[i] * y0 + Bi.W[Bi.ldw + i] * y1 + Bi.W[2 * Bi.ldw + i] * y2);
        if (!same && same_base && i == j) m *= 2.0;
        s_add(d, Bi.base + i, Bj.base + j, m);
      }
    }
  }
  if (d.deterministic) det_turn_pass(d, 1, w);
}

// ---------------------------------------------------------------------------------------------
// start values of the plenoptic parameters (reference src/CameraCalibration.cpp:456-499): the five sums of the
// normal equations of bL = v B + bL0 over the valid rows, and the number of rows with an index out of range
// out: [0] sum v^2  [1] sum v  [2] rows used  [3] sum v bL  [4] sum bL  [5] bad indices
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_sums(uint64_t n, const double* vdepth, const uint32_t* fr, const uint32_t* pt, uint32_t n_frames,
                                                   uint32_t n_points, const double* w2c, const double* pts, double fL, double* out) {
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t f = fr[i], q = pt[i];
    if (f >= n_frames || q >= n_points) { s[5] += 1.0; continue; }
    const double* M = w2c + (size_t)f * 16;   // column-major 4x4: row 2 is the camera-frame z
    const double* P = pts + 3 * (size_t)q;
    const double z = M[2] * P[0] + M[6] * P[1] + M[10] * P[2] + M[14];
    const double bL = (fL * z) / (z - fL);     // reference :483
    const double v = vdepth[i];
    if ((v < 2.0) || (bL < 0.0)) continue;     // reference :485-490 zeroes the row (NaN compares false: the row stays, as there)
    s[0] += v * v; s[1] += v; s[2] += 1.0; s[3] += v * bL; s[4] += bL;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) { s[k] = wave_sum(s[k]); if ((threadIdx.x & 63u) == 0 && s[k] != 0.0) atomicAdd(out + k, s[k]); }
}

// ---------------------------------------------------------------------------------------------
// multi-GPU exchange of the reduced block as frame slabs (see lifcal_ba_allgather_fn in include/lifcal_ba.h)
// slab of one rank: band [maxn][BS] | camera x pose [maxn][NA][6] | rhsacc,gB,hdiag [maxn][3][6] |
//                   tail: arrow x arrow [NA][NA] | camera entries of the three vectors [3][NA] | scal[SCAL_N]
// (only used when no point is promoted: the arrow rows are then the NA = nc camera rows)
// ---------------------------------------------------------------------------------------------
struct Xch {
  uint32_t world, rank, maxn, BS, NA, F6;
  uint32_t off_arrow, off_vec, off_tail, SL;
  const uint32_t *flo, *nfr;   // per rank
  double* send; const double* recv;
};

__global__ void k_xch_pack(Dev d, Xch x) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= x.SL) return;
  const uint32_t flo = x.flo[x.rank], n = x.nfr[x.rank];
  double v = 0.0;
  if (s < x.off_arrow) {
    const uint32_t fl = s / x.BS, e = s - fl * x.BS;
    if (fl >= n) return;
    v = d.Sband[(size_t)(flo + fl) * x.BS + e];
  } else if (s < x.off_vec) {
    const uint32_t t = s - x.off_arrow, fl = t / (x.NA * 6), j = (t / 6) % x.NA, k = t % 6;
    if (fl >= n) return;
    v = d.Sarrow[(size_t)j * d.ld + 6 * (flo + fl) + k];
  } else if (s < x.off_tail) {
    const uint32_t t = s - x.off_vec, fl = t / 18, w = (t / 6) % 3, k = t % 6;
    if (fl >= n) return;
    const double* src = w == 0 ? d.rhsacc : (w == 1 ? d.gB : d.hdiag);
    v = src[6 * (flo + fl) + k];
  } else {
    const uint32_t t = s - x.off_tail;
    if (t < x.NA * x.NA) v = d.Sarrow[(size_t)(t / x.NA) * d.ld + x.F6 + t % x.NA];
    else if (t < x.NA * x.NA + 3 * x.NA) { const uint32_t q = t - x.NA * x.NA, w = q / x.NA; const double* src = w == 0 ? d.rhsacc : (w == 1 ? d.gB : d.hdiag); v = src[x.F6 + q % x.NA]; }
    else v = d.scal[t - x.NA * x.NA - 3 * x.NA];
  }
  x.send[s] = v;
}

// one thread per entry of the reduced block: sum of the slabs that cover its frame (every entry is rewritten)
__global__ void k_xch_unpack(Dev d, Xch x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nB = (size_t)(x.F6 / 6) * x.BS, nA = (size_t)x.NA * x.F6, nV = (size_t)3 * x.F6;
  const uint32_t nT = x.NA * x.NA + 3 * x.NA + SCAL_N;
  if (i >= nB + nA + nV + nT) return;
  double sum = 0.0;
  double* dst;
  if (i < nB + nA + nV) {
    uint32_t f, within;   // frame of the entry, offset of the entry inside a slab relative to the frame's record
    size_t base;          // start of the piece inside a slab
    uint32_t rec;         // record length per frame in that piece
    if (i < nB) { f = (uint32_t)(i / x.BS); within = (uint32_t)(i - (size_t)f * x.BS); base = 0; rec = x.BS; dst = d.Sband + i; }
    else if (i < nB + nA) {
      const size_t t = i - nB; const uint32_t j = (uint32_t)(t / x.F6), col = (uint32_t)(t - (size_t)j * x.F6);
      f = col / 6; within = j * 6 + col % 6; base = x.off_arrow; rec = x.NA * 6; dst = d.Sarrow + (size_t)j * d.ld + col;
    } else {
      const size_t t = i - nB - nA; const uint32_t w = (uint32_t)(t / x.F6), col = (uint32_t)(t - (size_t)w * x.F6);
      f = col / 6; within = w * 6 + col % 6; base = x.off_vec; rec = 18; dst = (w == 0 ? d.rhsacc : (w == 1 ? d.gB : d.hdiag)) + col;
    }
    for (uint32_t r = 0; r < x.world; ++r) {
      const uint32_t fl = f - x.flo[r];   // wraps for f < flo: fails the range test
      if (fl < x.nfr[r]) sum += x.recv[(size_t)r * x.SL + base + (size_t)fl * rec + within];
    }
  } else {
    const uint32_t t = (uint32_t)(i - nB - nA - nV);
    for (uint32_t r = 0; r < x.world; ++r) sum += x.recv[(size_t)r * x.SL + x.off_tail + t];
    if (t < x.NA * x.NA) dst = d.Sarrow + (size_t)(t / x.NA) * d.ld + x.F6 + t % x.NA;
    else if (t < x.NA * x.NA + 3 * x.NA) { const uint32_t q = t - x.NA * x.NA, w = q / x.NA; dst = (w == 0 ? d.rhsacc : (w == 1 ? d.gB : d.hdiag)) + x.F6 + q % x.NA; }
    else dst = d.scal + (t - x.NA * x.NA - 3 * x.NA);
  }
  *dst = sum;
}

// ---------------------------------------------------------------------------------------------
// options.deterministic = 1: ordered reduction of the per-block window slabs k_sweep3 leaves in HBM (instead of its atomic
// flush).  One thread owns one entry of the reduced block and adds the contributions of the blocks whose frame window covers
// it, in ascending block order: bitwise reproducible, no atomics.  Blocks are in point order, i.e. in non-decreasing first
// frame, and a window spans at most Plan::NF_MAX frames, so the covering blocks of a frame are one short run.
// ---------------------------------------------------------------------------------------------
__global__ void k_det_reduce(Dev d, int mode) {
  const V2Lds lay(d.v2_nfmax, true, 256);
  const uint32_t NFm = lay.nfm, vlen = 6 * NFm + NCMAX + 3, NC = d.nc, NCC = NC * (NC + 1) / 2;
  const uint32_t F = d.F, F6 = 6 * F, BS = (d.bw + 1) * 36, nB = d.n_blocks;
  const uint64_t n1 = (uint64_t)F * BS, n2 = n1 + (uint64_t)NC * F6, n3 = n2 + NCC, n4 = n3 + 3ull * F6, n5 = n4 + 3ull * NC, n6 = n5 + 3;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n6) return;
  const uint32_t camrow = 3 * d.Q;
  // blocks whose window [flo, flo + nf) contains the frames fa <= fb: first / last candidate by first frame
  auto covering = [&](uint32_t fa, uint32_t fb, uint32_t& b0, uint32_t& b1) {
    uint32_t lo = 0, hi = nB;                       // first block with flo > fa
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (d.blk_flo[m] <= fa) lo = m + 1; else hi = m; }
    b1 = lo;                                        // candidates end here (exclusive)
    b0 = b1;
    while (b0 > 0 && d.blk_flo[b0 - 1] + DET_NF_MAX > fb) --b0;
  };
  double sum = 0.0;
  double* dst = nullptr;
  if (t < n1) {                                     // pose x pose band
    if (mode != 0) return;
    const uint32_t f = (uint32_t)(t / BS), r = (uint32_t)(t % BS), dd = r / 36, e = r % 36;
    if (dd > f || (dd == 0 && (e % 6) > (e / 6))) return;
    uint32_t b0, b1; covering(f - dd, f, b0, b1);
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t flo = d.blk_flo[b], nf = d.blk_nf[b];
      if (f >= flo + nf) continue;
      const uint32_t lf = f - flo, lj = f - dd - flo;
      sum += d.det_slab[(size_t)b * d.det_stride + (size_t)(lf * (lf + 1) / 2 + lj) * 36 + e];
    }
    dst = d.Sband + t;
  } else if (t < n2) {                              // camera x pose
    if (mode != 0) return;
    const uint32_t q = (uint32_t)(t - n1), j = q / F6, col = q % F6, f = col / 6;
    uint32_t b0, b1; covering(f, f, b0, b1);
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t flo = d.blk_flo[b], nf = d.blk_nf[b];
      if (f >= flo + nf) continue;
      sum += d.det_slab[(size_t)b * d.det_stride + lay.off_cp + (size_t)j * 6 * NFm + 6 * (f - flo) + col % 6];
    }
    dst = d.Sarrow + (size_t)(camrow + j) * d.ld + col;
  } else if (t < n3) {                              // camera x camera
    if (mode != 0) return;
    return;                                         // every block contributes: k_det_reduce_all (one wave per entry)
  } else if (t < n4) {                              // gB | hdiag | rhs, pose part
    const uint32_t q = (uint32_t)(t - n3), w = q / F6, col = q % F6, f = col / 6;
    if (mode != 0 && w != 1) return;
    uint32_t b0, b1; covering(f, f, b0, b1);
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t flo = d.blk_flo[b], nf = d.blk_nf[b];
      if (f >= flo + nf) continue;
      sum += d.det_slab[(size_t)b * d.det_stride + lay.off_vec + (size_t)w * vlen + 6 * (f - flo) + col % 6];
    }
    dst = (w == 0 ? d.gB : (w == 1 ? d.hdiag : d.rhsacc)) + col;
  } else if (t < n5) {                              // ... camera part
    return;                                         // k_det_reduce_all
  } else {
    return;                                         // k_det_reduce_all
  }
  *dst += sum;
}

// the entries EVERY block contributes to (camera x camera block, camera part of the three vectors, cost / bad-U count / max
// |g_p|): one wave per entry, lane l adds the blocks l, l + 64, ... in order, the 64 partial sums are combined by a fixed
// butterfly — a fixed association, hence bitwise reproducible, without a 255-long chain of dependent loads in one thread
__global__ __launch_bounds__(64) void k_det_reduce_all(Dev d, int mode) {
  const V2Lds lay(d.v2_nfmax, true, 256);
  const uint32_t NFm = lay.nfm, vlen = 6 * NFm + NCMAX + 3, NC = d.nc, NCC = NC * (NC + 1) / 2, nB = d.n_blocks;
  const uint32_t e = blockIdx.x, lane = threadIdx.x;
  const uint32_t camrow = 3 * d.Q, camcol = 6 * d.F + 3 * d.Q;
  uint32_t off; double* dst; bool is_max = false;
  if (e < NCC) {
    if (mode != 0) return;
    uint32_t i = 0; while ((i + 1) * (i + 2) / 2 <= e) ++i;
    off = lay.off_cc + e; dst = d.Sarrow + (size_t)(camrow + i) * d.ld + camcol + (e - i * (i + 1) / 2);
  } else if (e < NCC + 3 * NC) {
    const uint32_t q = e - NCC, w = q / NC, j = q % NC;
    if (mode != 0 && w != 1) return;
    off = lay.off_vec + w * vlen + 6 * NFm + j; dst = (w == 0 ? d.gB : (w == 1 ? d.hdiag : d.rhsacc)) + camcol + j;
  } else {
    if (mode != 0) return;
    const uint32_t q = e - NCC - 3 * NC;
    off = lay.off_fr + q; is_max = q == 2;
    dst = d.scal + (q == 0 ? SCAL_COST : (q == 1 ? SCAL_BAD_U : SCAL_GMAX0 + d.rank));
  }
  double v = 0.0;
  for (uint32_t b = lane; b < nB; b += 64) { const double x = d.det_slab[(size_t)b * d.det_stride + off]; v = is_max ? fmax(v, x) : v + x; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const double o = __shfl_xor(v, m, 64); v = is_max ? fmax(v, o) : v + o; }
  if (lane == 0) { if (is_max) *dst = fmax(*dst, v); else *dst += v; }
}

// per-workgroup partial sums of the value-only kernels, added up in workgroup order (options.deterministic)
__global__ void k_det_sum(const double* slots, uint32_t n_wg, uint32_t K, double* dst) {
  const uint32_t k = threadIdx.x;
  if (k >= K) return;
  double s = 0.0;
  for (uint32_t g = 0; g < n_wg; ++g) s += slots[(size_t)g * K + k];
  dst[k] += s;
}

// ---------------------------------------------------------------------------------------------
// finalize: LM diagonal on the reduced system + rhs row; identity on columns that are not solved for
// ---------------------------------------------------------------------------------------------
__global__ void k_finalize(Dev d, double radius) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) d.lm[LM_T0] = (double)__builtin_amdgcn_s_memrealtime();   // the linear solve starts here (k_lm_control reads the span: no event records in the device loop)
  double g = 0.0;
  if (t < d.n_red) g = finalize_column(d, t, radius);
  // max |g| over the reduced block: one atomic per wave (bit pattern of a non-negative double orders like an integer)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) g = fmax(g, __shfl_xor(g, m, 64));
  if ((threadIdx.x & 63u) == 0) atomicMax((unsigned long long*)(d.step + ST_GMAX_RED), (unsigned long long)__double_as_longlong(g));
}

// ---------------------------------------------------------------------------------------------
// block-banded + arrow Cholesky, single workgroup (the reduced system is small and latency bound).
// Storage: Sband[f][dd] = 6x6 block (pose f rows, pose f-dd cols); Sarrow rows (promoted | camera | rhs)
// over columns (poses | arrow).  The rhs row makes the forward substitution part of the factorisation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_band_chol(Dev d) {
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [0,36) L_jj, [36,72) L_jj^-1, [72] fail, [80..) panel
  double* Ld = smem; double* Li = smem + 36; double* failp = smem + 72;
  double* panel = d.panel_g ? d.panel_g : smem + 80;  // (6*bw + NA + 1) x 6
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t F = d.F, bw = d.bw, NAx = d.NA + 1, ld = d.ld;
  if (tid == 0) *failp = 0.0;
  __syncthreads();
  for (uint32_t j = 0; j < F; ++j) {
    double* Sjj = d.Sband + (size_t)j * (bw + 1) * 36;
    if (tid == 0) {  // 6x6 Cholesky and the inverse of its factor
      double L[6][6];
      for (int a = 0; a < 6; ++a) for (int b = 0; b <= a; ++b) L[a][b] = Sjj[a * 6 + b];
      bool ok = true;
      for (int c = 0; c < 6; ++c) {
        double dg = L[c][c];
        for (int k = 0; k < c; ++k) dg -= L[c][k] * L[c][k];
        if (!(dg > 0.0)) { ok = false; dg = 1.0; }
        dg = sqrt(dg); L[c][c] = dg;
        for (int r = c + 1; r < 6; ++r) { double s = L[r][c]; for (int k = 0; k < c; ++k) s -= L[r][k] * L[c][k]; L[r][c] = s / dg; }
      }
      if (!ok) *failp = 1.0;
      double I[6][6];
      for (int c = 0; c < 6; ++c) {  // columns of L^-1
        for (int r = 0; r < 6; ++r) I[r][c] = 0.0;
        I[c][c] = 1.0 / L[c][c];
        for (int r = c + 1; r < 6; ++r) { double s = 0.0; for (int k = c; k < r; ++k) s -= L[r][k] * I[k][c]; I[r][c] = s / L[r][r]; }
      }
      for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) { Ld[a * 6 + b] = (b <= a) ? L[a][b] : 0.0; Li[a * 6 + b] = (b <= a) ? I[a][b] : 0.0; }
      for (int a = 0; a < 6; ++a) for (int b = 0; b <= a; ++b) Sjj[a * 6 + b] = L[a][b];
      for (int k = 0; k < 36; ++k) d.Linv[(size_t)j * 36 + k] = Li[k];
    }
    __syncthreads();
    // panel rows: band rows of blocks (i, i-j) for i in (j, min(j+bw, F-1)], then the arrow rows
    const uint32_t nbel = min(bw, F - 1 - j);
    const uint32_t nrows = 6 * nbel + NAx;
    for (uint32_t r = tid; r < nrows; r += nt) {
      double* src;
      if (r < 6 * nbel) { const uint32_t i = j + 1 + r / 6; src = d.Sband + ((size_t)i * (bw + 1) + (i - j)) * 36 + (r % 6) * 6; }
      else src = d.Sarrow + (size_t)(r - 6 * nbel) * ld + 6 * j;
      double x[6], y[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = src[k];
#pragma unroll
      for (int c = 0; c < 6; ++c) { double s = 0.0;
#pragma unroll
        for (int k = 0; k <= c; ++k) s += x[k] * Li[c * 6 + k];
        y[c] = s; }
#pragma unroll
      for (int k = 0; k < 6; ++k) { src[k] = y[k]; panel[(size_t)r * 6 + k] = y[k]; }
    }
    __syncthreads();
    // trailing update: S(r, c) -= panel[r] . panel[c] for c <= r (only c inside the band / arrow)
    const uint32_t nb6 = 6 * nbel;
    const uint64_t nband = (uint64_t)nb6 * (nb6 + 1) / 2;
    const uint64_t narrow = (uint64_t)NAx * nb6;
    const uint64_t naa = (uint64_t)NAx * (NAx + 1) / 2;
    for (uint64_t t = tid; t < nband + narrow + naa; t += nt) {
      uint32_t r, cidx; double* dst;
      if (t < nband) {
        r = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((uint64_t)r * (r + 1) / 2 > t) --r;
        while ((uint64_t)(r + 1) * (r + 2) / 2 <= t) ++r;
        cidx = (uint32_t)(t - (uint64_t)r * (r + 1) / 2);
        const uint32_t bi = j + 1 + r / 6, bc = j + 1 + cidx / 6;
        dst = d.Sband + ((size_t)bi * (bw + 1) + (bi - bc)) * 36 + (r % 6) * 6 + (cidx % 6);
      } else if (t < nband + narrow) {
        const uint64_t u = t - nband; const uint32_t a = (uint32_t)(u / nb6); cidx = (uint32_t)(u % nb6); r = nb6 + a;
        dst = d.Sarrow + (size_t)a * ld + 6 * (j + 1) + cidx;
      } else {
        const uint64_t u = t - nband - narrow;
        uint32_t a = (uint32_t)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
        while ((uint64_t)a * (a + 1) / 2 > u) --a;
        while ((uint64_t)(a + 1) * (a + 2) / 2 <= u) ++a;
        const uint32_t b = (uint32_t)(u - (uint64_t)a * (a + 1) / 2);
        r = nb6 + a; cidx = nb6 + b;
        dst = d.Sarrow + (size_t)a * ld + 6 * F + b;
      }
      const double* pr = panel + (size_t)r * 6; const double* pc = panel + (size_t)cidx * 6;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += pr[k] * pc[k];
      *dst -= s;
    }
    __syncthreads();
  }
  // dense Cholesky of the arrow block (NA x NA) with the rhs row carried along
  double* Aa = d.Sarrow + 6 * F;  // Aa[a*ld + b]
  for (uint32_t c = 0; c < d.NA; ++c) {
    if (tid == 0) { double dg = Aa[(size_t)c * ld + c]; if (!(dg > 0.0)) { *failp = 1.0; dg = 1.0; } Aa[(size_t)c * ld + c] = sqrt(dg); }
    __syncthreads();
    const double dg = Aa[(size_t)c * ld + c];
    for (uint32_t r = c + 1 + tid; r < NAx; r += nt) Aa[(size_t)r * ld + c] /= dg;
    __syncthreads();
    const uint32_t m = NAx - c - 1;
    for (uint64_t t = tid; t < (uint64_t)m * (m + 1) / 2; t += nt) {
      uint32_t a = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
      while ((uint64_t)a * (a + 1) / 2 > t) --a;
      while ((uint64_t)(a + 1) * (a + 2) / 2 <= t) ++a;
      const uint32_t b = (uint32_t)(t - (uint64_t)a * (a + 1) / 2);
      const uint32_t r = c + 1 + a, cc2 = c + 1 + b;
      Aa[(size_t)r * ld + cc2] -= Aa[(size_t)r * ld + c] * Aa[(size_t)cc2 * ld + c];
    }
    __syncthreads();
  }
  if (tid == 0) d.step[ST_CHOL_FAIL] = *failp;
}

// backward substitution L^T x = y (y sits in the rhs arrow row), single workgroup
__global__ __launch_bounds__(1024) void k_band_backsolve(Dev d) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t F = d.F, bw = d.bw, NA = d.NA, ld = d.ld;
  double* x = d.delta_red;
  const double* y = d.Sarrow + (size_t)NA * ld;
  const double* Aa = d.Sarrow + 6 * F;
  __shared__ double part6[16][6];
  const uint32_t wv = tid >> 6, nwv = (nt + 63) >> 6;
  for (uint32_t k = tid; k < d.n_red; k += nt) x[k] = y[k];
  __syncthreads();
  // arrow part (dense, lower factor Aa): x_a = (y_a - sum_{b>a} L[b][a] x_b) / L[a][a]
  for (int a = (int)NA - 1; a >= 0; --a) {
    if (tid == 0) x[6 * F + a] /= Aa[(size_t)a * ld + a];
    __syncthreads();
    const double xa = x[6 * F + a];
    for (uint32_t b = tid; b < (uint32_t)a; b += nt) x[6 * F + b] -= Aa[(size_t)a * ld + b] * xa;
    __syncthreads();
  }
  // pose blocks, last to first: x_j = L_jj^-T (y_j - sum_{i>j} L_ij^T x_i - sum_a L_aj^T x_a)
  for (int j = (int)F - 1; j >= 0; --j) {
    const uint32_t nbel = min(bw, F - 1 - (uint32_t)j);
    const uint32_t nrows = 6 * nbel + NA;
    // each thread sums its rows, then a fixed-order wave / workgroup reduction: the solution is replicated across ranks and
    // must come out as the same bits everywhere (no atomics)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t r = tid; r < nrows; r += nt) {
      const double* src; double xv;
      if (r < 6 * nbel) { const uint32_t i = (uint32_t)j + 1 + r / 6; src = d.Sband + ((size_t)i * (bw + 1) + (i - j)) * 36 + (r % 6) * 6; xv = x[6 * i + r % 6]; }
      else { const uint32_t a = r - 6 * nbel; src = d.Sarrow + (size_t)a * ld + 6 * j; xv = x[6 * F + a]; }
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += src[k] * xv;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { acc[k] = wave_sum(acc[k]); if ((tid & 63u) == 0) part6[wv][k] = acc[k]; }
    __syncthreads();
    if (tid == 0) {
      const double* Li = d.Linv + (size_t)j * 36;
      double t6[6], o[6];
      for (int k = 0; k < 6; ++k) { double sum = 0.0; for (uint32_t w = 0; w < nwv; ++w) sum += part6[w][k]; t6[k] = x[6 * j + k] - sum; }
      for (int c = 0; c < 6; ++c) { double s = 0.0; for (int k = c; k < 6; ++k) s += Li[k * 6 + c] * t6[k]; o[c] = s; }  // L^-T t
      for (int k = 0; k < 6; ++k) x[6 * j + k] = o[k];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// step application
// ---------------------------------------------------------------------------------------------
// camera + poses + promoted points: candidate = Plus(x, delta) with box projection (ceres ParameterBlock::Plus)
__global__ void k_update_reduced(Dev d, double* partial) {
  __shared__ double red[4][4];
  const uint32_t F6 = 6 * d.F, camcol = F6 + 3 * d.Q;
  double gtd = 0.0, ddd = 0.0, st2 = 0.0, x2 = 0.0;
  for (uint32_t t = threadIdx.x; t < d.n_red; t += blockDim.x) {
    const double dl = d.delta_red[t];
    gtd += d.gB[t] * dl; ddd += d.lam_red[t] * dl * dl;
    if (t < F6) {
      const double xo = d.views[t], xn = xo + dl;
      d.views_c[t] = xn;
      if (d.use_poses && d.frame_live[t / 6]) { st2 += (xn - xo) * (xn - xo); x2 += xo * xo; }
    }   // promoted points are written by k_backsub (it knows the point id)
  }
  for (uint32_t j = threadIdx.x; j < LIFCAL_BA_MAX_CAMERA_PARAMETERS; j += blockDim.x) {
    const double xo = d.cam[j];
    double xn = xo;
    if (j < d.nc && d.camc->chm[j] != 0.0) xn = xo + d.delta_red[camcol + j];
    if (d.lower && xn < d.lower[j]) xn = d.lower[j];
    if (d.upper && xn > d.upper[j]) xn = d.upper[j];
    d.cam_c[j] = xn;
    st2 += (xn - xo) * (xn - xo); x2 += xo * xo;
  }
  // fixed-order reduction (no atomics: the sums must not depend on scheduling).  The scalars of the reduced part are
  // the same on every rank up to nothing at all, but they steer the host's accept / reject / terminate branches, which
  // contain collectives: only rank 0 contributes them to the all-reduced buffer, so every rank decides on the same bits.
  gtd = wave_sum(gtd); ddd = wave_sum(ddd); st2 = wave_sum(st2); x2 = wave_sum(x2);
  const uint32_t wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) { red[wv][0] = gtd; red[wv][1] = ddd; red[wv][2] = st2; red[wv][3] = x2; }
  __syncthreads();
  if (threadIdx.x < 8) {   // scalars of the candidate step (k_backsub / k_cost accumulate into them next)
    double v = 0.0;
    if (threadIdx.x < 4 && d.rank == 0) v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    partial[threadIdx.x] = v;
  }
}

// points: delta_P = -U^-1 (g_p + W_p delta_B) for eliminated points, reduced solution for promoted ones
// FOUR lanes per point (a quad): the ~10 groups of a regular point are dealt round the quad and the three partial sums meet
// with two quad shuffles — one thread per point walked its groups serially on a third of the chip (45 -> 15 us at the metric point)
__global__ void k_backsub(Dev d, double* partial /* 4 doubles, all-reduced by the host side */) {
  const uint32_t tq = blockIdx.x * blockDim.x + threadIdx.x;   // thread index: promoted points below
  const uint32_t t = tq >> 2, sub = threadIdx.x & 3u;
  double gtd = 0.0, ddd = 0.0, st2 = 0.0, x2 = 0.0;
  if (t < d.n_owned) {
    const uint32_t p = d.owned[t];
    if (d.promoted[p] < 0) {
      const double* acc = d.ptacc + (size_t)p * 36;
      double v[3] = {0.0, 0.0, 0.0};
      if (sub == 0) { v[0] = acc[6]; v[1] = acc[7]; v[2] = acc[8]; }
      const uint32_t ns = d.pt_nslots[p];
      const uint32_t ncon = d.pt_cons0 ? d.pt_cons0[p + 1] - d.pt_cons0[p] : 0;
      if (d.pt_special[p]) {
        if (sub == 0) for (uint32_t bi = 0; bi < 1 + ns + ncon; ++bi) {
          const WBlock B = point_block(d, p, bi, ns);
          for (uint32_t j = 0; j < B.width; ++j) {
            const double dl = d.delta_red[B.base + j];
            v[0] += B.W[j] * dl; v[1] += B.W[B.ldw + j] * dl; v[2] += B.W[2 * B.ldw + j] * dl;
          }
        }
      } else {
        // regular point (LDS-window kernel): the sweep stored only A per lane; W_pose delta_f = R^T A (Gr delta_a + delta_t)
        // with Gr = d(R P)/d(angles) rebuilt from the frame table and the point (k_sweep2's emission, same formulas)
        if (sub == 0) { const WBlock B = point_block(d, p, 0, ns);
          for (uint32_t j = 0; j < B.width; ++j) { const double dl = d.delta_red[B.base + j]; v[0] += B.W[j] * dl; v[1] += B.W[B.ldw + j] * dl; v[2] += B.W[2 * B.ldw + j] * dl; } }
        const double P0 = d.pts[3 * (size_t)p], P1 = d.pts[3 * (size_t)p + 1], P2 = d.pts[3 * (size_t)p + 2];
        const uint32_t s0 = d.pt_slot0[p];
        for (uint32_t k = sub; k < ns; k += 4) {
          const uint32_t sidx = s0 + k, f = d.gid_fr[sidx];
          const double* ft = d.ft + (size_t)f * FRAME_STRIDE;
          const double* A = d.Av + (size_t)sidx * 6;
          const double* dl = d.delta_red + 6 * f;
          double R[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) R[i] = ft[i];
          const double c0 = ft[12], sn0 = ft[13];
          const double Y0 = R[0] * P0 + R[1] * P1 + R[2] * P2, Y1 = R[3] * P0 + R[4] * P1 + R[5] * P2, Y2 = R[6] * P0 + R[7] * P1 + R[8] * P2;
          const double n0 = R[2], n1 = R[5], n2 = R[8];
          // t = Gr delta_a + delta_t, Gr columns: e_x x Y, (0,c0,s0) x Y, R[:,2] x Y
          const double a0 = dl[0], a1 = dl[1], a2 = dl[2];
          const double t0 = (c0 * Y2 - sn0 * Y1) * a1 + (n1 * Y2 - n2 * Y1) * a2 + dl[3];
          const double t1 = -Y2 * a0 + (sn0 * Y0) * a1 + (n2 * Y0 - n0 * Y2) * a2 + dl[4];
          const double t2 = Y1 * a0 + (-c0 * Y0) * a1 + (n0 * Y1 - n1 * Y0) * a2 + dl[5];
          const double u0 = A[0] * t0 + A[1] * t1 + A[2] * t2, u1 = A[1] * t0 + A[3] * t1 + A[4] * t2, u2 = A[2] * t0 + A[4] * t1 + A[5] * t2;
          v[0] += R[0] * u0 + R[3] * u1 + R[6] * u2; v[1] += R[1] * u0 + R[4] * u1 + R[7] * u2; v[2] += R[2] * u0 + R[5] * u1 + R[8] * u2;
        }
      }
      // the quad's partial sums (fixed order: the same bits on every run); the four lanes of a quad took the same branches above
#pragma unroll
      for (int k = 0; k < 3; ++k) { v[k] += __shfl_xor(v[k], 1, 64); v[k] += __shfl_xor(v[k], 2, 64); }
      const double* iv = d.Uinv + 9 * (size_t)p;
      if (sub == 0) for (int k = 0; k < 3; ++k) {
        const double dl = -(iv[3 * k] * v[0] + iv[3 * k + 1] * v[1] + iv[3 * k + 2] * v[2]);
        const double xo = d.pts[3 * (size_t)p + k];
        d.pts_c[3 * (size_t)p + k] = xo + dl;
        d.dP[3 * (size_t)p + k] = dl;
        gtd += acc[6 + k] * dl; ddd += d.lamP[3 * (size_t)p + k] * dl * dl; st2 += dl * dl; x2 += xo * xo;
      }
    }
  }
  // promoted points are replicated: every rank applies the same reduced step; rank 0 accounts for the norms
  if (tq < d.Q) {
    const uint32_t F6 = 6 * d.F;
    const uint32_t pid = d.promoted_ids[tq];
    for (int k = 0; k < 3; ++k) {
      const double dl = d.delta_red[F6 + 3 * tq + k];
      const double xo = d.pts[3 * (size_t)pid + k];
      d.pts_c[3 * (size_t)pid + k] = xo + dl;
      if (d.rank == 0) { st2 += dl * dl; x2 += xo * xo; }
    }
  }
  gtd = wave_sum(gtd); ddd = wave_sum(ddd); st2 = wave_sum(st2); x2 = wave_sum(x2);
  // one set of global atomics per workgroup (same-address atomics serialise in L2)
  __shared__ double part[16][4];
  const uint32_t wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { part[wv][0] = gtd; part[wv][1] = ddd; part[wv][2] = st2; part[wv][3] = x2; }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0; for (uint32_t k = 0; k < nwv; ++k) t += part[k][threadIdx.x];
    if (d.deterministic) d.det_slots[(size_t)blockIdx.x * 4 + threadIdx.x] = t;   // k_det_sum adds the workgroups up in order
    else atomicAdd(partial + threadIdx.x, t);
  }
}

// candidate = Plus(x, t * delta) for an arbitrary step length t (ceres ParameterBlock::Plus incl. box projection), from
// the stored reduced solution and point step; out[0] += |x - candidate|^2, out[1] += |x|^2 over this rank's share of the
// points; the (replicated) camera + pose part is contributed by rank 0 only, so that the all-reduced sums are the same bits
// on every rank (they steer host branches that contain collectives)
__global__ void k_apply_step(Dev d, double t, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t F6 = 6 * d.F, camcol = F6 + 3 * d.Q;
  double st2r = 0.0, x2r = 0.0, st2p = 0.0, x2p = 0.0;
  if (i < F6) {
    const double xo = d.views[i], xn = xo + t * d.delta_red[i];
    d.views_c[i] = xn;
    if (d.use_poses && d.frame_live[i / 6]) { st2r += (xn - xo) * (xn - xo); x2r += xo * xo; }
  }
  if (i < LIFCAL_BA_MAX_CAMERA_PARAMETERS) {
    const double xo = d.cam[i];
    double xn = xo;
    if (i < d.nc && d.camc->chm[i] != 0.0) xn = xo + t * d.delta_red[camcol + i];
    if (d.lower && xn < d.lower[i]) xn = d.lower[i];
    if (d.upper && xn > d.upper[i]) xn = d.upper[i];
    d.cam_c[i] = xn;
    st2r += (xn - xo) * (xn - xo); x2r += xo * xo;
  }
  if (d.use_points) {
    if (i < d.n_owned) {
      const uint32_t p = d.owned[i];
      if (d.promoted[p] < 0) for (int k = 0; k < 3; ++k) {
        const double xo = d.pts[3 * (size_t)p + k], dl = t * d.dP[3 * (size_t)p + k];
        d.pts_c[3 * (size_t)p + k] = xo + dl; st2p += dl * dl; x2p += xo * xo;
      }
    }
    if (i < d.Q) {
      const uint32_t pid = d.promoted_ids[i];
      for (int k = 0; k < 3; ++k) {
        const double xo = d.pts[3 * (size_t)pid + k], dl = t * d.delta_red[F6 + 3 * i + k];
        d.pts_c[3 * (size_t)pid + k] = xo + dl;
        if (d.rank == 0) { st2p += dl * dl; x2p += xo * xo; }
      }
    }
  }
  if (d.rank == 0) { st2p += st2r; x2p += x2r; }
  st2p = wave_sum(st2p); x2p = wave_sum(x2p);
  if (d.deterministic) {   // one slot pair per workgroup, waves added in order; k_det_sum adds the workgroups in order
    __shared__ double part[16][2];
    const uint32_t wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { part[wv][0] = st2p; part[wv][1] = x2p; }
    __syncthreads();
    if (threadIdx.x < 2) { double t2 = 0.0; for (uint32_t k = 0; k < nwv; ++k) t2 += part[k][threadIdx.x]; d.det_slots[(size_t)blockIdx.x * 2 + threadIdx.x] = t2; }
    return;
  }
  if ((threadIdx.x & 63) == 0 && (st2p != 0.0 || x2p != 0.0)) { atomicAdd(out + 0, st2p); atomicAdd(out + 1, x2p); }
}

// directional derivative grad(x_trial) . delta after a sweep at the trial point: point part of this rank (+ the replicated
// reduced part on rank 0 only, see k_apply_step)
__global__ void k_dirderiv(Dev d, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  double dp = 0.0;
  if (d.rank == 0 && i < d.n_red) dp = d.gB[i] * d.delta_red[i];
  if (d.use_points && i < d.n_owned) {
    const uint32_t p = d.owned[i];
    if (d.promoted[p] < 0) { const double* g = d.ptacc + (size_t)p * 36 + 6; for (int k = 0; k < 3; ++k) dp += g[k] * d.dP[3 * (size_t)p + k]; }
  }
  dp = wave_sum(dp);
  if (d.deterministic) {
    __shared__ double part[16];
    const uint32_t wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) part[wv] = dp;
    __syncthreads();
    if (threadIdx.x == 0) { double t2 = 0.0; for (uint32_t k = 0; k < nwv; ++k) t2 += part[k]; d.det_slots[blockIdx.x] = t2; }
    return;
  }
  if ((threadIdx.x & 63) == 0 && dp != 0.0) atomicAdd(out, dp);
}

// largest |component| of the step (line search: ceres' minimum-step test runs on the whole direction vector).  Maxima
// travel as bit patterns in one slot per rank, summed by the all-reduce (the other ranks' slots hold 0): slot[rank] = this
// rank's points, slot[64] = the replicated camera + pose part, written by rank 0 alone
__global__ void k_dir_max(Dev d, unsigned long long* slots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  double mr = 0.0, mp = 0.0;
  if (d.rank == 0 && i < d.n_red) mr = fabs(d.delta_red[i]);
  if (d.use_points && i < d.n_owned) {
    const uint32_t p = d.owned[i];
    if (d.promoted[p] < 0) for (int k = 0; k < 3; ++k) mp = fmax(mp, fabs(d.dP[3 * (size_t)p + k]));
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { mr = fmax(mr, __shfl_xor(mr, m, 64)); mp = fmax(mp, __shfl_xor(mp, m, 64)); }
  if ((threadIdx.x & 63) == 0) {
    if (mr > 0.0) atomicMax(slots + 64, (unsigned long long)__double_as_longlong(mr));
    if (mp > 0.0) atomicMax(slots + d.rank, (unsigned long long)__double_as_longlong(mp));
  }
}

// ---------------------------------------------------------------------------------------------
// Levenberg-Marquardt control on the device (one thread): what lifcal_ba_solve's host loop decides per iteration —
// ceres 2.1 TrustRegionMinimizer: step validity, the three convergence tests, accept / reject, radius update — from the
// scalars the sweep and the candidate evaluation left in HBM.  The host enqueues sweep | linear solve | candidate | this |
// k_lm_commit | next sweep without waiting and reads the state back once per iteration while the next sweep runs.
// Unbounded problems on one rank (bounds need the host's line search, ranks need rank-consistent host decisions).
// ---------------------------------------------------------------------------------------------
LIFCAL_DEV void lm_control_step(const Dev& d, const LmOpts& o, const double* partial) {
  double* lm = d.lm;
  lm[LM_COMMIT] = 0.0;
  if (lm[LM_TERMINATION] != 0.0) return;
  lm[LM_SWEEPS] += 1.0;
  // the sweep at the head of this iteration: cost / gradient norm of a NEW point, or the same point at a new radius
  double bad = d.scal[SCAL_BAD_U];
  if (lm[LM_FRESH] != 0.0) {
    double g = 0.0;
    for (int r = 0; r < 64; ++r) g = fmax(g, d.scal[SCAL_GMAX0 + r]);
    g = fmax(g, d.step[ST_GMAX_RED]);
    lm[LM_X_COST] = d.scal[SCAL_COST]; lm[LM_GMAX] = g;
    if (lm[LM_INITIAL_COST] < 0.0) {   // the first sweep of the solve
      lm[LM_INITIAL_COST] = lm[LM_X_COST];
      if (!(fabs(lm[LM_X_COST]) < 1.7e308)) { lm[LM_TERMINATION] = -1.0; return; }   // non-finite cost at the initial point
      if (g <= o.g_tol) { lm[LM_TERMINATION] = (double)LIFCAL_BA_TERM_GRADIENT_TOLERANCE; return; }
    }
    lm[LM_FRESH] = 0.0;
  } else {
    bad = 0.0;   // (the host loop does not re-read the flag of a re-sweep at a smaller radius either)
  }
  const double x_cost = lm[LM_X_COST];
  // top of the loop
  if (lm[LM_ITER] >= (double)o.max_iterations) { lm[LM_TERMINATION] = (double)LIFCAL_BA_TERM_MAX_ITERATIONS; return; }
  if (lm[LM_STEP_OK] != 0.0 && lm[LM_GMAX] <= o.g_tol) { lm[LM_TERMINATION] = (double)LIFCAL_BA_TERM_GRADIENT_TOLERANCE; return; }
  if (lm[LM_RADIUS] < o.min_radius) { lm[LM_TERMINATION] = (double)LIFCAL_BA_TERM_MIN_RADIUS; return; }
  lm[LM_ITER] += 1.0;
  const double gtd = partial[0], ddd = partial[1], step2 = partial[2], x2 = partial[3];
  double cand_cost = partial[4];
  const double chol_fail = d.step[ST_CHOL_FAIL];
  // model_cost_change = -g^T d - 1/2 d^T J^T J d with (J^T J + Lambda) d = -g  =>  1/2 (d^T Lambda d - g^T d)
  const double mcc = 0.5 * (ddd - gtd);
  const bool valid = chol_fail == 0.0 && bad == 0.0 && fabs(mcc) < 1.7e308 && mcc > 0.0;
  if (!valid) {
    lm[LM_INVALID] += 1.0;
    if (lm[LM_INVALID] >= 5.0) { lm[LM_TERMINATION] = (double)LIFCAL_BA_TERM_INVALID_STEPS; return; }
    lm[LM_RADIUS] *= 0.5; lm[LM_STEP_OK] = 0.0; lm[LM_UNSUCCESSFUL] += 1.0;
    return;
  }
  lm[LM_INVALID] = 0.0;
  if (!(fabs(cand_cost) < 1.7e308)) cand_cost = 1.7976931348623157e308;
  const double step_norm = sqrt(step2), x_norm = sqrt(x2);
  lm[LM_LAST_STEP] = step_norm;
  if (step_norm <= o.p_tol * (x_norm + o.p_tol)) { lm[LM_TERMINATION] = (double)LIFCAL_BA_TERM_PARAMETER_TOLERANCE; return; }
  const double cost_change = x_cost - cand_cost;
  lm[LM_LAST_CHANGE] = cost_change;
  if (fabs(cost_change) <= o.f_tol * x_cost) { lm[LM_TERMINATION] = (double)LIFCAL_BA_TERM_FUNCTION_TOLERANCE; return; }
  const double rel = (cand_cost >= 1.7976931348623157e308) ? -1.7976931348623157e308 : cost_change / mcc;
  lm[LM_LAST_REL] = rel;
  if (rel > o.min_rel_decrease) {
    const double t = 2.0 * rel - 1.0;
    double radius = lm[LM_RADIUS] / fmax(1.0 / 3.0, 1.0 - t * t * t);
    lm[LM_RADIUS] = fmin(o.max_radius, radius);
    lm[LM_DECREASE] = 2.0;
    lm[LM_COMMIT] = 1.0; lm[LM_FRESH] = 1.0; lm[LM_STEP_OK] = 1.0; lm[LM_SUCCESSFUL] += 1.0;
  } else {
    lm[LM_RADIUS] = lm[LM_RADIUS] / lm[LM_DECREASE]; lm[LM_DECREASE] *= 2.0; lm[LM_STEP_OK] = 0.0; lm[LM_UNSUCCESSFUL] += 1.0;
  }
}

// mirror: the state is copied into MAPPED host memory by the kernel itself and the round number is written last (system-scope
// fence in between), so the host follows the loop by polling one word — no copy kernel, no event record (each a barrier packet
// with ~5 us of idle queue) between the kernels of an iteration
__global__ void k_lm_control(Dev d, LmOpts o, const double* partial, double* mirror, double seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  d.lm[LM_TICKS_LINEAR] += (double)__builtin_amdgcn_s_memrealtime() - d.lm[LM_T0];
  lm_control_step(d, o, partial);
  for (int i = 0; i < LM_N; ++i) if (i != LM_SEQ) mirror[i] = d.lm[i];
  __threadfence_system();
  *(volatile double*)(mirror + LM_SEQ) = seq;
}

// an accepted candidate becomes the current point (the host loop swaps pointers; device code keeps its arguments and copies)
__global__ void k_lm_commit(Dev d) {
  if (d.lm[LM_COMMIT] == 0.0) return;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  for (uint32_t i = t; i < LIFCAL_BA_MAX_CAMERA_PARAMETERS; i += n) d.cam[i] = d.cam_c[i];
  for (uint32_t i = t; i < 6 * d.F; i += n) d.views[i] = d.views_c[i];
  if (d.use_points) for (uint32_t i = t; i < 3 * d.P; i += n) d.pts[i] = d.pts_c[i];
}

// ---------------------------------------------------------------------------------------------
// cost at the candidate point (values only), optional directional derivative for the line search
// ---------------------------------------------------------------------------------------------
template <int NR, bool TAN, bool ADJ>
__global__ __launch_bounds__(256) void k_cost(Dev d, TileSet ts, const CamConsts* camc, const double* ft_tab, const double* lt_tab,
                                              const double* pts, double* cost_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
  const CamConsts c = *camc;
  double cost = 0.0, lmant = 1.0; int lexp = 0;
  for (uint32_t tile = wave; tile < ts.n_tiles; tile += n_waves) {
    const uint32_t slot = tile * 64 + lane;
    const uint32_t cnt = ts.slot_cnt[slot];
    const uint32_t row0 = ts.tile_row0[tile], kmax = ts.tile_row0[tile + 1] - row0;
    const double* ft = ft_tab + (size_t)ts.slot_fr[slot] * FRAME_STRIDE;
    const double* P = pts + 3 * (size_t)ts.slot_pt[slot];
    GroupConsts g;
    {
      const double P0 = P[0], P1 = P[1], P2 = P[2];
      const double X = ft[0] * P0 + ft[1] * P1 + ft[2] * P2 + ft[9];
      const double Y = ft[3] * P0 + ft[4] * P1 + ft[5] * P2 + ft[10];
      const double Z = ft[6] * P0 + ft[7] * P1 + ft[8] * P2 + ft[11];
      group_prepare(c, X, Y, Z, g);
    }
    // Eight steps at a time, every load of a stage in flight together (clamped rows: unconditional loads): lens indices, then the
    // four lens values + u, v, then the arithmetic — with a load -> wait -> load chain per step the kernel was two dependent memory
    // round trips per observation step long (22 us at the metric point for 20 MB of input).  The Cauchy cost is accumulated as a
    // running mantissa / exponent product, one log per lane at the end (as in k_sweep3).
    for (uint32_t k0 = 0; k0 < kmax; k0 += 8) {
      uint32_t li[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) li[q] = ts.ell_lens[((size_t)row0 + min(k0 + (uint32_t)q, kmax - 1)) * 64 + lane];
      double2 La[8], Lb[8]; double uu[8], vv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const size_t at = ((size_t)row0 + min(k0 + (uint32_t)q, kmax - 1)) * 64 + lane;
        const double* L = lt_tab + (size_t)li[q] * LENS_STRIDE;
        La[q] = *reinterpret_cast<const double2*>(L); Lb[q] = *reinterpret_cast<const double2*>(L + 2);
        uu[q] = ts.ell_u[at]; vv[q] = ts.ell_v[at];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (k0 + (uint32_t)q < cnt) {
          double rx, ry;
          obs_value<NR, TAN, ADJ>(c, g, La[q].x, La[q].y, Lb[q].x, Lb[q].y, uu[q], vv[q], rx, ry);
          const double sq = rx * rx + ry * ry;
          if (d.robust) { int ex; lmant = frexp(lmant * (1.0 + sq * c.loss_c), &ex); lexp += ex; }
          else cost += 0.5 * sq;
        }
      }
    }
  }
  if (d.robust) cost += 0.5 * c.loss_b * (log(lmant) + (double)lexp * 0.6931471805599453);
  // one global atomic per WORKGROUP: thousands of atomics on one address serialise in L2 (they were most of this kernel's time)
  __shared__ double part[4];
  cost = wave_sum(cost);
  if (lane == 0) part[threadIdx.x >> 6] = cost;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (d.deterministic) d.det_slots[blockIdx.x] = part[0] + part[1] + part[2] + part[3];   // k_det_sum adds the workgroups up in order
    else atomicAdd(cost_out, part[0] + part[1] + part[2] + part[3]);
  }
}

// projected micro-image coordinates of every observation at the stored parameters, scattered back to the caller's
// observation order (reference storeRawImagePointsCsv, src/CameraCalibration.cpp:1504-1538: x_proj, y_proj columns)
template <int NR, bool TAN, bool ADJ>
__global__ __launch_bounds__(256) void k_project_obs(Dev d, TileSet ts, const uint32_t* __restrict__ src, const CamConsts* camc, const double* ft_tab,
                                                     const double* lt_tab, const double* pts, double* __restrict__ xp, double* __restrict__ yp) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
  const CamConsts c = *camc;
  for (uint32_t tile = wave; tile < ts.n_tiles; tile += n_waves) {
    const uint32_t slot = tile * 64 + lane;
    const uint32_t cnt = ts.slot_cnt[slot];
    const uint32_t row0 = ts.tile_row0[tile], kmax = ts.tile_row0[tile + 1] - row0;
    const double* ft = ft_tab + (size_t)ts.slot_fr[slot] * FRAME_STRIDE;
    const double* P = pts + 3 * (size_t)ts.slot_pt[slot];
    GroupConsts g;
    {
      const double P0 = P[0], P1 = P[1], P2 = P[2];
      group_prepare(c, ft[0] * P0 + ft[1] * P1 + ft[2] * P2 + ft[9], ft[3] * P0 + ft[4] * P1 + ft[5] * P2 + ft[10],
                    ft[6] * P0 + ft[7] * P1 + ft[8] * P2 + ft[11], g);
    }
    for (uint32_t k = 0; k < kmax; ++k) {
      if (k < cnt) {
        const size_t at = ((size_t)row0 + k) * 64 + lane;
        const double* L = lt_tab + (size_t)ts.ell_lens[at] * LENS_STRIDE;
        double ex, ey;
        const double u = ts.ell_u[at], v = ts.ell_v[at];
        obs_value<NR, TAN, ADJ>(c, g, L[0], L[1], L[2], L[3], u, v, ex, ey);
        const uint32_t i = src[at];
        xp[i] = ex + u; yp[i] = ey + v;
      }
    }
  }
}

// reprojection statistics (reference src/CameraCalibration.cpp:1026-1103): out = {sum ex^2, sum ey^2, n, inliers}, max as bits
template <int NR, bool TAN, bool ADJ>
__global__ __launch_bounds__(256) void k_stats(Dev d, TileSet ts, const CamConsts* camc, const double* ft_tab, const double* lt_tab,
                                               const double* pts, double thr2, double* sums, unsigned long long* maxbits) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
  const CamConsts c = *camc;
  double sx = 0.0, sy = 0.0, n = 0.0, inl = 0.0, mx = 0.0, my = 0.0;
  for (uint32_t tile = wave; tile < ts.n_tiles; tile += n_waves) {
    const uint32_t slot = tile * 64 + lane;
    const uint32_t cnt = ts.slot_cnt[slot];
    const uint32_t row0 = ts.tile_row0[tile], kmax = ts.tile_row0[tile + 1] - row0;
    const double* ft = ft_tab + (size_t)ts.slot_fr[slot] * FRAME_STRIDE;
    const double* P = pts + 3 * (size_t)ts.slot_pt[slot];
    GroupConsts g;
    {
      const double P0 = P[0], P1 = P[1], P2 = P[2];
      group_prepare(c, ft[0] * P0 + ft[1] * P1 + ft[2] * P2 + ft[9], ft[3] * P0 + ft[4] * P1 + ft[5] * P2 + ft[10],
                    ft[6] * P0 + ft[7] * P1 + ft[8] * P2 + ft[11], g);
    }
    for (uint32_t k = 0; k < kmax; ++k) {
      if (k < cnt) {
        const size_t at = ((size_t)row0 + k) * 64 + lane;
        const double* L = lt_tab + (size_t)ts.ell_lens[at] * LENS_STRIDE;
        double ex, ey;
        obs_value<NR, TAN, ADJ>(c, g, L[0], L[1], L[2], L[3], ts.ell_u[at], ts.ell_v[at], ex, ey);
        sx += ex * ex; sy += ey * ey; n += 1.0;
        if (ex * ex + ey * ey <= thr2) inl += 1.0;
        mx = fmax(mx, fabs(ex)); my = fmax(my, fabs(ey));
      }
    }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); n = wave_sum(n); inl = wave_sum(inl);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { mx = fmax(mx, __shfl_xor(mx, m, 64)); my = fmax(my, __shfl_xor(my, m, 64)); }
  // one set of global atomics per WORKGROUP (same-address atomics serialise in L2)
  __shared__ double part[4][6];
  if (lane == 0) { double* q = part[threadIdx.x >> 6]; q[0] = sx; q[1] = sy; q[2] = n; q[3] = inl; q[4] = mx; q[5] = my; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    if (d.deterministic) d.det_slots[(size_t)blockIdx.x * 4 + threadIdx.x] = t;   // k_det_sum adds the workgroups up in order
    else atomicAdd(sums + threadIdx.x, t);
  } else if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    atomicMax(maxbits + (k - 4), (unsigned long long)__double_as_longlong(fmax(fmax(part[0][k], part[1][k]), fmax(part[2][k], part[3][k]))));
  }
}

}  // namespace lifcal
#include "bandchol.hpp"   // single-wave LDS-window band Cholesky + back-substitution
#include "bandchol2.hpp"  // the same as segment chains: twisted (two-ended) factorisation on two workgroups
#include "bandchol3.hpp"  // block odd-even reduction over many workgroups (long sequences)
