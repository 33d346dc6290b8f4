// plan.hpp — host-side marshalling of LiFCal's bundle-adjustment problem into the HBM layout.
//
// Replaces the residual-block construction loop of the reference
// (src/CameraCalibration.cpp:859-914: one heap-allocated functor + AutoDiffCostFunction + loss per
// micro-image observation, frame-major) by a one-off re-ordering:
//   * observations are sorted point-major, then by frame; every (point, frame) run is a "group"
//     (the ~6 micro images that see one 3D point in one frame share pose, point and camera-frame
//     coordinates, so one GPU lane walks one group);
//   * 64 consecutive groups form a "tile" (one wavefront); observation payload is stored per tile
//     as [k][lane] ("ELL"), so a wave's loads of step k are fully coalesced;
//   * micro-lens centres are de-duplicated into a lens table (the 10-sweep undistortion of
//     reference src/CameraModel.h:92-125 depends on the lens and the camera only);
//   * points are ordered by the first frame that sees them; ranks own contiguous ranges of that
//     order balanced by observation count;
//   * points named as pointID_2 of a distance constraint (reference :916-925) are "promoted" into
//     the reduced system so that the eliminated point blocks stay independent.
// Pure host C++ (no HIP), so the CPU test-suite can exercise it through lifcal_ba_plan().
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <atomic>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/lifcal_ba.h"

namespace lifcal {

// Storage for the large padded [k][lane] arrays: pages come zeroed from the OS (calloc) and value-initialisation is a no-op, so
// a resize does not touch them; the first touch happens in the threads that fill the rows.
template <class T>
struct LazyZeroAlloc {
  using value_type = T;
  LazyZeroAlloc() = default;
  template <class U> LazyZeroAlloc(const LazyZeroAlloc<U>&) {}
  T* allocate(size_t n) { void* q = std::calloc(n ? n : 1, sizeof(T)); if (!q) throw std::bad_alloc(); return (T*)q; }
  void deallocate(T* q, size_t) { std::free(q); }
  template <class U> void construct(U*) noexcept {}
  template <class U, class A0, class... A> void construct(U* q, A0&& a0, A&&... a) { ::new ((void*)q) U(std::forward<A0>(a0), std::forward<A>(a)...); }
  template <class U> bool operator==(const LazyZeroAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const LazyZeroAlloc<U>&) const { return false; }
};
template <class T> using ZeroVec = std::vector<T, LazyZeroAlloc<T>>;

struct Plan {
  // problem sizes
  uint32_t F = 0, P = 0, N = 0, M = 0;
  int rank = 0, world = 1;
  // config
  uint32_t config = 0;
  int n_radial = 0; bool tangential = false, refine_poses = false, robust = false, refine_points = false, adj = false;
  int nc = 5;                       // live camera slots 5 + nRad + 2*tan
  bool use_poses = false, use_points = false, use_constraints = false;
  // point bookkeeping (global ids)
  std::vector<uint32_t> point_order;   // rank in the (first frame, id) order -> point id
  std::vector<int32_t> owner;          // per point: owning rank (-1: not observed / unused)
  std::vector<int32_t> promoted;       // per point: arrow index q or -1
  std::vector<uint32_t> promoted_ids;
  std::vector<uint8_t> frame_used, point_used;
  uint32_t Q = 0;                      // number of promoted points
  uint32_t bw = 0;                     // block bandwidth: max over points of (last frame - first frame)
  uint32_t NA = 0;                     // arrow rows: 3Q + nc
  uint32_t n_red_int = 0;              // 6F + NA  (internal ordering: poses | promoted | camera)
  uint32_t n_red_canon = 0;            // 17 + 6F + 3Q (canonical ordering of the C ABI)
  // local (this rank) observation layout
  uint32_t n_obs_local = 0, n_groups = 0, n_tiles = 0, n_slots = 0, max_group_obs = 0;
  std::vector<uint32_t> obs_order;     // sorted position -> input index (local obs only)
  std::vector<uint32_t> slot_pt, slot_fr, slot_cnt;   // per slot (tile*64+lane); cnt 0 = idle lane
  std::vector<uint32_t> tile_row0;     // n_tiles+1: first 64-wide row of each tile in the ELL payload
  std::vector<double> ell_u, ell_v;    // rows*64
  std::vector<uint32_t> ell_lens;      // rows*64
  std::vector<uint32_t> ell_src;       // rows*64: input index of the observation (UINT32_MAX = padding)
  // per point: its slots are contiguous [pt_slot0[p], pt_slot0[p]+pt_nslots[p])
  std::vector<uint32_t> pt_slot0, pt_nslots;
  std::vector<uint32_t> owned_points;  // points this rank owns (observed or constrained), in point_order
  // "group id" (gid): running index of (point, frame) groups in point-major order over BOTH paths; the W_pose
  // staging in HBM (18 doubles per group) is indexed by gid, a point's groups are contiguous.
  std::vector<uint32_t> gid_fr;        // frame of each group
  std::vector<uint32_t> slot_gid;      // v1 path: gid of each slot
  // ---- v2 (LDS-window) path: regular points only -------------------------------------------------
  // block  = one workgroup: contiguous range of points whose frames fit a window of <= NF_MAX frames
  // pass   = <= pass_lanes (256 or 128) lanes / <= pass_lanes / 4 points of a block, in tiles of 64 lanes
  static constexpr uint32_t NF_MAX = 20, NP_MAX = 64, PASS_GROUPS = 256, ZD_DOUBLES = 8192;
  uint32_t pass_lanes = PASS_GROUPS;   // lanes per pass of this plan: 256 (four tiles), 128 (two tiles, half the Z matrix: k_sweep3 with two waves per role) or 64 (one tile of whole points: k_front4 + k_back4, sweep4.hpp)
  uint32_t pass_tiles() const { return pass_lanes / 64; }
  uint32_t zd_doubles() const { return pass_lanes >= 256 ? ZD_DOUBLES : ZD_DOUBLES / 2; }
  uint32_t np_max() const { return pass_lanes == 64 ? 64u : pass_lanes / 4; }   // (64-lane passes: no Z matrix per pass, a tile may hold 64 one-lane points)
  uint32_t n_blocks = 0, n_passes = 0, max_block_nf = 0;
  std::vector<uint8_t> pt_special;     // P: the point takes the v1 (global-atomic) kernels
  std::vector<uint32_t> rk_flo, rk_nfr;   // per rank: first frame and number of frames its points observe (identical on every rank)
  uint32_t n_pairs = 0;                // (point, frame) pairs with observations; n_groups counts LANES (pairs after splitting)
  std::vector<uint32_t> blk_pass0;     // n_blocks+1
  std::vector<uint32_t> blk_flo, blk_nf;
  std::vector<uint32_t> pass_pt0, pass_np, pass_gid0, pass_ng;   // per pass: first entry in v2_points, #points, first gid, #groups
  std::vector<uint32_t> v2_points;     // point ids in processing order
  std::vector<uint32_t> v2_ptinfo;     // per entry of v2_points: first pass-local group | number of groups << 16
  std::vector<uint32_t> v2_gidx;       // per (pass*256 + wave*64 + lane): group id of the lane (index of its A block for the back-substitution)
  std::vector<uint32_t> v2_passpt;     // per (pass*64 + k): k-th point of the pass (0-padded): descriptor-free lookup for the factor phase
  std::vector<uint32_t> v2_slot;       // per (pass*256 + wave*64 + lane): cnt | lf<<8 | lp<<16 | rep<<24 ; 0 = idle
  std::vector<uint32_t> v2f_pt, v2f_fr, v2f_cnt;   // the same slots, flat (value-only kernels: cost, statistics)
  std::vector<uint32_t> v2_tile_row0;  // 4*n_passes+1
  ZeroVec<double> v2_u, v2_v; ZeroVec<uint32_t> v2_lens; std::vector<uint32_t> v2_src;
  ZeroVec<float> v2_du, v2_dv;         // options.precision = 1 only: (u - mcx, v - mcy) of the same rows in fp32
  std::vector<uint32_t> special_owned; // owned points handled by the v1 kernels (promoted / constrained / oversized)
  uint32_t n_obs_v2 = 0;
  // lenses
  std::vector<double> lens_xy;         // 2 per lens
  uint32_t n_lenses = 0;
  // constraints (global ids); processed by the rank owning c_i's point
  std::vector<uint32_t> c_i, c_j; std::vector<double> c_dist, c_sigma;
  std::vector<uint32_t> my_constraints;          // indices into c_*
  std::vector<uint32_t> pt_cons0, pt_cons_list;  // CSR over points: constraints in which the point is c_i and eliminated
};

inline int plan_validate(const lifcal_ba_problem* p) {
  if (!p || !p->cam) return LIFCAL_BA_ERR_INVALID_ARG;
  if (p->n_obs && (!p->u || !p->v || !p->mcx || !p->mcy || !p->pt || !p->fr)) return LIFCAL_BA_ERR_INVALID_ARG;
  if ((p->n_frames && !p->views) || (p->n_points && !p->pts)) return LIFCAL_BA_ERR_INVALID_ARG;
  if (!(p->scale > 0.0) || !(p->spx > 0.0) || !(p->spy > 0.0)) return LIFCAL_BA_ERR_INVALID_ARG;
  for (uint32_t i = 0; i < p->n_obs; ++i)
    if (p->pt[i] >= p->n_points || p->fr[i] >= p->n_frames) return LIFCAL_BA_ERR_OUT_OF_RANGE;
  if (p->n_constraints && p->use_constraints) {
    if (!p->c_i || !p->c_j || !p->c_dist || !p->c_sigma) return LIFCAL_BA_ERR_INVALID_ARG;
    for (uint32_t c = 0; c < p->n_constraints; ++c) {
      if (p->c_i[c] >= p->n_points || p->c_j[c] >= p->n_points) return LIFCAL_BA_ERR_OUT_OF_RANGE;
      if (p->c_i[c] == p->c_j[c]) return LIFCAL_BA_ERR_INVALID_ARG;
    }
  }
  return 0;
}

// The per-pass layout work (sorting a pass's lanes, gathering its observations into the [k][lane] rows) is independent
// from pass to pass: a few host threads share it (LIFCAL_PLAN_THREADS overrides the default of min(8, cores)).
template <class F>
inline void plan_parallel_for(uint32_t n, uint32_t grain, F&& fn) {
  unsigned hw = std::thread::hardware_concurrency();
  unsigned T = std::min(8u, hw ? hw : 1u);
  if (const char* e = getenv("LIFCAL_PLAN_THREADS")) T = (unsigned)std::max(1, atoi(e));
  if (T <= 1 || n <= grain) { for (uint32_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<uint32_t> next{0};
  auto work = [&]() {
    for (;;) {
      const uint32_t b = next.fetch_add(grain);
      if (b >= n) break;
      for (uint32_t i = b; i < std::min(n, b + grain); ++i) fn(i);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < T; ++t) {
    try { pool.emplace_back(work); } catch (...) { break; }   // no more threads to be had: the ones running share the work
  }
  work();
  for (std::thread& t : pool) t.join();
}

// LIFCAL_PLAN_TIMING=1: wall time of each planner phase on stderr
// n items cut into T contiguous chunks, chunk t worked by thread t: fn(t, begin, end).  Returns T (1: ran inline).  The serial
// front of the planner (extents, observation sort, lens de-duplication, group runs) uses it with per-chunk partial results that
// are merged IN CHUNK ORDER, so every array comes out exactly as the one-thread code wrote it (LIFCAL_PLAN_HASH).
inline unsigned plan_threads() {
  unsigned hw = std::thread::hardware_concurrency();
  unsigned T = std::min(8u, hw ? hw : 1u);
  if (const char* e = getenv("LIFCAL_PLAN_THREADS")) T = (unsigned)std::max(1, atoi(e));
  return T;
}
template <class F>
inline void plan_parallel_chunks(uint32_t n, unsigned T, F&& fn) {
  if (T <= 1) { fn(0u, 0u, n); return; }
  auto span = [&](unsigned t) { return std::pair<uint32_t, uint32_t>((uint32_t)((uint64_t)n * t / T), (uint32_t)((uint64_t)n * (t + 1) / T)); };
  std::vector<std::thread> pool;
  std::vector<uint8_t> started(T, 0);
  for (unsigned t = 1; t < T; ++t) {
    try { pool.emplace_back([&, t]() { const auto r = span(t); fn(t, r.first, r.second); }); started[t] = 1; } catch (...) { break; }
  }
  { const auto r = span(0); fn(0u, r.first, r.second); }
  for (std::thread& th : pool) th.join();
  for (unsigned t = 1; t < T; ++t) if (!started[t]) { const auto r = span(t); fn(t, r.first, r.second); }   // (no thread to be had: the caller works the chunk)
}

struct PlanClock {
  bool on = getenv("LIFCAL_PLAN_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[plan] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

inline int build_plan(const lifcal_ba_problem* p, int rank, int world, Plan* pl, bool enable_v2 = true, uint32_t target_blocks = 256, uint32_t split_obs = UINT32_MAX, bool frame_order = false,
                      uint32_t pass_lanes = Plan::PASS_GROUPS, bool want_f32 = false, const lifcal_ba_partition* part = nullptr) {
  PlanClock clk;
  if (int rc = plan_validate(p)) return rc;
  clk.lap("validate");
  if (world < 1 || rank < 0 || rank >= world) return LIFCAL_BA_ERR_INVALID_ARG;
  Plan& L = *pl;
  if (pass_lanes != 64 && pass_lanes != 128 && pass_lanes != 256) return LIFCAL_BA_ERR_INVALID_ARG;
  L.pass_lanes = pass_lanes;
  const uint32_t PL = L.pass_lanes, PT = L.pass_tiles();
  L.F = p->n_frames; L.P = p->n_points; L.N = p->n_obs; L.M = p->n_constraints;
  L.rank = rank; L.world = world; L.config = p->config;
  L.n_radial = p->config & 3; L.tangential = (p->config & LIFCAL_BA_CFG_TANGENTIAL) != 0;
  L.refine_poses = (p->config & LIFCAL_BA_CFG_REFINE_POSES) != 0;
  L.robust = (p->config & LIFCAL_BA_CFG_ROBUST) != 0;
  L.refine_points = (p->config & LIFCAL_BA_CFG_REFINE_POINTS) != 0;
  L.adj = (p->config & LIFCAL_BA_CFG_ML_CENTER_ADJ) != 0;
  L.nc = 5 + L.n_radial + (L.tangential ? 2 : 0);
  L.use_poses = L.refine_poses;
  L.use_points = L.refine_poses && L.refine_points;  // reference :879-912: points are blocks only in <2,17,6,3>
  L.use_constraints = L.use_points && p->use_constraints && p->n_constraints > 0;

  // --- per point: first / last frame, observation count ---
  std::vector<uint32_t> first(L.P, UINT32_MAX), last(L.P, 0), cnt(L.P, 0);
  L.frame_used.assign(L.F, 0); L.point_used.assign(L.P, 0);
  const unsigned TP = (L.N >= (1u << 16)) ? plan_threads() : 1u;   // threads of the chunked phases (small problems: inline)
  if (TP <= 1) {
    for (uint32_t i = 0; i < L.N; ++i) {
      const uint32_t q = p->pt[i], f = p->fr[i];
      first[q] = std::min(first[q], f); last[q] = std::max(last[q], f); cnt[q]++;
      L.frame_used[f] = 1; L.point_used[q] = 1;
    }
  } else {
    // per-chunk extents, merged per point (min / max / sum: the order of the chunks does not matter)
    std::vector<std::vector<uint32_t>> pf(TP), pl(TP), pc(TP);
    std::vector<std::vector<uint8_t>> fu(TP);
    plan_parallel_chunks(L.N, TP, [&](unsigned t, uint32_t b, uint32_t e) {
      pf[t].assign(L.P, UINT32_MAX); pl[t].assign(L.P, 0); pc[t].assign(L.P, 0); fu[t].assign(L.F, 0);
      uint32_t* f1 = pf[t].data(); uint32_t* l1 = pl[t].data(); uint32_t* c1 = pc[t].data(); uint8_t* u1 = fu[t].data();
      for (uint32_t i = b; i < e; ++i) {
        const uint32_t q = p->pt[i], f = p->fr[i];
        f1[q] = std::min(f1[q], f); l1[q] = std::max(l1[q], f); c1[q]++; u1[f] = 1;
      }
    });
    plan_parallel_chunks(L.P, TP, [&](unsigned, uint32_t b, uint32_t e) {
      for (uint32_t q = b; q < e; ++q) {
        uint32_t f0 = UINT32_MAX, l0 = 0, c0 = 0;
        for (unsigned t = 0; t < TP; ++t) { f0 = std::min(f0, pf[t][q]); l0 = std::max(l0, pl[t][q]); c0 += pc[t][q]; }
        first[q] = f0; last[q] = l0; cnt[q] = c0; L.point_used[q] = c0 ? 1 : 0;
      }
    });
    for (uint32_t f = 0; f < L.F; ++f) { uint8_t u = 0; for (unsigned t = 0; t < TP; ++t) u |= fu[t][f]; L.frame_used[f] = u; }
  }
  L.bw = 0;
  for (uint32_t q = 0; q < L.P; ++q) if (cnt[q]) L.bw = std::max(L.bw, last[q] - first[q]);
  if (part) {
    // shard-local problem (lifcal_ba_create_shard): `p` holds the observations of this rank's points only; what depends on the
    // other ranks' observations comes from the partition computed once from the index arrays of the whole problem
    if (part->world_size != (uint32_t)world || part->n_frames != L.F || part->n_points != L.P || !part->point_owner || !part->rank_first ||
        !part->rank_frames || !part->frame_used || part->band_width < L.bw) return LIFCAL_BA_ERR_INVALID_ARG;
    if ((p->config & LIFCAL_BA_CFG_REFINE_POINTS) && p->use_constraints && p->n_constraints > 0) return LIFCAL_BA_ERR_INVALID_ARG;   // constraints couple ranks
    for (uint32_t i = 0; i < L.N; ++i) if (part->point_owner[p->pt[i]] != rank) return LIFCAL_BA_ERR_OUT_OF_RANGE;   // not this rank's observation
    L.bw = part->band_width;
    for (uint32_t f = 0; f < L.F; ++f) L.frame_used[f] = part->frame_used[f] ? 1 : 0;
    for (uint32_t q = 0; q < L.P; ++q) L.point_used[q] = part->point_owner[q] >= 0 ? 1 : 0;
  }

  clk.lap("per-point extents");
  // --- constraints / promotion ---
  L.promoted.assign(L.P, -1); L.promoted_ids.clear();
  L.c_i.clear(); L.c_j.clear(); L.c_dist.clear(); L.c_sigma.clear();
  if (L.use_constraints) {
    std::vector<uint8_t> flag(L.P, 0);
    for (uint32_t c = 0; c < L.M; ++c) {
      L.c_i.push_back(p->c_i[c]); L.c_j.push_back(p->c_j[c]); L.c_dist.push_back(p->c_dist[c]); L.c_sigma.push_back(p->c_sigma[c]);
      flag[p->c_j[c]] = 1; L.point_used[p->c_i[c]] = 1; L.point_used[p->c_j[c]] = 1;
    }
    for (uint32_t q = 0; q < L.P; ++q) if (flag[q]) { L.promoted[q] = (int32_t)L.promoted_ids.size(); L.promoted_ids.push_back(q); }
  }
  L.Q = (uint32_t)L.promoted_ids.size();
  L.NA = 3 * L.Q + (uint32_t)L.nc;
  L.n_red_int = 6 * L.F + L.NA;
  L.n_red_canon = LIFCAL_BA_MAX_CAMERA_PARAMETERS + 6 * L.F + 3 * L.Q;

  clk.lap("constraints/promotion");
  // --- point order and ownership ---
  L.point_order.resize(L.P);
  std::iota(L.point_order.begin(), L.point_order.end(), 0u);
  std::stable_sort(L.point_order.begin(), L.point_order.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
  L.owner.assign(L.P, -1);
  if (part) {
    for (uint32_t q = 0; q < L.P; ++q) L.owner[q] = part->point_owner[q];
  } else {
    const uint64_t total = L.N;
    uint64_t acc = 0;
    for (uint32_t r = 0; r < L.P; ++r) {
      const uint32_t q = L.point_order[r];
      if (!L.point_used[q]) continue;
      // rank k owns the points whose running observation count falls in [k, k+1) * total / world
      int o = total ? (int)std::min<uint64_t>((uint64_t)world - 1, acc * (uint64_t)world / std::max<uint64_t>(total, 1)) : 0;
      L.owner[q] = o;
      acc += cnt[q];
    }
  }
  L.owned_points.clear();
  for (uint32_t r = 0; r < L.P; ++r) { const uint32_t q = L.point_order[r]; if (L.owner[q] == rank) L.owned_points.push_back(q); }
  if (part) {
    L.rk_flo.assign(part->rank_first, part->rank_first + world); L.rk_nfr.assign(part->rank_frames, part->rank_frames + world);
  } else {
    std::vector<uint32_t> lo(world, UINT32_MAX), hi(world, 0);
    for (uint32_t q = 0; q < L.P; ++q) if (cnt[q] && L.owner[q] >= 0) { lo[L.owner[q]] = std::min(lo[L.owner[q]], first[q]); hi[L.owner[q]] = std::max(hi[L.owner[q]], last[q]); }
    L.rk_flo.assign(world, 0); L.rk_nfr.assign(world, 0);
    for (int r = 0; r < world; ++r) if (lo[r] != UINT32_MAX) { L.rk_flo[r] = lo[r]; L.rk_nfr[r] = hi[r] - lo[r] + 1; }
  }

  clk.lap("point order");
  // --- local observations sorted by (point order, frame) ---
  std::vector<uint32_t> order_rank(L.P, 0);
  for (uint32_t r = 0; r < L.P; ++r) order_rank[L.point_order[r]] = r;
  {
    // stable counting sort by the point's rank, then each point's short run by frame (stable: input order breaks ties)
    std::vector<uint32_t> start(L.P + 1, 0);
    if (TP <= 1) {
      for (uint32_t i = 0; i < L.N; ++i) if (L.owner[p->pt[i]] == rank) ++start[order_rank[p->pt[i]] + 1];
      for (uint32_t r = 0; r < L.P; ++r) start[r + 1] += start[r];
      L.obs_order.assign(start[L.P], 0);
      std::vector<uint32_t> fill(start.begin(), start.end() - 1);
      for (uint32_t i = 0; i < L.N; ++i) if (L.owner[p->pt[i]] == rank) L.obs_order[fill[order_rank[p->pt[i]]]++] = i;
    } else {
      // the same stable counting sort with per-chunk histograms: chunk t's observations of a point go behind those of the chunks
      // before it, in input order inside the chunk
      std::vector<std::vector<uint32_t>> hist(TP);
      plan_parallel_chunks(L.N, TP, [&](unsigned t, uint32_t b, uint32_t e) {
        hist[t].assign(L.P, 0);
        uint32_t* h = hist[t].data();
        for (uint32_t i = b; i < e; ++i) if (L.owner[p->pt[i]] == rank) ++h[order_rank[p->pt[i]]];
      });
      for (uint32_t r = 0; r < L.P; ++r) {
        uint32_t at = start[r];
        for (unsigned t = 0; t < TP; ++t) { const uint32_t c = hist[t][r]; hist[t][r] = at; at += c; }   // histogram -> first slot of the chunk's run
        start[r + 1] = at;
      }
      L.obs_order.assign(start[L.P], 0);
      plan_parallel_chunks(L.N, TP, [&](unsigned t, uint32_t b, uint32_t e) {
        uint32_t* fill = hist[t].data();
        for (uint32_t i = b; i < e; ++i) if (L.owner[p->pt[i]] == rank) L.obs_order[fill[order_rank[p->pt[i]]]++] = i;
      });
    }
    plan_parallel_chunks(L.P, TP, [&](unsigned, uint32_t rb, uint32_t re) {
      for (uint32_t r = rb; r < re; ++r) {
        uint32_t* b = L.obs_order.data() + start[r]; uint32_t* e = L.obs_order.data() + start[r + 1];
        bool sorted = true;
        for (uint32_t* q = b; q + 1 < e; ++q) if (p->fr[q[1]] < p->fr[q[0]]) { sorted = false; break; }
        if (!sorted) std::stable_sort(b, e, [&](uint32_t a, uint32_t c) { return p->fr[a] < p->fr[c]; });
      }
    });
  }
  L.n_obs_local = (uint32_t)L.obs_order.size();

  clk.lap("obs sort");
  // --- lenses: exact-bit de-duplication of (mcx, mcy) over the local observations ---
  struct Key { uint64_t a, b; bool operator==(const Key& o) const { return a == o.a && b == o.b; } };
  struct KeyHash {   // the low bits index the table: finish with an avalanche step (doubles share most of their low mantissa bits)
    size_t operator()(const Key& k) const {
      uint64_t h = k.a * 0x9E3779B97F4A7C15ull ^ (k.b * 0xC2B2AE3D27D4EB4Full + (k.a >> 31));
      h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
      return (size_t)h;
    }
  };
  // open-addressing table (linear probing, load <= 1/2, doubles when full): ids in order of first appearance
  struct LensTable {
    std::vector<double> xy; std::vector<uint32_t> table; size_t cap = 0;
    explicit LensTable(size_t c0 = 1 << 15) : table(c0, UINT32_MAX), cap(c0) {}
    Key key_of(uint32_t id) const { Key k; std::memcpy(&k.a, &xy[2 * (size_t)id], 8); std::memcpy(&k.b, &xy[2 * (size_t)id + 1], 8); return k; }
    void grow() {
      cap *= 2;
      table.assign(cap, UINT32_MAX);
      KeyHash hash;
      for (uint32_t id = 0; id < (uint32_t)(xy.size() / 2); ++id) {
        size_t at = hash(key_of(id)) & (cap - 1);
        while (table[at] != UINT32_MAX) at = (at + 1) & (cap - 1);
        table[at] = id;
      }
    }
    uint32_t find_or_insert(double x, double y) {
      Key k; std::memcpy(&k.a, &x, 8); std::memcpy(&k.b, &y, 8);
      KeyHash hash;
      size_t at = hash(k) & (cap - 1);
      uint32_t id;
      while ((id = table[at]) != UINT32_MAX && !(key_of(id) == k)) at = (at + 1) & (cap - 1);
      if (id == UINT32_MAX) {
        id = (uint32_t)(xy.size() / 2);
        table[at] = id;
        xy.push_back(x); xy.push_back(y);
        if ((size_t)(id + 1) * 2 > cap) grow();
      }
      return id;
    }
  };
  std::vector<uint32_t> obs_lens(L.n_obs_local);
  L.lens_xy.clear();
  {
    // chunk t numbers the lenses of its stretch of the sorted observations in order of first appearance; the chunks' lists are
    // then entered into ONE table in chunk order — the global order of first appearance, as a single pass would number them —
    // and the observations' local ids are translated
    const unsigned TL = (L.n_obs_local >= (1u << 16)) ? TP : 1u;
    std::vector<LensTable> local;
    local.reserve(TL);
    for (unsigned t = 0; t < TL; ++t) local.emplace_back(TL > 1 ? (size_t)1 << 13 : (size_t)1 << 15);
    std::vector<std::pair<uint32_t, uint32_t>> span(TL);
    plan_parallel_chunks(L.n_obs_local, TL, [&](unsigned t, uint32_t b, uint32_t e) {
      span[t] = {b, e};
      LensTable& lt = local[t];
      for (uint32_t sidx = b; sidx < e; ++sidx) { const uint32_t i = L.obs_order[sidx]; obs_lens[sidx] = lt.find_or_insert(p->mcx[i], p->mcy[i]); }
    });
    if (TL <= 1) {
      L.lens_xy.swap(local[0].xy);
    } else {
      LensTable global;
      std::vector<std::vector<uint32_t>> remap(TL);
      for (unsigned t = 0; t < TL; ++t) {
        const uint32_t nl = (uint32_t)(local[t].xy.size() / 2);
        remap[t].resize(nl);
        for (uint32_t id = 0; id < nl; ++id) remap[t][id] = global.find_or_insert(local[t].xy[2 * (size_t)id], local[t].xy[2 * (size_t)id + 1]);
      }
      plan_parallel_chunks(TL, TL, [&](unsigned t, uint32_t, uint32_t) {
        const uint32_t* m = remap[t].data();
        for (uint32_t sidx = span[t].first; sidx < span[t].second; ++sidx) obs_lens[sidx] = m[obs_lens[sidx]];
      });
      L.lens_xy.swap(global.xy);
    }
  }
  L.n_lenses = (uint32_t)(L.lens_xy.size() / 2);

  clk.lap("lens dedup");
  // --- groups (runs of equal (point, frame)), gid numbering ---
  struct Group { uint32_t pt, fr, s0, n; };
  std::vector<Group> groups;
  {
    // run starts per chunk (an observation whose (point, frame) differs from its predecessor's), concatenated in chunk order
    const unsigned TG = (L.n_obs_local >= (1u << 16)) ? TP : 1u;
    std::vector<std::vector<uint32_t>> heads(TG);
    plan_parallel_chunks(L.n_obs_local, TG, [&](unsigned t, uint32_t b, uint32_t e) {
      std::vector<uint32_t>& h = heads[t];
      h.reserve((e - b) / 3 + 16);
      for (uint32_t sidx = b; sidx < e; ++sidx) {
        const uint32_t i = L.obs_order[sidx];
        if (sidx == 0) { h.push_back(0); continue; }
        const uint32_t j = L.obs_order[sidx - 1];
        if (p->pt[j] != p->pt[i] || p->fr[j] != p->fr[i]) h.push_back(sidx);
      }
    });
    std::vector<uint32_t> off(TG + 1, 0);
    for (unsigned t = 0; t < TG; ++t) off[t + 1] = off[t] + (uint32_t)heads[t].size();
    groups.resize(off[TG]);
    plan_parallel_chunks(TG, TG, [&](unsigned t, uint32_t, uint32_t) {
      const std::vector<uint32_t>& h = heads[t];
      for (size_t k = 0; k < h.size(); ++k) {
        const uint32_t s0 = h[k];
        uint32_t next;   // the next run's start: in this chunk, in the next non-empty chunk, or the end
        if (k + 1 < h.size()) next = h[k + 1];
        else { next = L.n_obs_local; for (unsigned u = t + 1; u < TG; ++u) if (!heads[u].empty()) { next = heads[u][0]; break; } }
        const uint32_t i = L.obs_order[s0];
        groups[off[t] + k] = Group{p->pt[i], p->fr[i], s0, next - s0};
      }
    });
  }
  L.n_groups = (uint32_t)groups.size();
  L.gid_fr.resize(L.n_groups);
  L.pt_slot0.assign(L.P, 0); L.pt_nslots.assign(L.P, 0);
  for (uint32_t g = 0; g < L.n_groups; ++g) {
    L.gid_fr[g] = groups[g].fr;
    if (L.pt_nslots[groups[g].pt] == 0) L.pt_slot0[groups[g].pt] = g;
    L.pt_nslots[groups[g].pt]++;
  }
  L.max_group_obs = 0;
  for (const Group& G : groups) L.max_group_obs = std::max(L.max_group_obs, G.n);

  clk.lap("groups");
  // --- classify points: regular (v2) or special (v1 fallback) ---
  std::vector<uint8_t> special(L.P, 0);
  if (L.use_constraints) for (uint32_t c = 0; c < L.M; ++c) { special[L.c_i[c]] = 1; special[L.c_j[c]] = 1; }
  for (uint32_t q = 0; q < L.P; ++q) {
    if (!cnt[q]) continue;
    if (!enable_v2 || !L.use_points) special[q] = 1;                    // camera-only / pose-only arities: v1 kernels
    if (last[q] - first[q] + 1 > Plan::NF_MAX || L.pt_nslots[q] > PL) special[q] = 1;
  }
  for (const Group& G : groups) if (G.n > 255) special[G.pt] = 1;   // the v2 slot word keeps the group size in 8 bits

  clk.lap("classify");
  // --- v2 blocks: contiguous ranges of regular points with ~reg_obs/target observations each, window <= NF_MAX ---
  std::vector<uint32_t> reg;                 // regular points in point order
  std::vector<size_t> blk_begin;             // block b = reg[blk_begin[b] .. blk_begin[b+1])
  L.blk_flo.clear(); L.blk_nf.clear(); L.max_block_nf = 0;
  L.n_obs_v2 = 0;
  {
    uint64_t reg_obs = 0;
    for (uint32_t q : L.owned_points) if (!special[q] && L.pt_nslots[q] > 0) { reg.push_back(q); reg_obs += cnt[q]; }
    L.n_obs_v2 = (uint32_t)reg_obs;
    const uint64_t per_block = std::max<uint64_t>(1, (reg_obs + target_blocks - 1) / std::max(1u, target_blocks));
    size_t i = 0;
    while (i < reg.size()) {
      uint32_t flo = UINT32_MAX, fhi = 0; uint64_t obs = 0; size_t j = i;
      while (j < reg.size()) {
        const uint32_t q = reg[j];
        const uint32_t nlo = std::min(flo, first[q]), nhi = std::max(fhi, last[q]);
        if (j > i && (nhi - nlo + 1 > Plan::NF_MAX || obs >= per_block)) break;
        flo = nlo; fhi = nhi; obs += cnt[q]; ++j;
      }
      blk_begin.push_back(i);
      L.blk_flo.push_back(flo); L.blk_nf.push_back(fhi - flo + 1);
      L.max_block_nf = std::max(L.max_block_nf, fhi - flo + 1);
      i = j;
    }
    blk_begin.push_back(reg.size());
  }
  auto block_np_cap = [&](size_t b) {
    if (PL == 64) return L.np_max();   // k_front4 keeps no Z matrix per pass (k_back4 chunks the block's points itself)
    // the dense Z matrix of a pass (3 np rows x padded window columns) must fit its LDS budget
    const uint32_t ncolp = ((6 * L.blk_nf[b] + (uint32_t)L.nc + 1) + 15u) & ~15u;
    return std::max(1u, std::min<uint32_t>(L.np_max(), ((L.zd_doubles() / (ncolp + 2)) & ~7u) / 3));
  };

  clk.lap("v2 blocks");
  // --- lanes: a (point, frame) group of a regular point with more than T observations is cut into near-equal parts, one
  // lane each.  A wave walks max(lane size) observation steps, so this trades a few more lanes (and their block emission)
  // for fewer idle steps; every consumer (LDS accumulation, W staging, back-substitution) is linear in the per-lane
  // blocks, so two lanes with the same (point, frame) simply add up.  T is chosen per PASS by a small dynamic program
  // over a cost model of k_sweep2 (cycles measured with the in-kernel stamps at the 1 M-observation point).
  L.n_pairs = L.n_groups;
  std::vector<uint8_t> pass_break(L.P, 0);   // a pass of the plan below starts at this point
  {
    std::vector<uint32_t> split_of(L.P, 0);   // 0 = leave the point's groups whole
    auto parts_of = [](uint32_t n, uint32_t T) { return T ? (n + T - 1) / T : 1u; };
    // cycles per observation step of a pass, per pass, per lane (LIFCAL_PLAN_COST="step,pass,lane" overrides: tuning aid)
    // (64-lane passes, k_front4: a tile costs its observation steps plus the emission / gather of one wave; lanes are nearly free)
    double C_STEP = PL == 64 ? 1000.0 : 4700.0, C_PASS = (PL >= 256 ? 21500.0 : (PL == 128 ? 13000.0 : 3000.0)), C_LANE = PL == 64 ? 10.0 : 65.0;
    if (const char* e = getenv("LIFCAL_PLAN_COST")) { double a, b2, c2; if (sscanf(e, "%lf,%lf,%lf", &a, &b2, &c2) == 3) { C_STEP = a; C_PASS = b2; C_LANE = c2; } }
    // blocks are independent (each writes split_of / pass_break of its own points only)
    plan_parallel_for(blk_begin.empty() ? 0u : (uint32_t)(blk_begin.size() - 1), 4, [&](uint32_t b) {
      const uint32_t np_cap = block_np_cap(b);
      const size_t i0 = blk_begin[b], n = blk_begin[b + 1] - i0;
      uint32_t nmax = 0;
      for (size_t k = 0; k < n; ++k)
        for (uint32_t g = L.pt_slot0[reg[i0 + k]]; g < L.pt_slot0[reg[i0 + k]] + L.pt_nslots[reg[i0 + k]]; ++g) nmax = std::max(nmax, groups[g].n);
      // candidate split sizes: 0 (whole groups) and, in automatic mode, 2 .. min(largest group, 16); a fixed request otherwise
      std::vector<uint32_t> Ts(1, 0u);
      if (split_obs == UINT32_MAX) { for (uint32_t T = 2; T <= std::min(nmax, 16u); ++T) Ts.push_back(T); }
      else if (split_obs > 0) Ts.assign(1, split_obs);
      const size_t nT = Ts.size();
      std::vector<uint32_t> lanes(n * nT), steps(n * nT);
      for (size_t k = 0; k < n; ++k)
        for (size_t t = 0; t < nT; ++t) {
          uint32_t l = 0, st = 0;
          for (uint32_t g = L.pt_slot0[reg[i0 + k]]; g < L.pt_slot0[reg[i0 + k]] + L.pt_nslots[reg[i0 + k]]; ++g) {
            const uint32_t parts = parts_of(groups[g].n, Ts[t]);
            l += parts; st = std::max(st, (groups[g].n + parts - 1) / parts);
          }
          lanes[k * nT + t] = l; steps[k * nT + t] = st;
        }
      // dynamic program over the block's points: a pass = maximal run of points that fits 256 lanes / np_cap points at ONE
      // split size; cost[i] = cheapest way to cover points i.. (more lanes can mean one more pass, fewer steps per pass)
      std::vector<double> cost(n + 1, 0.0);
      std::vector<uint32_t> pick(n, 0), nxt(n, 0);
      for (size_t i = n; i-- > 0;) {
        double best = 1e300;
        for (size_t t = 0; t < nT; ++t) {
          uint32_t ng = 0, st = 0; size_t e = i;
          while (e < n && e - i < np_cap && ng + lanes[e * nT + t] <= PL) { ng += lanes[e * nT + t]; st = std::max(st, steps[e * nT + t]); ++e; }
          if (e == i) continue;                                                  // a single point too wide for this split size
          if (Ts[t] != 0 && lanes[i * nT + t] > (PL == 64 ? PL / 2 : PL / 4) && nT > 1) continue;   // keep several points per pass
          const double c = C_STEP * st + C_PASS + C_LANE * ng + cost[e];
          if (c < best) { best = c; pick[i] = (uint32_t)t; nxt[i] = (uint32_t)e; }
        }
        if (best == 1e300) { pick[i] = 0; nxt[i] = (uint32_t)i + 1; best = C_PASS + cost[i + 1]; }   // whole groups always fit (regular points have <= 256 groups)
        cost[i] = best;
      }
      for (size_t i = 0; i < n; i = nxt[i]) {
        for (size_t k = i; k < nxt[i]; ++k) split_of[reg[i0 + k]] = Ts[pick[i]];
        if (i > 0) pass_break[reg[i0 + i]] = 1;
      }
    });
    std::vector<Group> cut;
    cut.reserve(groups.size() * 2);
    for (const Group& G : groups) {
      const uint32_t parts = parts_of(G.n, split_of[G.pt]);
      uint32_t s0 = G.s0;
      for (uint32_t k = 0; k < parts; ++k) {
        const uint32_t n = G.n / parts + (k < G.n % parts ? 1u : 0u);
        cut.push_back({G.pt, G.fr, s0, n});
        s0 += n;
      }
    }
    groups.swap(cut);
    L.n_groups = (uint32_t)groups.size();
    L.gid_fr.resize(L.n_groups);
    L.pt_slot0.assign(L.P, 0); L.pt_nslots.assign(L.P, 0);
    for (uint32_t g = 0; g < L.n_groups; ++g) {
      L.gid_fr[g] = groups[g].fr;
      if (L.pt_nslots[groups[g].pt] == 0) L.pt_slot0[groups[g].pt] = g;
      L.pt_nslots[groups[g].pt]++;
    }
  }

  clk.lap("lanes/passes");
  // --- v1 tiles from the special points' groups ---
  std::vector<uint32_t> g1;
  for (uint32_t g = 0; g < L.n_groups; ++g) if (special[groups[g].pt]) g1.push_back(g);
  L.n_tiles = ((uint32_t)g1.size() + 63) / 64;
  L.n_slots = L.n_tiles * 64;
  L.slot_pt.assign(L.n_slots, 0); L.slot_fr.assign(L.n_slots, 0); L.slot_cnt.assign(L.n_slots, 0); L.slot_gid.assign(L.n_slots, 0);
  L.tile_row0.assign(L.n_tiles + 1, 0);
  for (uint32_t t = 0; t < L.n_tiles; ++t) {
    uint32_t kmax = 0;
    for (uint32_t l = 0; l < 64; ++l) { const uint32_t k = t * 64 + l; if (k < g1.size()) kmax = std::max(kmax, groups[g1[k]].n); }
    L.tile_row0[t + 1] = L.tile_row0[t] + kmax;
  }
  {
    const size_t rows = L.tile_row0[L.n_tiles];
    L.ell_u.assign(rows * 64, 0.0); L.ell_v.assign(rows * 64, 0.0);
    L.ell_lens.assign(rows * 64, 0); L.ell_src.assign(rows * 64, UINT32_MAX);
    for (uint32_t k = 0; k < g1.size(); ++k) {
      const Group& G = groups[g1[k]];
      const uint32_t t = k / 64, l = k % 64;
      L.slot_pt[k] = G.pt; L.slot_fr[k] = G.fr; L.slot_cnt[k] = G.n; L.slot_gid[k] = g1[k];
      for (uint32_t j = 0; j < G.n; ++j) {
        const size_t at = ((size_t)L.tile_row0[t] + j) * 64 + l;
        const uint32_t i = L.obs_order[G.s0 + j];
        L.ell_u[at] = p->u[i]; L.ell_v[at] = p->v[i]; L.ell_lens[at] = obs_lens[G.s0 + j]; L.ell_src[at] = i;
      }
    }
  }

  clk.lap("v1 tiles");
  // --- v2 passes inside each block ---
  L.v2_points.clear(); L.v2_ptinfo.clear(); L.blk_pass0.assign(1, 0);
  L.pass_pt0.clear(); L.pass_np.clear(); L.pass_gid0.clear(); L.pass_ng.clear();
  for (size_t b = 0; b + 1 < blk_begin.size(); ++b) {
    const uint32_t np_cap = block_np_cap(b);
    const size_t j = blk_begin[b + 1];
    size_t a = blk_begin[b];
    while (a < j) {
      uint32_t ng = 0, np = 0; size_t e = a;
      // a pass covers one contiguous run of gids (a special point's groups in between end the pass)
      while (e < j && np < np_cap && ng + L.pt_nslots[reg[e]] <= PL && !(e > a && pass_break[reg[e]]) &&
             (e == a || L.pt_slot0[reg[e]] == L.pt_slot0[reg[e - 1]] + L.pt_nslots[reg[e - 1]])) { ng += L.pt_nslots[reg[e]]; ++np; ++e; }
      L.pass_pt0.push_back((uint32_t)L.v2_points.size()); L.pass_np.push_back(np);
      L.pass_gid0.push_back(L.pt_slot0[reg[a]]); L.pass_ng.push_back(ng);
      { uint32_t g0 = 0; for (size_t k = a; k < e; ++k) { L.v2_points.push_back(reg[k]); L.v2_ptinfo.push_back(g0 | (L.pt_nslots[reg[k]] << 16)); g0 += L.pt_nslots[reg[k]]; } }
      a = e;
    }
    L.blk_pass0.push_back((uint32_t)L.pass_pt0.size());
  }
  clk.lap("v2: pass ranges");
  L.n_blocks = (uint32_t)L.blk_flo.size(); L.n_passes = (uint32_t)L.pass_pt0.size();
  L.v2_passpt.assign((size_t)L.n_passes * Plan::NP_MAX, 0);
  for (uint32_t ps = 0; ps < L.n_passes; ++ps)
    for (uint32_t k = 0; k < L.pass_np[ps]; ++k) L.v2_passpt[(size_t)ps * Plan::NP_MAX + k] = L.v2_points[L.pass_pt0[ps] + k];
  L.v2_slot.assign((size_t)L.n_passes * PL, 0);
  L.v2f_pt.assign((size_t)L.n_passes * PL, 0); L.v2f_fr.assign((size_t)L.n_passes * PL, 0); L.v2f_cnt.assign((size_t)L.n_passes * PL, 0);
  L.v2_tile_row0.assign((size_t)L.n_passes * PT + 1, 0);
  {
    // a pass's groups are contiguous gids (regular points between two specials may be split by a special
    // point's groups, so walk the pass's points explicitly)
    std::vector<std::vector<uint32_t>> pass_groups(L.n_passes);
    for (uint32_t b = 0; b < L.n_blocks; ++b)
      for (uint32_t ps = L.blk_pass0[b]; ps < L.blk_pass0[b + 1]; ++ps)
        for (uint32_t k = 0; k < L.pass_np[ps]; ++k) {
          const uint32_t q = L.v2_points[L.pass_pt0[ps] + k];
          for (uint32_t g = L.pt_slot0[q]; g < L.pt_slot0[q] + L.pt_nslots[q]; ++g) pass_groups[ps].push_back(g);
        }
  clk.lap("v2: pass groups");
    // lane slot of the k-th group of a pass.  Point order (k_sweep2): group k -> wave k % 4, lane k / 4, the lanes of a
    // point sit side by side and the frame accumulators are replicated.  Frame order (k_sweep3): the pass's lanes are sorted
    // by frame and cut into four tiles of 64, so the lanes of ONE frame sit side by side and their frame-level blocks are
    // summed with DPP before a single lane adds them to LDS.
    struct LaneOf { uint32_t gid, lp; };
    std::vector<std::vector<LaneOf>> lanes(L.n_passes);
    plan_parallel_for(L.n_passes, 16, [&](uint32_t ps) {
      uint32_t lp = 0, left = 0;
      lanes[ps].reserve(pass_groups[ps].size());
      for (uint32_t k = 0; k < pass_groups[ps].size(); ++k) {
        const Group& G = groups[pass_groups[ps][k]];
        if (k == 0) { lp = 0; left = L.pt_nslots[G.pt]; }
        else if (left == 0) { ++lp; left = L.pt_nslots[G.pt]; }
        --left;
        lanes[ps].push_back({pass_groups[ps][k], lp});
      }
      if (frame_order)
        std::stable_sort(lanes[ps].begin(), lanes[ps].end(), [&](const LaneOf& x, const LaneOf& y) { return groups[x.gid].fr < groups[y.gid].fr; });
    });
  clk.lap("v2: lanes+sort");
    auto slot_of = [&](uint32_t k, uint32_t& w, uint32_t& l) { if (frame_order) { w = k / 64; l = k % 64; } else { w = k % PT; l = k / PT; } };
    for (uint32_t ps = 0; ps < L.n_passes; ++ps) {
      uint32_t kmax[4] = {0, 0, 0, 0};
      for (uint32_t k = 0; k < lanes[ps].size(); ++k) { uint32_t w, l; slot_of(k, w, l); kmax[w] = std::max(kmax[w], groups[lanes[ps][k].gid].n); }
      for (uint32_t w = 0; w < PT; ++w) L.v2_tile_row0[(size_t)ps * PT + w + 1] = L.v2_tile_row0[(size_t)ps * PT + w] + kmax[w];
    }
    const size_t rows = L.v2_tile_row0[(size_t)L.n_passes * PT];
    L.v2_u = ZeroVec<double>(); L.v2_v = ZeroVec<double>(); L.v2_lens = ZeroVec<uint32_t>();   // fresh storage: zero pages, untouched
    // (+ 1 row of padding: k_sweep3 requests the first two rows of a tile unconditionally, also for an empty last tile)
    L.v2_u.resize((rows + 1) * 64); L.v2_v.resize((rows + 1) * 64); L.v2_lens.resize((rows + 1) * 64); L.v2_src.assign((rows + 1) * 64, UINT32_MAX);
    L.v2_du = ZeroVec<float>(); L.v2_dv = ZeroVec<float>();
    if (want_f32) { L.v2_du.resize((rows + 1) * 64); L.v2_dv.resize((rows + 1) * 64); }
    L.v2_gidx.assign((size_t)L.n_passes * PL, 0);
  clk.lap("v2: rows+alloc");
    std::vector<uint32_t> pass_block(L.n_passes, 0);
    for (uint32_t b = 0; b < L.n_blocks; ++b) for (uint32_t ps = L.blk_pass0[b]; ps < L.blk_pass0[b + 1]; ++ps) pass_block[ps] = b;
    plan_parallel_for(L.n_passes, 8, [&](uint32_t ps) {
      {
        const uint32_t b = pass_block[ps];
        // replica = occurrence rank of the lane's frame inside its wave (point order only: k_sweep2's replicated accumulators)
        std::vector<uint32_t> occ(4 * (Plan::NF_MAX + 1), 0);
        for (uint32_t k = 0; k < lanes[ps].size(); ++k) {
          const Group& G = groups[lanes[ps][k].gid];
          const uint32_t lp = lanes[ps][k].lp;
          uint32_t w, l; slot_of(k, w, l);
          const uint32_t lfl = G.fr - L.blk_flo[b];
          const uint32_t rep = std::min(255u, occ[w * (Plan::NF_MAX + 1) + lfl]++);
          const size_t at_slot = (size_t)ps * PL + w * 64 + l;
          L.v2_slot[at_slot] = std::min(G.n, 255u) | (lfl << 8) | (lp << 16) | (rep << 24);
          L.v2f_pt[at_slot] = G.pt; L.v2f_fr[at_slot] = G.fr; L.v2f_cnt[at_slot] = G.n;
          L.v2_gidx[at_slot] = lanes[ps][k].gid;
          for (uint32_t j = 0; j < G.n; ++j) {
            const size_t at = ((size_t)L.v2_tile_row0[(size_t)ps * PT + w] + j) * 64 + l;
            const uint32_t i = L.obs_order[G.s0 + j];
            L.v2_u[at] = p->u[i]; L.v2_v[at] = p->v[i]; L.v2_lens[at] = obs_lens[G.s0 + j]; L.v2_src[at] = i;
            if (want_f32) { L.v2_du[at] = (float)(p->u[i] - p->mcx[i]); L.v2_dv[at] = (float)(p->v[i] - p->mcy[i]); }
          }
        }
        // the pass's gids must be one contiguous run for the W staging (true unless a special point sits inside)
        L.pass_gid0[ps] = pass_groups[ps].empty() ? 0 : pass_groups[ps][0];
      }
    });
  }

  clk.lap("v2 passes");
  // --- constraints owned by this rank; CSR of partner columns per eliminated point ---
  L.my_constraints.clear();
  L.pt_cons0.assign(L.P + 1, 0); L.pt_cons_list.clear();
  if (L.use_constraints) {
    for (uint32_t c = 0; c < L.M; ++c) if (L.owner[L.c_i[c]] == rank || (L.owner[L.c_i[c]] < 0 && rank == 0)) L.my_constraints.push_back(c);
    for (uint32_t c : L.my_constraints) if (L.promoted[L.c_i[c]] < 0) L.pt_cons0[L.c_i[c] + 1]++;
    for (uint32_t q = 0; q < L.P; ++q) L.pt_cons0[q + 1] += L.pt_cons0[q];
    L.pt_cons_list.resize(L.pt_cons0[L.P]);
    std::vector<uint32_t> fill(L.pt_cons0.begin(), L.pt_cons0.end() - 1);
    for (uint32_t c : L.my_constraints) if (L.promoted[L.c_i[c]] < 0) L.pt_cons_list[fill[L.c_i[c]]++] = c;
    // a point that is only constrained (never observed) still needs an owner
    for (uint32_t c = 0; c < L.M; ++c) if (L.owner[L.c_i[c]] < 0) { L.owner[L.c_i[c]] = 0; if (rank == 0) L.owned_points.push_back(L.c_i[c]); }
  }
  clk.lap("constraints csr");
  if (clk.on) {
    uint64_t lanes = 0, steps = 0;
    for (uint32_t ps = 0; ps < L.n_passes; ++ps) { lanes += L.pass_ng[ps]; for (uint32_t w = 0; w < PT; ++w) steps += L.v2_tile_row0[(size_t)ps * PT + w + 1] - L.v2_tile_row0[(size_t)ps * PT + w]; }
    std::fprintf(stderr, "[plan] blocks %u passes %u (lanes/pass %u) max block frames %u, lanes %llu (%.1f per pass), tile steps %llu (%.2f per tile), obs %u\n", L.n_blocks, L.n_passes, PL,
                 L.max_block_nf, (unsigned long long)lanes, L.n_passes ? (double)lanes / L.n_passes : 0.0, (unsigned long long)steps, L.n_passes ? (double)steps / (L.n_passes * PT) : 0.0, L.n_obs_v2);
  }
  L.pt_special = special;
  L.special_owned.clear();
  for (uint32_t q : L.owned_points) if (special[q] || L.pt_nslots[q] == 0) { L.special_owned.push_back(q); L.pt_special[q] = 1; }
  if (getenv("LIFCAL_PLAN_HASH")) {   // fingerprint of the layout, to compare planner versions
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* q, size_t bytes) { const unsigned char* c = (const unsigned char*)q; for (size_t k = 0; k < bytes; ++k) { h ^= c[k]; h *= 1099511628211ull; } };
    auto mixv = [&](const auto& v) { if (!v.empty()) mix(v.data(), v.size() * sizeof(v[0])); };
    mixv(L.obs_order); mixv(L.lens_xy); mixv(L.ell_u); mixv(L.ell_v); mixv(L.ell_lens); mixv(L.ell_src); mixv(L.v2_u); mixv(L.v2_v); mixv(L.v2_lens); mixv(L.v2_src);
    mixv(L.v2_slot); mixv(L.v2f_pt); mixv(L.v2f_fr); mixv(L.v2f_cnt); mixv(L.v2_gidx); mixv(L.v2_tile_row0); mixv(L.v2_passpt); mixv(L.tile_row0); mixv(L.pt_slot0); mixv(L.pt_nslots);
    std::fprintf(stderr, "[plan] hash %016llx\n", (unsigned long long)h);
  }
  return 0;
}

// the ownership rule of build_plan applied to the index arrays of the whole problem (lifcal_ba_partition_points)
inline int partition_points(const lifcal_ba_problem* p, lifcal_ba_partition* out) {
  if (!p || !out || out->world_size < 1 || out->world_size > 64 || !out->point_owner || !out->rank_first || !out->rank_frames || !out->rank_obs || !out->frame_used) return LIFCAL_BA_ERR_INVALID_ARG;
  if (p->n_obs && (!p->pt || !p->fr)) return LIFCAL_BA_ERR_INVALID_ARG;
  const uint32_t P = p->n_points, F = p->n_frames, N = p->n_obs, W = out->world_size;
  for (uint32_t i = 0; i < N; ++i) if (p->pt[i] >= P || p->fr[i] >= F) return LIFCAL_BA_ERR_OUT_OF_RANGE;
  std::vector<uint32_t> first(P, UINT32_MAX), last(P, 0), cnt(P, 0);
  for (uint32_t f = 0; f < F; ++f) out->frame_used[f] = 0;
  for (uint32_t i = 0; i < N; ++i) { const uint32_t q = p->pt[i], f = p->fr[i]; first[q] = std::min(first[q], f); last[q] = std::max(last[q], f); cnt[q]++; out->frame_used[f] = 1; }
  uint32_t bw = 0;
  for (uint32_t q = 0; q < P; ++q) if (cnt[q]) bw = std::max(bw, last[q] - first[q]);
  std::vector<uint32_t> order(P);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
  uint64_t acc = 0;
  std::vector<uint32_t> lo(W, UINT32_MAX), hi(W, 0);
  for (uint32_t r = 0; r < W; ++r) out->rank_obs[r] = 0;
  for (uint32_t r = 0; r < P; ++r) {
    const uint32_t q = order[r];
    out->point_owner[q] = -1;
    if (!cnt[q]) continue;
    const int o = N ? (int)std::min<uint64_t>((uint64_t)W - 1, acc * (uint64_t)W / std::max<uint64_t>(N, 1)) : 0;
    out->point_owner[q] = o;
    acc += cnt[q];
    out->rank_obs[o] += cnt[q];
    lo[o] = std::min(lo[o], first[q]); hi[o] = std::max(hi[o], last[q]);
  }
  for (uint32_t r = 0; r < W; ++r) { out->rank_first[r] = lo[r] == UINT32_MAX ? 0 : lo[r]; out->rank_frames[r] = lo[r] == UINT32_MAX ? 0 : hi[r] - lo[r] + 1; }
  out->n_frames = F; out->n_points = P; out->band_width = bw; out->n_obs = N;
  return 0;
}

}  // namespace lifcal
