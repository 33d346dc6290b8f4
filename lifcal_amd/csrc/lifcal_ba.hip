// lifcal_ba.hip — C ABI (include/lifcal_ba.h) + host driver of the MI355X bundle-adjustment path.
//
// Host side of the seam reference src/CameraCalibration.cpp:858-965: lifcal_ba_create() replaces the
// residual-block construction loop, lifcal_ba_solve() replaces ceres::Solve() with a Levenberg–Marquardt
// trust-region loop whose arithmetic stays in HBM (only a handful of scalars per iteration cross PCIe
// for the accept/reject decision), lifcal_ba_reproj_stats() replaces calcReprojectionError() (:1026-1103).
// Trust-region logic follows Ceres 2.1 TrustRegionMinimizer / LevenbergMarquardtStrategy (out-of-tree,
// restated): radius 1e4, accept if rho > 1e-3, radius /= max(1/3, 1-(2 rho-1)^3), reject: radius /= 2^k,
// Jacobi scaling fixed at iteration 0, LM diagonal clamp [1e-6, 1e32].
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lifcal_ba.h"
#include "kernels.hpp"
#include "plan.hpp"
#include "linesearch.hpp"

using namespace lifcal;

static_assert(lifcal::DET_NF_MAX == lifcal::Plan::NF_MAX, "k_det_reduce searches the covering blocks of a frame inside Plan::NF_MAX frames");
static_assert(lifcal::Plan::PASS_GROUPS == 256, "k_det_reduce / k_det_reduce_all read the window layout of 256-lane passes (V2Lds(nf, true, 256))");

#ifndef LIFCAL_DEFAULT_SWEEP_WAVES
#define LIFCAL_DEFAULT_SWEEP_WAVES 4
#endif

namespace {

thread_local std::string g_last_error;

// Streams of destroyed handles are kept for the next create on the same device (a stream costs ~4 ms to make: the frame-windowed
// driver creates one handle per window).  A handle returns its stream idle (lifcal_ba_destroy synchronises first).
static std::mutex g_stream_pool_mutex;
static std::vector<hipStream_t> g_stream_pool[64];
static hipStream_t stream_pool_take(int device) {
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
  std::vector<hipStream_t>& v = g_stream_pool[device];
  if (v.empty()) return nullptr;
  hipStream_t s = v.back(); v.pop_back();
  return s;
}
static bool stream_pool_give(int device, hipStream_t s) {
  if (device < 0 || device >= 64) return false;
  std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
  std::vector<hipStream_t>& v = g_stream_pool[device];
  if (v.size() >= 8) return false;
  v.push_back(s);
  return true;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
      return LIFCAL_BA_ERR_HIP;                                                               \
    }                                                                                         \
  } while (0)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// RCCL entry points resolved at run time (the library has no link-time dependency on librccl)
struct NcclUniqueId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  bool load() {
    if (lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return false;
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    return GetUniqueId && CommInitRank && AllReduce && AllGather && CommDestroy;
  }
};
Rccl g_rccl;
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

}  // namespace

struct lifcal_ba_handle {
  Plan plan;
  lifcal_ba_options opt;
  lifcal_ba_problem prob;
  Dev d;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<void*> allocs;
  uint64_t bytes = 0;
  double* red_block = nullptr; size_t red_count = 0;
  size_t v2_lds_bytes = 0;
  bool use_sweep3 = true;        // wave-specialised LDS-window kernel (LIFCAL_SWEEP_KERNEL=2 selects k_sweep2)
  int sweep_waves = 4;           // k_sweep3: waves per role, 4 (512 threads, 256-lane passes) or 2 (256 threads, 128-lane passes, two workgroups per CU)
  bool use_sweep4 = false;       // k_front4 + k_back4 (sweep4.hpp): LIFCAL_SWEEP_KERNEL=4, fp64 and non-deterministic problems only
  size_t f4_lds = 0, b4_lds = 0; int b4_tpt = 1;
  TileSet ts1{}, ts2{};   // v1 tiles, v2 tiles (flat view)
  double* partial = nullptr;     // 4 doubles + 1 cand cost (all-reduced)
  double* hdiag_tmp = nullptr;
  double* stats_buf = nullptr;   // 4 sums + 2 max bit patterns
  double* stats_slots = nullptr; // multi-rank: 4 sums + one (max x, max y) pair per rank
  double* lens_xy = nullptr;
  double* pts_gather = nullptr;
  CamConsts* camc_stats = nullptr;
  double* h_scal = nullptr;      // pinned host mirror
  double* h_lm = nullptr;        // mapped host mirror of the device-resident LM state (d.lm), written by k_lm_control
  double* h_lm_dev = nullptr;    // ... its device address
  bool sigma_valid = false;
  bool constrained = false;
  lifcal_ba_allreduce_fn hook = nullptr; void* hook_ctx = nullptr;
  lifcal_ba_allgather_fn ghook = nullptr; void* ghook_ctx = nullptr;
  Xch xch{}; bool xch_ok = false, force_exchange = false;   // slab exchange of the reduced block (multi-GPU, no promoted points)
  void* comm = nullptr; bool comm_borrowed = false;   // RCCL communicator (borrowed: it belongs to another handle, see lifcal_ba_solve_windowed)
  double last_cost = 0, last_gmax = 0;
  size_t chol_lds = 0;
  double* ls_buf = nullptr;      // line search scalars, all-reduced: [0] |step|^2 [1] |x|^2 [2] grad . dir (each: this rank's points; rank 0 adds the replicated camera + pose part)
  double* dirmax_buf = nullptr;  // [0..63] per-rank max |point step| (one-hot slots, all-reduced), [64] max |reduced step| (replicated)
  uint8_t* frame_live_dev = nullptr;   // d.frame_live (frame is observed AND its pose is free), writable copy of the pointer
  bool trace = false;            // LIFCAL_TRACE=1: one stderr line per host decision of the LM loop, tagged with the rank
  double* Lpanel = nullptr; size_t bandw_lds = 0, backw_lds = 0; bool bandw_ok = false;
  CrPlan cr; bool use_cr = false;   // block odd-even reduction (bandchol3.hpp): long sequences
  bool twisted = false; uint32_t tw_m = 0; double *dumpA = nullptr, *dumpB = nullptr;   // two-ended factorisation (bandchol2.hpp): frames [0, tw_m) | bw middle frames | the rest
  // profiling: 5 events per sweep (start, after tables, after k_sweep, before k_schur, after k_schur, end)
  // A profile spans prof_cap sweeps (two records: before the first, behind the last); every prof_stride-th sweep of the span is
  // SAMPLED: its dominant kernel carries its own start / stop events — which costs ~5 us of queue time per sampled sweep, so a
  // timing loop samples a few of its sweeps rather than all of them (lifcal_ba_profile_begin_sampled)
  std::vector<hipEvent_t> prof_events; uint32_t prof_cap = 0, prof_used = 0, prof_seen = 0, prof_stride = 1; bool prof_on = false, prof_closed = false;
  hipEvent_t prof_span0 = nullptr, prof_span1 = nullptr;
  hipEvent_t prof_ev(int which) { return prof_events[(size_t)prof_used * 6 + which]; }
  bool prof_in_span() const { return prof_on && prof_seen < prof_cap; }
  bool prof_active() const { return prof_in_span() && prof_seen % prof_stride == 0; }
};

namespace {

template <class T>
int dev_alloc(lifcal_ba_handle* h, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIP_TRY(hipMalloc((void**)p, n * sizeof(T)));
  h->allocs.push_back(*p);
  h->bytes += n * sizeof(T);
  return 0;
}
template <class T, class A>
int dev_upload(lifcal_ba_handle* h, T** p, const std::vector<T, A>& v) {
  if (int rc = dev_alloc(h, p, v.size())) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

int do_allreduce(lifcal_ba_handle* h, double* buf, size_t count) {
  if (h->opt.world_size <= 1) return 0;
  if (h->hook) { if (h->hook(h->hook_ctx, buf, count, (void*)h->stream) != 0) { g_last_error = "all-reduce hook failed"; return LIFCAL_BA_ERR_COMM; } return 0; }
  if (h->comm) { if (g_rccl.AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, h->comm, h->stream) != 0) { g_last_error = "ncclAllReduce failed"; return LIFCAL_BA_ERR_COMM; } return 0; }
  g_last_error = "world_size > 1 but neither lifcal_ba_set_allreduce nor lifcal_ba_comm_init_rccl was called";
  return LIFCAL_BA_ERR_COMM;
}

// sum of the per-rank partial reduced blocks: slab all-gather + local add where it applies, else one sum all-reduce
int exchange_reduced(lifcal_ba_handle* h) {
  if (h->opt.world_size <= 1 && !h->force_exchange) return 0;   // (LIFCAL_FORCE_EXCHANGE: run pack / all-gather / unpack at world size 1, for tests)
  const bool have_gather = h->ghook ? (h->hook != nullptr) : (h->hook == nullptr && h->comm != nullptr);
  if (!h->xch_ok || !have_gather) return do_allreduce(h, h->red_block, h->red_count);
  const Xch& x = h->xch;
  hipLaunchKernelGGL(k_xch_pack, dim3((x.SL + 255) / 256), dim3(256), 0, h->stream, h->d, x);
  HIP_TRY(hipGetLastError());
  if (h->ghook) {
    if (h->ghook(h->ghook_ctx, x.send, (void*)x.recv, x.SL, (void*)h->stream) != 0) { g_last_error = "all-gather hook failed"; return LIFCAL_BA_ERR_COMM; }
  } else if (g_rccl.AllGather(x.send, (void*)x.recv, x.SL, kNcclFloat64, h->comm, h->stream) != 0) {
    // the partial block is still intact (packing only reads it): take the plain all-reduce from now on
    h->xch_ok = false;
    return do_allreduce(h, h->red_block, h->red_count);
  }
  const size_t n = (size_t)(x.F6 / 6) * x.BS + (size_t)x.NA * x.F6 + (size_t)3 * x.F6 + x.NA * x.NA + 3 * x.NA + SCAL_N;
  hipLaunchKernelGGL(k_xch_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d, x);
  HIP_TRY(hipGetLastError());
  return 0;
}

#ifdef LIFCAL_ONLY_FULL_CFG
// development builds (make dev): only the <2 radial, tangential, mlCenterAdj> instantiations, a fraction of the compile time
#define DISPATCH_CFG(h, CALL) do { if ((h)->plan.n_radial == 2 && (h)->plan.tangential && (h)->plan.adj) { CALL(2, true, true); } else { g_last_error = "development build: configuration not compiled in"; return LIFCAL_BA_ERR_INVALID_ARG; } } while (0)
#define DISPATCH_LENS(h, CALL) do { if ((h)->plan.n_radial == 2 && (h)->plan.tangential) { CALL(2, true); } else { g_last_error = "development build: configuration not compiled in"; return LIFCAL_BA_ERR_INVALID_ARG; } } while (0)
#else
#define DISPATCH_CFG(h, CALL)                                                       \
  do {                                                                              \
    const int nr_ = (h)->plan.n_radial; const bool tn_ = (h)->plan.tangential, aj_ = (h)->plan.adj; \
    if (nr_ == 0 && !tn_ && !aj_) { CALL(0, false, false); }                        \
    else if (nr_ == 0 && !tn_ && aj_) { CALL(0, false, true); }                     \
    else if (nr_ == 0 && tn_ && !aj_) { CALL(0, true, false); }                     \
    else if (nr_ == 0 && tn_ && aj_) { CALL(0, true, true); }                       \
    else if (nr_ == 1 && !tn_ && !aj_) { CALL(1, false, false); }                   \
    else if (nr_ == 1 && !tn_ && aj_) { CALL(1, false, true); }                     \
    else if (nr_ == 1 && tn_ && !aj_) { CALL(1, true, false); }                     \
    else if (nr_ == 1 && tn_ && aj_) { CALL(1, true, true); }                       \
    else if (nr_ == 2 && !tn_ && !aj_) { CALL(2, false, false); }                   \
    else if (nr_ == 2 && !tn_ && aj_) { CALL(2, false, true); }                     \
    else if (nr_ == 2 && tn_ && !aj_) { CALL(2, true, false); }                     \
    else { CALL(2, true, true); }                                                   \
  } while (0)

#define DISPATCH_LENS(h, CALL)                                                      \
  do {                                                                              \
    const int nr_ = (h)->plan.n_radial; const bool tn_ = (h)->plan.tangential;      \
    if (nr_ == 0 && !tn_) { CALL(0, false); } else if (nr_ == 0) { CALL(0, true); } \
    else if (nr_ == 1 && !tn_) { CALL(1, false); } else if (nr_ == 1) { CALL(1, true); } \
    else if (!tn_) { CALL(2, false); } else { CALL(2, true); }                      \
  } while (0)
#endif

uint32_t sweep_grid(const lifcal_ba_handle* h) {
  const uint32_t tiles = h->plan.n_tiles;
  const uint32_t wgs = (tiles + 3) / 4;
  if (h->d.deterministic) return std::max(1u, wgs);   // one tile per wave: the waves emit in tile order (det_turn_wait)
  return std::max(1u, std::min(wgs, 2048u));
}

// tables for a parameter set (camera constants, frames, lenses) + zero-fill of up to two buffers, ONE launch
int launch_tables(lifcal_ba_handle* h, const double* cam, const double* views, CamConsts* camc, double* ft, double* lt, bool tangents, bool fold,
                  double* zero0 = nullptr, size_t n_zero0 = 0, double* zero1 = nullptr, size_t n_zero1 = 0, float* ltf = nullptr) {
  const Dev& d = h->d;
  const uint32_t work = std::max<uint32_t>(std::max(d.n_lenses, d.F), (uint32_t)std::min<size_t>((n_zero0 + n_zero1 + 7) / 8, 1u << 20));
  const uint32_t grid = std::max(1u, (work + 255) / 256);
#define CALL_TABLES(NR, TAN) hipLaunchKernelGGL((k_tables<NR, TAN>), dim3(grid), dim3(256), 0, h->stream, d, cam, views, camc, ft, lt, (const double*)h->lens_xy, tangents ? 1 : 0, fold ? 1 : 0, zero0, (uint32_t)n_zero0, zero1, (uint32_t)n_zero1, ltf)
  DISPATCH_LENS(h, CALL_TABLES);
#undef CALL_TABLES
  HIP_TRY(hipGetLastError());
  return 0;
}

// the kernels that turn observations into blocks.  mode 1 = Hessian diagonal only (Jacobi scaling, iteration 0)
int launch_blocks(lifcal_ba_handle* h, double radius, int mode, bool zeroed) {
  Dev& d = h->d;
  if (!zeroed) HIP_TRY(hipMemsetAsync(h->red_block, 0, h->red_count * sizeof(double), h->stream));
  if (d.n_special) hipLaunchKernelGGL(k_zero_special, dim3((d.n_special * 36 + 255) / 256), dim3(256), 0, h->stream, d);
  if (d.deterministic && (d.n_tiles || d.n_special)) HIP_TRY(hipMemsetAsync(d.det_turn, 0, 4 * sizeof(uint32_t), h->stream));   // turn counters of k_sweep / k_schur
  // profiling: the dominant kernel carries its own start / stop events (hipExtLaunchKernelGGL: the time stamps are written by the
  // kernel's dispatch packet itself) — event records around it are separate barrier packets, ~2 us of idle queue each, four per sweep
  const bool prof = mode == 0 && h->prof_active();
  hipEvent_t ev_a = prof ? h->prof_ev(1) : nullptr, ev_b = prof ? h->prof_ev(2) : nullptr;
  if (d.n_blocks && h->use_sweep4) {   // regular points: observations -> blocks (k_front4), then the point elimination per block (k_back4)
#define CALL_FRONT4(NR, TAN, ADJ) hipExtLaunchKernelGGL((k_front4<NR, TAN, ADJ>), dim3(d.n_fwg), dim3(F4_THREADS), h->f4_lds, h->stream, ev_a, mode == 0 ? nullptr : ev_b, 0, d, mode)
    DISPATCH_CFG(h, CALL_FRONT4);
#undef CALL_FRONT4
    if (mode == 0) {
#define CALL_BACK4(NCV) do { if (h->b4_tpt == 1) hipExtLaunchKernelGGL((k_back4<NCV, 1>), dim3(d.n_blocks), dim3(B4_THREADS), h->b4_lds, h->stream, nullptr, ev_b, 0, d, radius); \
                             else hipExtLaunchKernelGGL((k_back4<NCV, 2>), dim3(d.n_blocks), dim3(B4_THREADS), h->b4_lds, h->stream, nullptr, ev_b, 0, d, radius); } while (0)
      switch (d.nc) { case 5: CALL_BACK4(5); break; case 6: CALL_BACK4(6); break; case 7: CALL_BACK4(7); break; case 8: CALL_BACK4(8); break; default: CALL_BACK4(9); break; }
#undef CALL_BACK4
    }
  } else if (d.n_blocks) {   // regular points: LDS-window kernel, one workgroup per block
    hipEvent_t ev_stop = d.deterministic ? nullptr : ev_b;   // (deterministic: the slab reduction below belongs to the dominant work)
#define LAUNCH_DOM(KERNEL, THREADS) hipExtLaunchKernelGGL(KERNEL, dim3(d.n_blocks), dim3(THREADS), h->v2_lds_bytes, h->stream, ev_a, ev_stop, 0, d, radius, mode)
#define CALL_SWEEP2(NR, TAN, ADJ) LAUNCH_DOM((k_sweep2<NR, TAN, ADJ>), 256)
#define CALL_SWEEP3(NR, TAN, ADJ) LAUNCH_DOM((k_sweep3<NR, TAN, ADJ, 4>), 512)
#define CALL_SWEEP3F(NR, TAN, ADJ) LAUNCH_DOM((k_sweep3<NR, TAN, ADJ, 4, float>), 512)
#define CALL_SWEEP3H(NR, TAN, ADJ) LAUNCH_DOM((k_sweep3<NR, TAN, ADJ, 2>), 256)
    if (h->opt.precision == 1) DISPATCH_CFG(h, CALL_SWEEP3F);
    else if (h->use_sweep3 && h->sweep_waves == 2) DISPATCH_CFG(h, CALL_SWEEP3H); else if (h->use_sweep3) DISPATCH_CFG(h, CALL_SWEEP3); else DISPATCH_CFG(h, CALL_SWEEP2);
#undef CALL_SWEEP3H
#undef CALL_SWEEP3F
    if (d.deterministic) {   // ordered sum of the per-block window slabs instead of the kernel's atomic flush
      const uint32_t NCd = d.nc, F6 = 6 * d.F;
      const uint64_t n = (uint64_t)d.F * (d.bw + 1) * 36 + (uint64_t)NCd * F6 + NCd * (NCd + 1) / 2 + 3ull * F6 + 3ull * NCd + 3;
      hipLaunchKernelGGL(k_det_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d, mode);
      hipExtLaunchKernelGGL(k_det_reduce_all, dim3(NCd * (NCd + 1) / 2 + 3 * NCd + 3), dim3(64), 0, h->stream, nullptr, ev_b, 0, d, mode);
    }
#undef CALL_SWEEP3
#undef CALL_SWEEP2
#undef LAUNCH_DOM
  } else if (prof) {   // no LDS-window blocks (special arities): bracket the fallback kernels with records
    HIP_TRY(hipEventRecord(ev_a, h->stream));
  }
  if (d.n_tiles) {    // special points (constraints, promoted, oversized, camera-only / pose-only arities): global atomics
#define CALL_SWEEP(NR, TAN, ADJ) hipLaunchKernelGGL((k_sweep<NR, TAN, ADJ>), dim3(sweep_grid(h)), dim3(256), 0, h->stream, d)
    DISPATCH_CFG(h, CALL_SWEEP);
#undef CALL_SWEEP
  }
  if (prof && !d.n_blocks) HIP_TRY(hipEventRecord(ev_b, h->stream));
  if (d.M_local) hipLaunchKernelGGL(k_constraints, dim3(d.deterministic ? 1 : (d.M_local + 63) / 64), dim3(d.deterministic ? 1 : 64), 0, h->stream, d, 0, (const double*)d.pts, d.scal + SCAL_COST);
  if (d.Q && d.use_points) hipLaunchKernelGGL(k_promote_diag, dim3((d.Q + 63) / 64), dim3(64), 0, h->stream, d);
  HIP_TRY(hipGetLastError());
  return 0;
}

// one Jacobian + Schur sweep at the current point and the given trust-region radius
int launch_sweep(lifcal_ba_handle* h, double radius) {
  Dev& d = h->d;
  if (h->prof_in_span() && h->prof_seen == 0) HIP_TRY(hipEventRecord(h->prof_span0, h->stream));   // the span opens with the first profiled sweep ...
  // the table kernel also zero-fills the reduced block and the step scalars
  if (int rc = launch_tables(h, d.cam, d.views, d.camc, d.ft, d.lt, true, true, h->red_block, h->red_count, d.step, ST_N, d.ltf)) return rc;
  bool zeroed = true;
  if (!h->sigma_valid) {
    // ceres fixes the Jacobi scaling at iteration 0 from the column norms of the (loss-corrected) Jacobian:
    // a diagonal-only pass, then 1 / (1 + sqrt(diag)) for every column
    if (int rc = launch_blocks(h, radius, 1, zeroed)) return rc;
    zeroed = false;
    const double* hd = d.hdiag;
    if (h->opt.world_size > 1) {
      HIP_TRY(hipMemcpyAsync(h->hdiag_tmp, d.hdiag, d.n_red * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      if (int rc = do_allreduce(h, h->hdiag_tmp, d.n_red)) return rc;
      hd = h->hdiag_tmp;
    }
    Dev dj = d; dj.hdiag = const_cast<double*>(hd);
    const uint32_t n = std::max(d.n_red, d.P);
    hipLaunchKernelGGL(k_jacobi, dim3((n + 255) / 256), dim3(256), 0, h->stream, dj, h->opt.jacobi_scaling ? 1 : 0);
    HIP_TRY(hipGetLastError());
    h->sigma_valid = true;
  }
  if (int rc = launch_blocks(h, radius, 0, zeroed)) return rc;
  if (d.use_points && d.n_special) hipLaunchKernelGGL(k_schur, dim3((d.n_special + 3) / 4), dim3(256), 0, h->stream, d, radius);
  HIP_TRY(hipGetLastError());
  {
    // multi-rank: the exchange of the partial reduced blocks, bracketed by two records when profiling (world > 1 only: a record is
    // a barrier packet, and a single-rank sweep has no exchange to time)
    const bool prof_x = h->prof_active() && (h->opt.world_size > 1 || h->force_exchange);
    if (prof_x) HIP_TRY(hipEventRecord(h->prof_ev(3), h->stream));
    if (int rc = exchange_reduced(h)) return rc;
    if (prof_x) HIP_TRY(hipEventRecord(h->prof_ev(4), h->stream));
  }
  hipLaunchKernelGGL(k_finalize, dim3((d.n_red + 255) / 256), dim3(256), 0, h->stream, d, radius);
  HIP_TRY(hipGetLastError());
  if (h->prof_in_span()) {   // ... and closes with the last one (a record per sweep is a barrier packet per sweep)
    if (h->prof_seen + 1 == h->prof_cap) { HIP_TRY(hipEventRecord(h->prof_span1, h->stream)); h->prof_closed = true; }
    if (h->prof_active()) h->prof_used++;
    h->prof_seen++;
  }
  return 0;
}

// read back cost and gradient max-norm of the last accumulate+reduce
int read_sweep_scalars(lifcal_ba_handle* h, double* cost, double* gmax, double* bad_u) {
  Dev& d = h->d;
  HIP_TRY(hipMemcpyAsync(h->h_scal, d.scal, (SCAL_N + ST_N) * sizeof(double), hipMemcpyDeviceToHost, h->stream));   // scal | step: contiguous
  HIP_TRY(hipStreamSynchronize(h->stream));
  *cost = h->h_scal[SCAL_COST];
  *bad_u = h->h_scal[SCAL_BAD_U];
  double g = 0.0;
  for (int r = 0; r < 64; ++r) { double v; std::memcpy(&v, &h->h_scal[SCAL_GMAX0 + r], 8); if (v > g) g = v; }
  double gr; std::memcpy(&gr, &h->h_scal[SCAL_N + ST_GMAX_RED], 8);
  *gmax = std::max(g, gr);
  return 0;
}

int launch_linear_solve(lifcal_ba_handle* h) {
  Dev& d = h->d;
  if (h->use_cr) {
    const CrSys sys{d.Sband, d.Sarrow, d.delta_red, d.step + ST_CHOL_FAIL, d.F, d.bw, d.NA, d.ld};
    cr_solve_launch(h->cr, sys, h->stream);
  } else if (h->bandw_ok && h->twisted) {
    // twisted factorisation: the chain from both ends on two workgroups, the bw middle frames + arrow last; back-substitution inside out
    const uint32_t m = h->tw_m, bw = d.bw, n2 = d.F - m - bw;
    const BandSeg sa{+1, 0u, m + bw, m, 0u, h->dumpA}, sb{-1, d.F - 1, n2 + bw, n2, 0u, h->dumpB}, sm{+1, m, bw, bw, 1u, nullptr};
    hipLaunchKernelGGL(k_band_chol_seg, dim3(2), dim3(256), h->bandw_lds, h->stream, d, h->Lpanel, sa, sb);
    const uint32_t nmerge = bw * (bw + 1) / 2 * 36 + (d.NA + 1) * bw * 6 + (d.NA + 1) * (d.NA + 1);
    hipLaunchKernelGGL(k_band_merge, dim3((nmerge + 255) / 256), dim3(256), 0, h->stream, d, m, bw, (const double*)h->dumpA, (const double*)h->dumpB);
    hipLaunchKernelGGL(k_band_chol_seg, dim3(1), dim3(256), h->bandw_lds, h->stream, d, h->Lpanel, sm, sm);
    hipLaunchKernelGGL(k_band_backsolve_seg, dim3(1), dim3(64), h->backw_lds, h->stream, d, (const double*)h->Lpanel, sm, sm);
    hipLaunchKernelGGL(k_band_backsolve_seg, dim3(2), dim3(64), h->backw_lds, h->stream, d, (const double*)h->Lpanel, sa, sb);
  } else if (h->bandw_ok) {   // window and solution vector fit LDS: one wave walks the chain
    hipLaunchKernelGGL(k_band_chol_w, dim3(1), dim3(256), h->bandw_lds, h->stream, d, h->Lpanel);
    hipLaunchKernelGGL(k_band_backsolve_w, dim3(1), dim3(64), h->backw_lds, h->stream, d, (const double*)h->Lpanel);
  } else {
    hipLaunchKernelGGL(k_band_chol, dim3(1), dim3(1024), h->chol_lds, h->stream, d);
    hipLaunchKernelGGL(k_band_backsolve, dim3(1), dim3(1024), 0, h->stream, d);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// candidate point x + delta, its cost and the scalars of the step-quality test
int launch_candidate(lifcal_ba_handle* h) {
  Dev& d = h->d;
  hipLaunchKernelGGL(k_update_reduced, dim3(1), dim3(256), 0, h->stream, d, h->partial);
  const uint32_t n = std::max(4 * d.n_owned, d.Q);   // four lanes per owned point
  if (d.use_points && n) {
    hipLaunchKernelGGL(k_backsub, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, h->partial);
    if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, (n + 255) / 256, 4u, h->partial);
  }
  HIP_TRY(hipGetLastError());
  if (int rc = launch_tables(h, d.cam_c, d.views_c, d.camc_c, d.ft_c, d.lt_c, false, true)) return rc;
  const double* pts_eval = d.use_points ? d.pts_c : d.pts;
  for (const TileSet* ts : {&h->ts1, &h->ts2}) {
    if (!ts->n_tiles) continue;
    const uint32_t grid = std::max(1u, std::min((ts->n_tiles + 3) / 4, 1024u));
#define CALL_COST(NR, TAN, ADJ) hipLaunchKernelGGL((k_cost<NR, TAN, ADJ>), dim3(grid), dim3(256), 0, h->stream, d, *ts, (const CamConsts*)d.camc_c, (const double*)d.ft_c, (const double*)d.lt_c, pts_eval, h->partial + 4)
    DISPATCH_CFG(h, CALL_COST);
#undef CALL_COST
    if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, grid, 1u, h->partial + 4);
  }
  if (d.M_local) hipLaunchKernelGGL(k_constraints, dim3(d.deterministic ? 1 : (d.M_local + 63) / 64), dim3(d.deterministic ? 1 : 64), 0, h->stream, d, 1, pts_eval, h->partial + 4);
  HIP_TRY(hipGetLastError());
  if (int rc = do_allreduce(h, h->partial, 8)) return rc;
  return 0;
}

// fp64 cost of the CURRENT point through the value-only kernel (options.precision = 1: the LM decisions compare costs of one
// arithmetic — current and candidate both from k_cost — while the sweep's own cost comes from its fp32 residuals)
int cost64_current(lifcal_ba_handle* h, double* cost) {
  Dev& d = h->d;
  HIP_TRY(hipMemsetAsync(h->partial, 0, 8 * sizeof(double), h->stream));
  const double* pts_eval = d.pts;
  for (const TileSet* ts : {&h->ts1, &h->ts2}) {
    if (!ts->n_tiles) continue;
    const uint32_t grid = std::max(1u, std::min((ts->n_tiles + 3) / 4, 1024u));
#define CALL_COST0(NR, TAN, ADJ) hipLaunchKernelGGL((k_cost<NR, TAN, ADJ>), dim3(grid), dim3(256), 0, h->stream, d, *ts, (const CamConsts*)d.camc, (const double*)d.ft, (const double*)d.lt, pts_eval, h->partial + 4)
    DISPATCH_CFG(h, CALL_COST0);
#undef CALL_COST0
    if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, grid, 1u, h->partial + 4);
  }
  if (d.M_local) hipLaunchKernelGGL(k_constraints, dim3(d.deterministic ? 1 : (d.M_local + 63) / 64), dim3(d.deterministic ? 1 : 64), 0, h->stream, d, 1, pts_eval, h->partial + 4);
  HIP_TRY(hipGetLastError());
  if (int rc = do_allreduce(h, h->partial, 8)) return rc;
  double hp[8];
  HIP_TRY(hipMemcpyAsync(hp, h->partial, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *cost = hp[4];
  return 0;
}

// ---- Armijo line search (bounded problems) ---------------------------------------------------------------------
// candidate = Plus(x, t delta); returns |x - candidate|^2 and |x|^2 in ls_buf
int launch_apply_step(lifcal_ba_handle* h, double t) {
  Dev& d = h->d;
  HIP_TRY(hipMemsetAsync(h->ls_buf, 0, 8 * sizeof(double), h->stream));
  const uint32_t n = std::max(std::max(6 * d.F, 17u), std::max(d.n_owned, d.Q));
  hipLaunchKernelGGL(k_apply_step, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, t, h->ls_buf);
  if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, (n + 255) / 256, 2u, h->ls_buf);
  HIP_TRY(hipGetLastError());
  return 0;
}

// phi(t) and phi'(t): one fused sweep at the trial point gives cost and gradient (the blocks of the current point
// are overwritten: the step, its model cost and the stored deltas have already been read)
int eval_trial(lifcal_ba_handle* h, double t, double radius, LsSample* smp) {
  Dev& d = h->d;
  if (int rc = launch_apply_step(h, t)) return rc;
  auto swap_all = [&]() { std::swap(d.cam, d.cam_c); std::swap(d.views, d.views_c); if (d.use_points) std::swap(d.pts, d.pts_c); };
  swap_all();
  int rc = launch_tables(h, d.cam, d.views, d.camc, d.ft, d.lt, true, true, h->red_block, h->red_count, d.step, ST_N, d.ltf);
  if (!rc) rc = launch_blocks(h, radius, 0, true);
  if (!rc && d.use_points && d.n_special) { hipLaunchKernelGGL(k_schur, dim3((d.n_special + 3) / 4), dim3(256), 0, h->stream, d, radius); }
  if (!rc) rc = do_allreduce(h, h->red_block, h->red_count);
  if (!rc) {
    const uint32_t n = std::max(d.n_red, d.n_owned);
    hipLaunchKernelGGL(k_dirderiv, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, h->ls_buf + 2);
    if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, (n + 255) / 256, 1u, h->ls_buf + 2);
    if (hipGetLastError() != hipSuccess) rc = LIFCAL_BA_ERR_HIP;
  }
  // options.precision = 1: the sweep's cost is that of the fp32 residuals, while phi(0) = x_cost and the candidate costs of the LM
  // loop come from the fp64 value kernel — near convergence the offset between the two arithmetics (~1e-7 relative) is far larger
  // than the Armijo margin, so the trial value is taken from the SAME fp64 kernel on the trial point's tables (ls_buf[3])
  const bool value64 = h->opt.precision == 1;
  if (!rc && value64) {
    const double* pts_eval = d.pts;   // (swapped: the trial point)
    for (const TileSet* ts : {&h->ts1, &h->ts2}) {
      if (!ts->n_tiles) continue;
      const uint32_t grid = std::max(1u, std::min((ts->n_tiles + 3) / 4, 1024u));
#define CALL_COSTT(NR, TAN, ADJ) hipLaunchKernelGGL((k_cost<NR, TAN, ADJ>), dim3(grid), dim3(256), 0, h->stream, d, *ts, (const CamConsts*)d.camc, (const double*)d.ft, (const double*)d.lt, pts_eval, h->ls_buf + 3)
      DISPATCH_CFG(h, CALL_COSTT);
#undef CALL_COSTT
      if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, grid, 1u, h->ls_buf + 3);
    }
    if (d.M_local) hipLaunchKernelGGL(k_constraints, dim3(d.deterministic ? 1 : (d.M_local + 63) / 64), dim3(d.deterministic ? 1 : 64), 0, h->stream, d, 1, pts_eval, h->ls_buf + 3);
    if (hipGetLastError() != hipSuccess) rc = LIFCAL_BA_ERR_HIP;
  }
  swap_all();
  if (rc) return rc;
  if (int rc2 = do_allreduce(h, h->ls_buf, value64 ? 4 : 3)) return rc2;
  double hb[8], cost;
  HIP_TRY(hipMemcpyAsync(hb, h->ls_buf, sizeof(hb), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(&cost, d.scal + SCAL_COST, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (value64) cost = hb[3];
  smp->x = t; smp->value = cost; smp->value_valid = std::isfinite(cost);
  smp->gradient = hb[2]; smp->gradient_valid = smp->value_valid && std::isfinite(smp->gradient);
  return 0;
}

struct StepScalars { double gtd, ddd, step2, x2, cand_cost, chol_fail; };

int read_step_scalars(lifcal_ba_handle* h, StepScalars* s) {
  HIP_TRY(hipMemcpyAsync(h->h_scal, h->d.step, (ST_N + 8) * sizeof(double), hipMemcpyDeviceToHost, h->stream));   // step | partial: contiguous
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double* a = h->h_scal; const double* p = h->h_scal + ST_N;
  // p[0..3]: all-reduced sums over every rank's points + the replicated camera / pose part contributed by rank 0 alone
  // (k_update_reduced): identical bits on every rank, as the branches they steer require
  s->gtd = p[0]; s->ddd = p[1]; s->step2 = p[2]; s->x2 = p[3];
  s->cand_cost = p[4]; s->chol_fail = a[ST_CHOL_FAIL];
  return 0;
}

void swap_current_candidate(lifcal_ba_handle* h) {
  Dev& d = h->d;
  std::swap(d.cam, d.cam_c); std::swap(d.views, d.views_c);
  if (d.use_points) std::swap(d.pts, d.pts_c);
}

int upload_parameters(lifcal_ba_handle* h) {
  Dev& d = h->d; const lifcal_ba_problem& p = h->prob;
  double cam[LIFCAL_BA_MAX_CAMERA_PARAMETERS];
  std::memcpy(cam, p.cam, sizeof(cam));
  if (h->constrained)  // ceres IterationZero: project the starting point onto the feasible set
    for (int k = 0; k < LIFCAL_BA_MAX_CAMERA_PARAMETERS; ++k) {
      if (p.lower && cam[k] < p.lower[k]) cam[k] = p.lower[k];
      if (p.upper && cam[k] > p.upper[k]) cam[k] = p.upper[k];
    }
  // (the candidate copies are filled on the device: one trip over PCIe per array)
  HIP_TRY(hipMemcpyAsync(d.cam, cam, sizeof(cam), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(d.cam_c, d.cam, sizeof(cam), hipMemcpyDeviceToDevice, h->stream));
  if (d.F) { HIP_TRY(hipMemcpyAsync(d.views, p.views, 6 * (size_t)d.F * 8, hipMemcpyHostToDevice, h->stream));
             HIP_TRY(hipMemcpyAsync(d.views_c, d.views, 6 * (size_t)d.F * 8, hipMemcpyDeviceToDevice, h->stream)); }
  if (d.P) { HIP_TRY(hipMemcpyAsync(d.pts, p.pts, 3 * (size_t)d.P * 8, hipMemcpyHostToDevice, h->stream));
             HIP_TRY(hipMemcpyAsync(d.pts_c, d.pts, 3 * (size_t)d.P * 8, hipMemcpyDeviceToDevice, h->stream)); }
  HIP_TRY(hipStreamSynchronize(h->stream));   // (cam is a stack array, the caller's arrays may change after the call)
  h->sigma_valid = false;
  return 0;
}

int download_parameters(lifcal_ba_handle* h) {
  Dev& d = h->d; const lifcal_ba_problem& p = h->prob;
  HIP_TRY(hipMemcpyAsync(p.cam, d.cam, LIFCAL_BA_MAX_CAMERA_PARAMETERS * 8, hipMemcpyDeviceToHost, h->stream));
  if (d.F) HIP_TRY(hipMemcpyAsync(p.views, d.views, 6 * (size_t)d.F * 8, hipMemcpyDeviceToHost, h->stream));
  if (d.P) {
    if (h->opt.world_size > 1 && d.use_points) {
      // every rank contributes the points it owns (promoted ones: rank 0), the sum is the full set
      std::vector<double> host(3 * (size_t)d.P), mine(3 * (size_t)d.P, 0.0);
      HIP_TRY(hipMemcpyAsync(host.data(), d.pts, host.size() * 8, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      for (uint32_t q = 0; q < d.P; ++q) {
        const bool take = (h->plan.promoted[q] >= 0) ? (h->opt.rank == 0) : (h->plan.owner[q] == h->opt.rank || (h->plan.owner[q] < 0 && h->opt.rank == 0));
        if (take) for (int k = 0; k < 3; ++k) mine[3 * (size_t)q + k] = host[3 * (size_t)q + k];
      }
      HIP_TRY(hipMemcpyAsync(h->pts_gather, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, h->stream));
      if (int rc = do_allreduce(h, h->pts_gather, mine.size())) return rc;
      HIP_TRY(hipMemcpyAsync(p.pts, h->pts_gather, mine.size() * 8, hipMemcpyDeviceToHost, h->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(p.pts, d.pts, 3 * (size_t)d.P * 8, hipMemcpyDeviceToHost, h->stream));
    }
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

}  // namespace

extern "C" {

void lifcal_ba_default_options(lifcal_ba_options* o) {
  if (!o) return;
  o->function_tolerance = 1e-6; o->parameter_tolerance = 1e-8; o->gradient_tolerance = 1e-10;
  o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->loss_scale = 0.5; o->max_iterations = 200; o->jacobi_scaling = 1; o->precision = 0; o->device = 0;
  o->rank = 0; o->world_size = 1; o->verbose = 0; o->deterministic = 0;
}

const char* lifcal_ba_strerror(int code) {
  switch (code) {
    case LIFCAL_BA_OK: return "ok";
    case LIFCAL_BA_ERR_INVALID_ARG: return "invalid argument";
    case LIFCAL_BA_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
    case LIFCAL_BA_ERR_HIP: return "HIP runtime error";
    case LIFCAL_BA_ERR_OUT_OF_RANGE: return "point/frame index out of range";
    case LIFCAL_BA_ERR_NOMEM: return "out of memory";
    case LIFCAL_BA_ERR_COMM: return "collective communication error";
    case LIFCAL_BA_ERR_NUMERIC: return "non-finite cost";
    default: return "unknown error";
  }
}
const char* lifcal_ba_last_error(void) { return g_last_error.c_str(); }
const char* lifcal_ba_version(void) { return "lifcal_amd 0.1 (gfx950)"; }

// min-norm least squares for the n x 2 system through the eigen-decomposition of its 2 x 2 Gram matrix: what
// Eigen::JacobiSVD::solve returns (singular values <= 2 eps sigma_max count as zero), reference CameraCalibration.cpp:491
static void solve_n_by_2(double a, double b, double c, double r0, double r1, double* x0, double* x1, int* rank) {
  // G = [a b; b c] = A^T A, r = A^T rhs
  const double tr = a + c, df = a - c, rad = std::sqrt(df * df + 4.0 * b * b);
  double l1 = 0.5 * (tr + rad), l2 = 0.5 * (tr - rad);
  if (l2 < 0.0) l2 = 0.0;
  double v1x, v1y;   // unit eigenvector of l1
  if (std::fabs(b) > 0.0) { v1x = l1 - c; v1y = b; } else if (a >= c) { v1x = 1.0; v1y = 0.0; } else { v1x = 0.0; v1y = 1.0; }
  const double nv = std::sqrt(v1x * v1x + v1y * v1y);
  if (nv > 0.0) { v1x /= nv; v1y /= nv; } else { v1x = 1.0; v1y = 0.0; }
  const double v2x = -v1y, v2y = v1x;
  const double smax = std::sqrt(l1), thr = 2.0 * 2.220446049250313e-16 * smax;
  *rank = 0; *x0 = 0.0; *x1 = 0.0;
  if (l1 > 0.0 && std::sqrt(l1) > thr) { const double t = (v1x * r0 + v1y * r1) / l1; *x0 += t * v1x; *x1 += t * v1y; ++*rank; }
  if (l2 > 0.0 && std::sqrt(l2) > thr) { const double t = (v2x * r0 + v2y * r1) / l2; *x0 += t * v2x; *x1 += t * v2y; ++*rank; }
}

int lifcal_init_plenoptic(const lifcal_init_problem* p, int32_t device, lifcal_init_result* out) {
  if (!p || !out || (p->n && (!p->vdepth || !p->fr || !p->pt)) || (p->n_frames && !p->world_to_cam) || (p->n_points && !p->pts)) {
    g_last_error = "lifcal_init_plenoptic: null argument"; return LIFCAL_BA_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { g_last_error = "no HIP device (this path has no CPU fallback)"; return LIFCAL_BA_ERR_NO_DEVICE; }
  HIP_TRY(hipSetDevice(device));
  double *dv = nullptr, *dw = nullptr, *dp = nullptr, *dout = nullptr; uint32_t *df = nullptr, *dq = nullptr;
  auto release = [&]() { for (void* q : {(void*)dv, (void*)dw, (void*)dp, (void*)dout, (void*)df, (void*)dq}) if (q) (void)hipFree(q); };
  auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 8); if (e != hipSuccess) return e;
    return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  hipError_t e = up((void**)&dv, p->vdepth, p->n * 8);
  if (e == hipSuccess) e = up((void**)&df, p->fr, p->n * 4);
  if (e == hipSuccess) e = up((void**)&dq, p->pt, p->n * 4);
  if (e == hipSuccess) e = up((void**)&dw, p->world_to_cam, (size_t)p->n_frames * 16 * 8);
  if (e == hipSuccess) e = up((void**)&dp, p->pts, (size_t)p->n_points * 3 * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&dout, 6 * 8);
  if (e == hipSuccess) e = hipMemset(dout, 0, 6 * 8);
  double h[6] = {0, 0, 0, 0, 0, 0};
  if (e == hipSuccess) {
    const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>((p->n + 255) / 256, 1), 4096);
    hipLaunchKernelGGL(k_init_sums, dim3(grid), dim3(256), 0, 0, p->n, dv, df, dq, p->n_frames, p->n_points, dw, dp, p->fL_init, dout);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  }
  release();
  if (e != hipSuccess) { g_last_error = std::string("lifcal_init_plenoptic: ") + hipGetErrorString(e); return LIFCAL_BA_ERR_HIP; }
  if (h[5] != 0.0) { g_last_error = "lifcal_init_plenoptic: frame or point index out of range"; return LIFCAL_BA_ERR_OUT_OF_RANGE; }
  int rank = 0;
  solve_n_by_2(h[0], h[1], h[2], h[3], h[4], &out->B_init, &out->bL0_init, &rank);
  out->n_used = (uint64_t)h[2]; out->rank = rank; out->reserved = 0;
  return 0;
}

// the sweep implementation for regular points (LIFCAL_SWEEP_KERNEL): 3 = k_sweep3 (default), 2 = k_sweep2, 4 = k_front4 + k_back4
// (sweep4.hpp: the two-kernel, pipelined-evaluator design of round 3 — parity green, measured slower than k_sweep3: DESIGN.md 4.5)
static int sweep_kernel_from_env() {
  const char* e = getenv("LIFCAL_SWEEP_KERNEL");
  const int k = e ? atoi(e) : 3;
  return (k == 2 || k == 4) ? k : 3;
}
// the planner's layout for a kernel choice: lanes per pass and number of blocks
static void sweep_layout(int kernel, int waves, uint32_t* pass_lanes, uint32_t* blocks) {
  if (kernel == 4) { *pass_lanes = 64u; *blocks = 256u; }
  else { *pass_lanes = waves == 2 ? 128u : 256u; *blocks = waves == 2 ? 512u : 256u; }
}

// k_sweep3's waves per role: LIFCAL_SWEEP_WAVES = 2 | 4 (k_sweep2 always works on 256-lane passes)
static int sweep_waves_from_env(bool sweep3) {
  if (!sweep3) return 4;
  const char* e = getenv("LIFCAL_SWEEP_WAVES");
  const int w = e ? atoi(e) : LIFCAL_DEFAULT_SWEEP_WAVES;
  return w == 2 ? 2 : 4;
}

int lifcal_init_plenoptic_recalibration(double fL_fixed, double B_fixed, lifcal_init_result* out) {
  if (!out) return LIFCAL_BA_ERR_INVALID_ARG;
  out->B_init = B_fixed; out->bL0_init = fL_fixed - 2 * B_fixed;   // reference :509
  out->n_used = 0; out->rank = 2; out->reserved = 0;
  return 0;
}

int lifcal_ba_plan(const lifcal_ba_problem* p, int32_t rank, int32_t world_size, lifcal_ba_plan_info* info,
                   uint32_t* obs_order, uint32_t* point_owner) {
  Plan pl;
  // the same layout lifcal_ba_create builds by default (lanes in frame order for the wave-specialised sweep kernel)
  const int kernel = sweep_kernel_from_env();
  const bool frame_order = kernel != 2;
  const int waves = sweep_waves_from_env(kernel == 3);
  uint32_t plan_lanes, plan_blocks; sweep_layout(kernel, waves, &plan_lanes, &plan_blocks);
  if (int rc = build_plan(p, rank, world_size, &pl, true, plan_blocks, UINT32_MAX, frame_order, plan_lanes)) return rc;
  if (info) {
    info->n_groups = pl.n_pairs; info->n_tiles = pl.n_tiles; info->n_lenses = pl.n_lenses; info->n_promoted = pl.Q;
    info->n_reduced = pl.n_red_canon; info->max_group_obs = pl.max_group_obs; info->n_chunks = pl.n_blocks; info->max_window_frames = pl.bw + 1;
    info->n_tiles = pl.n_tiles + pl.pass_tiles() * pl.n_passes;
  }
  if (obs_order) { for (uint32_t i = 0; i < p->n_obs; ++i) obs_order[i] = UINT32_MAX; for (size_t s = 0; s < pl.obs_order.size(); ++s) obs_order[s] = pl.obs_order[s]; }
  if (point_owner) for (uint32_t q = 0; q < p->n_points; ++q) point_owner[q] = (uint32_t)pl.owner[q];
  return 0;
}

static int create_impl(const lifcal_ba_problem* p, const lifcal_ba_options* o, lifcal_ba_handle** out, const lifcal_ba_partition* part);

int lifcal_ba_create(const lifcal_ba_problem* p, const lifcal_ba_options* o, lifcal_ba_handle** out) { return create_impl(p, o, out, nullptr); }

int lifcal_ba_create_shard(const lifcal_ba_problem* local, const lifcal_ba_partition* part, const lifcal_ba_options* o, lifcal_ba_handle** out) {
  if (!part) return LIFCAL_BA_ERR_INVALID_ARG;
  return create_impl(local, o, out, part);
}

int lifcal_ba_partition_points(const lifcal_ba_problem* index_only, lifcal_ba_partition* part) { return partition_points(index_only, part); }

int lifcal_ba_plan_shard(const lifcal_ba_problem* local, const lifcal_ba_partition* part, int32_t rank, lifcal_ba_plan_info* info) {
  if (!part || !info) return LIFCAL_BA_ERR_INVALID_ARG;
  Plan pl;
  const int kernel = sweep_kernel_from_env();
  const bool frame_order = kernel != 2;
  const int waves = sweep_waves_from_env(kernel == 3);
  uint32_t plan_lanes, plan_blocks; sweep_layout(kernel, waves, &plan_lanes, &plan_blocks);
  if (int rc = build_plan(local, rank, (int)part->world_size, &pl, true, plan_blocks, UINT32_MAX, frame_order, plan_lanes, false, part)) return rc;
  info->n_groups = pl.n_pairs; info->n_lenses = pl.n_lenses; info->n_promoted = pl.Q;
  info->n_reduced = pl.n_red_canon; info->max_group_obs = pl.max_group_obs; info->n_chunks = pl.n_blocks; info->max_window_frames = pl.bw + 1;
  info->n_tiles = pl.n_tiles + pl.pass_tiles() * pl.n_passes;
  if (getenv("LIFCAL_PLAN_HASH")) {}   // (build_plan prints the layout fingerprint itself)
  return 0;
}

static int create_impl(const lifcal_ba_problem* p, const lifcal_ba_options* o, lifcal_ba_handle** out, const lifcal_ba_partition* part) {
  if (!out) return LIFCAL_BA_ERR_INVALID_ARG;
  *out = nullptr;
  lifcal_ba_options opt; if (o) opt = *o; else lifcal_ba_default_options(&opt);
  if (opt.world_size < 1 || opt.world_size > 64 || opt.rank < 0 || opt.rank >= opt.world_size) return LIFCAL_BA_ERR_INVALID_ARG;
  if ((opt.precision != 0 && opt.precision != 1) || (opt.deterministic != 0 && opt.deterministic != 1)) {
    g_last_error = "options.precision and options.deterministic must be 0 or 1";
    return LIFCAL_BA_ERR_INVALID_ARG;
  }
  lifcal_ba_handle* h = new (std::nothrow) lifcal_ba_handle();
  if (!h) return LIFCAL_BA_ERR_NOMEM;
  h->opt = opt;
  // tuning / A-B knobs (not part of the ABI): LIFCAL_DISABLE_V2=1 forces the global-atomic kernels,
  // LIFCAL_V2_BLOCKS sets the number of workgroups the LDS-window sweep is cut into (default: one per CU)
  const bool enable_v2 = getenv("LIFCAL_DISABLE_V2") == nullptr;
  // the wave-specialised kernel wants the lanes of a pass sorted by frame, k_sweep2 (LIFCAL_SWEEP_KERNEL=2) by point
  int kernel = sweep_kernel_from_env();
  // ordered reductions and the fp32 evaluation exist in k_sweep3 only
  if (kernel == 4 && (opt.deterministic == 1 || opt.precision == 1)) kernel = 3;
  h->use_sweep3 = kernel != 2;   // (frame-ordered lanes: k_sweep3 and k_front4)
  h->use_sweep4 = kernel == 4;
  h->sweep_waves = sweep_waves_from_env(kernel == 3);
  if (opt.deterministic == 1) {
    if (!h->use_sweep3) { g_last_error = "options.deterministic = 1 needs k_sweep3 (unset LIFCAL_SWEEP_KERNEL)"; delete h; return LIFCAL_BA_ERR_INVALID_ARG; }
    h->sweep_waves = 4;
  }
  if (opt.precision == 1) {
    // fp32 residual / Jacobian evaluation exists in the wave-specialised kernel with four waves per role only
    if (!h->use_sweep3) { g_last_error = "options.precision = 1 needs k_sweep3 (unset LIFCAL_SWEEP_KERNEL)"; delete h; return LIFCAL_BA_ERR_INVALID_ARG; }
    h->sweep_waves = 4;
  }
  uint32_t plan_lanes, plan_blocks; sweep_layout(kernel, h->sweep_waves, &plan_lanes, &plan_blocks);
  const uint32_t v2_blocks = getenv("LIFCAL_V2_BLOCKS") ? (uint32_t)std::max(1, atoi(getenv("LIFCAL_V2_BLOCKS"))) : plan_blocks;
  // LIFCAL_GROUP_SPLIT: observations per lane above which a (point, frame) group is cut into several lanes
  // (0 = never; default: chosen per block by the planner's cost model)
  const uint32_t split_obs = getenv("LIFCAL_GROUP_SPLIT") ? (uint32_t)std::max(0, atoi(getenv("LIFCAL_GROUP_SPLIT"))) : UINT32_MAX;
  h->trace = getenv("LIFCAL_TRACE") != nullptr;
  PlanClock cclk;
  int rc = build_plan(p, opt.rank, opt.world_size, &h->plan, enable_v2, v2_blocks, split_obs, h->use_sweep3, plan_lanes, opt.precision == 1, part);
  if (rc) { delete h; return rc; }
  if (h->use_sweep4) {
    // k_back4 keeps at most two 4x4 tiles of the window per thread of a 256-thread group: wider frame windows take k_sweep3
    uint32_t max_ntri = 0;
    for (uint32_t b = 0; b < h->plan.n_blocks; ++b) max_ntri = std::max(max_ntri, V4Back::ntri_of(h->plan.blk_nf[b], (uint32_t)h->plan.nc));
    if (max_ntri > 512) {
      h->use_sweep4 = false; h->sweep_waves = 4;
      sweep_layout(3, 4, &plan_lanes, &plan_blocks);
      h->plan = Plan();
      rc = build_plan(p, opt.rank, opt.world_size, &h->plan, enable_v2, getenv("LIFCAL_V2_BLOCKS") ? v2_blocks : plan_blocks, split_obs, true, plan_lanes, false, part);
      if (rc) { delete h; return rc; }
    } else {
      h->b4_tpt = max_ntri > 256 ? 2 : 1;
    }
  }
  cclk.lap("create: plan");
  h->prob = *p;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || opt.device >= ndev) {
    g_last_error = "no HIP device available: the bundle-adjustment path has no CPU fallback"; delete h; return LIFCAL_BA_ERR_NO_DEVICE;
  }
  if (hipSetDevice(opt.device) != hipSuccess) { g_last_error = "hipSetDevice failed"; delete h; return LIFCAL_BA_ERR_NO_DEVICE; }
  int n_cus = 0;
  {
    // the architecture of a device ordinal does not change while the process lives: hipGetDeviceProperties costs ~3 ms per call
    static std::mutex arch_mutex;
    static std::string arch_name[64];
    static int arch_cus[64];
    std::lock_guard<std::mutex> lock(arch_mutex);
    std::string& name = arch_name[opt.device & 63];
    if (name.empty() || opt.device >= 64) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, opt.device) != hipSuccess) { delete h; return LIFCAL_BA_ERR_NO_DEVICE; }
      name = prop.gcnArchName; arch_cus[opt.device & 63] = prop.multiProcessorCount;
    }
    n_cus = arch_cus[opt.device & 63];
    if (std::strncmp(name.c_str(), "gfx950", 6) != 0) {
      g_last_error = std::string("device is ") + name + ", this library carries gfx950 code objects only"; delete h; return LIFCAL_BA_ERR_NO_DEVICE;
    }
  }
  auto fail = [&](int code) { lifcal_ba_destroy(h); return code; };
  if ((h->stream = stream_pool_take(opt.device)) == nullptr &&
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
  cclk.lap("create: device, stream");

  const Plan& L = h->plan;
  Dev& d = h->d;
  std::memset(&d, 0, sizeof(d));
  d.F = L.F; d.P = L.P; d.Q = L.Q; d.NA = L.NA; d.bw = std::min(L.bw, L.F ? L.F - 1 : 0); d.nc = (uint32_t)L.nc;
  d.n_tiles = L.n_tiles; d.n_slots = L.n_slots; d.n_lenses = L.n_lenses; d.n_red = L.n_red_int; d.ld = 6 * L.F + L.NA + 1;
  d.M_local = (uint32_t)L.my_constraints.size(); d.n_owned = (uint32_t)L.owned_points.size();
  d.n_radial = L.n_radial; d.tangential = L.tangential; d.adj = L.adj; d.robust = L.robust;
  d.use_poses = L.use_poses; d.use_points = L.use_points; d.rank = opt.rank; d.world = opt.world_size;
  d.fixed_mask = p->fixed_mask; d.spx = p->spx; d.spy = p->spy; d.scale = p->scale; d.loss_scale = opt.loss_scale;
  d.lm_min = opt.min_lm_diagonal; d.lm_max = opt.max_lm_diagonal;
  h->constrained = false;
  for (int k = 0; k < LIFCAL_BA_MAX_CAMERA_PARAMETERS; ++k) {
    if (p->lower && p->lower[k] > -std::numeric_limits<double>::max()) h->constrained = true;
    if (p->upper && p->upper[k] < std::numeric_limits<double>::max()) h->constrained = true;
  }
#define A(ptr, n) do { if (int rc_ = dev_alloc(h, &(ptr), (n))) return fail(rc_); } while (0)
#define U(ptr, vec) do { if (int rc_ = dev_upload(h, &(ptr), (vec))) return fail(rc_); } while (0)
  A(d.cam, 17); A(d.cam_c, 17); A(d.views, 6 * (size_t)d.F); A(d.views_c, 6 * (size_t)d.F); A(d.pts, 3 * (size_t)d.P); A(d.pts_c, 3 * (size_t)d.P);
  if (p->lower) { std::vector<double> v(p->lower, p->lower + 17); double* t; U(t, v); d.lower = t; }
  if (p->upper) { std::vector<double> v(p->upper, p->upper + 17); double* t; U(t, v); d.upper = t; }
  A(d.camc, 1); A(d.camc_c, 1); A(h->camc_stats, 1);
  A(d.ft, (size_t)d.F * FRAME_STRIDE); A(d.ft_c, (size_t)d.F * FRAME_STRIDE);
  A(d.lt, (size_t)d.n_lenses * LENS_STRIDE); A(d.lt_c, (size_t)d.n_lenses * LENS_STRIDE);
  U(h->lens_xy, L.lens_xy);
  { uint32_t* t; U(t, L.tile_row0); d.tile_row0 = t; U(t, L.slot_pt); d.slot_pt = t; U(t, L.slot_fr); d.slot_fr = t; U(t, L.slot_cnt); d.slot_cnt = t; U(t, L.ell_lens); d.ell_lens = t;
    U(t, L.gid_fr); d.gid_fr = t; U(t, L.slot_gid); d.slot_gid = t; U(t, L.special_owned); d.special_owned = t;
    U(t, L.blk_pass0); d.blk_pass0 = t; U(t, L.blk_flo); d.blk_flo = t; U(t, L.blk_nf); d.blk_nf = t;
    U(t, L.pass_pt0); d.pass_pt0 = t; U(t, L.pass_np); d.pass_np = t; U(t, L.pass_gid0); d.pass_gid0 = t; U(t, L.pass_ng); d.pass_ng = t;
    U(t, L.v2_points); d.v2_points = t; U(t, L.v2_ptinfo); d.v2_ptinfo = t; U(t, L.v2_passpt); d.v2_passpt = t; U(t, L.v2_gidx); d.v2_gidx = t; U(t, L.v2_slot); d.v2_slot = t; U(t, L.v2_tile_row0); d.v2_tile_row0 = t; U(t, L.v2_lens); d.v2_lens = t; }
  { double* t; U(t, L.v2_u); d.v2_u = t; U(t, L.v2_v); d.v2_v = t; }
  if (opt.precision == 1) { float* t; U(t, L.v2_du); d.v2_du = t; U(t, L.v2_dv); d.v2_dv = t; A(d.ltf, (size_t)d.n_lenses * LENS_STRIDE); A(d.ltw, 2 * (size_t)d.n_lenses); }
  { uint32_t *a, *b, *c; U(a, L.v2f_pt); U(b, L.v2f_fr); U(c, L.v2f_cnt);
    d.v2f_pt = a;
    h->ts2 = TileSet{L.pass_tiles() * L.n_passes, d.v2_tile_row0, a, b, c, d.v2_lens, d.v2_u, d.v2_v}; }
  d.n_blocks = L.n_blocks; d.v2_nfmax = std::max(1u, L.max_block_nf); d.n_special = (uint32_t)L.special_owned.size();
  // the LDS-window kernels trust the plan: the dense Z matrix of every pass (rows rounded up to 8) must fit its LDS region
  for (uint32_t b = 0; b < L.n_blocks; ++b) {
    const uint32_t ncolp = ((6 * L.blk_nf[b] + (uint32_t)L.nc + 1) + 15u) & ~15u;
    for (uint32_t ps = L.blk_pass0[b]; ps < L.blk_pass0[b + 1]; ++ps)
      if ((!h->use_sweep4 && (size_t)((3 * L.pass_np[ps] + 7u) & ~7u) * (ncolp + 2) > L.zd_doubles()) || L.pass_ng[ps] > L.pass_lanes || L.pass_np[ps] > L.np_max()) {
        g_last_error = "internal: a planned pass does not fit the LDS window";
        return fail(LIFCAL_BA_ERR_INVALID_ARG);
      }
  }
  h->v2_lds_bytes = (size_t)V2Lds(d.v2_nfmax, h->use_sweep3, L.pass_lanes).total * sizeof(double);
  if (h->use_sweep4) {
    h->f4_lds = (size_t)V4Front(d.v2_nfmax).total * sizeof(double);
    h->b4_lds = (size_t)V4Back(d.v2_nfmax).total * sizeof(double);
  }
  d.deterministic = opt.deterministic == 1 ? 1u : 0u;
  if (d.deterministic) {
    d.det_stride = (V2Lds(d.v2_nfmax, true, 256).off_fr + 3u + 1u) & ~1u;
    A(d.det_slab, (size_t)std::max(1u, d.n_blocks) * d.det_stride);
    A(d.det_slots, 4 * (size_t)std::max<uint32_t>(1024u, (std::max(4 * d.n_owned, d.Q) + 255u) / 256u));
    A(d.det_turn, 4);
  }
  if (d.n_blocks && h->use_sweep4) {
    // k_front4: as many workgroups as the chip holds at once (each walks a contiguous share of one block's tiles)
    int per_cu = 0;
#define SET_F4(NR, TAN, ADJ) do { if (hipFuncSetAttribute((const void*)k_front4<NR, TAN, ADJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->f4_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_front4<NR, TAN, ADJ>, (int)F4_THREADS, h->f4_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); } while (0)
    DISPATCH_CFG(h, SET_F4);
#undef SET_F4
#define SET_B4(NCV) do { if (hipFuncSetAttribute((const void*)k_back4<NCV, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->b4_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); \
    if (hipFuncSetAttribute((const void*)k_back4<NCV, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->b4_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); } while (0)
    switch (d.nc) { case 5: SET_B4(5); break; case 6: SET_B4(6); break; case 7: SET_B4(7); break; case 8: SET_B4(8); break; default: SET_B4(9); break; }
#undef SET_B4
    per_cu = std::max(1, std::min(per_cu, 5));
    if (const char* e = getenv("LIFCAL_F4_PER_CU")) per_cu = std::max(1, atoi(e));
    const uint32_t capacity = (uint32_t)per_cu * (uint32_t)std::max(1, n_cus);
    const uint32_t parts = std::max(1u, capacity / std::max(1u, d.n_blocks));
    std::vector<uint32_t> fwg_blk, fwg_pass0, blk_pt0(L.n_blocks + 1, 0);
    for (uint32_t b = 0; b < L.n_blocks; ++b) {
      const uint32_t p0 = L.blk_pass0[b], p1 = L.blk_pass0[b + 1];
      blk_pt0[b] = p0 < L.n_passes ? L.pass_pt0[p0] : (uint32_t)L.v2_points.size();
      // a tile costs its observation steps plus a fixed share (emission, gather): cut the block's tiles into `parts` equal runs
      auto tile_cost = [&](uint32_t ps) { return (uint64_t)(L.v2_tile_row0[ps + 1] - L.v2_tile_row0[ps]) + 3u; };
      uint64_t total = 0;
      for (uint32_t ps = p0; ps < p1; ++ps) total += tile_cost(ps);
      const uint32_t np = std::min(parts, std::max(1u, p1 - p0));
      uint64_t acc = 0; uint32_t next = 0;
      for (uint32_t ps = p0; ps < p1; ++ps) {
        if (next < np && acc * np >= (uint64_t)next * total) { fwg_blk.push_back(b); fwg_pass0.push_back(ps); ++next; }
        acc += tile_cost(ps);
      }
    }
    blk_pt0[L.n_blocks] = (uint32_t)L.v2_points.size();
    fwg_pass0.push_back(L.n_passes);
    d.n_fwg = (uint32_t)fwg_blk.size();
    if (cclk.on) std::fprintf(stderr, "[create] k_front4: %d workgroups per CU (occupancy API), %u workgroups over %u blocks, LDS %zu B; k_back4: LDS %zu B, %d tile(s) per thread\n",
                              per_cu, d.n_fwg, d.n_blocks, h->f4_lds, h->b4_lds, h->b4_tpt);
    { uint32_t* t; U(t, fwg_blk); d.fwg_blk = t; U(t, fwg_pass0); d.fwg_pass0 = t; U(t, blk_pt0); d.blk_pt0 = t; }
  } else if (d.n_blocks) {
#define SET_LDS(NR, TAN, ADJ) do { if (hipFuncSetAttribute((const void*)k_sweep2<NR, TAN, ADJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2_lds_bytes) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); \
    if (hipFuncSetAttribute((const void*)k_sweep3<NR, TAN, ADJ, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2_lds_bytes) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); \
    if (hipFuncSetAttribute((const void*)k_sweep3<NR, TAN, ADJ, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2_lds_bytes) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); \
    if (hipFuncSetAttribute((const void*)k_sweep3<NR, TAN, ADJ, 4, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2_lds_bytes) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); } while (0)
    DISPATCH_CFG(h, SET_LDS);
#undef SET_LDS
  }
  { double* t; U(t, L.ell_u); d.ell_u = t; U(t, L.ell_v); d.ell_v = t; }
  h->ts1 = TileSet{d.n_tiles, d.tile_row0, d.slot_pt, d.slot_fr, d.slot_cnt, d.ell_lens, d.ell_u, d.ell_v};
  { int32_t* t; U(t, L.promoted); d.promoted = t; }
  { uint32_t* t; U(t, L.promoted_ids); d.promoted_ids = t; U(t, L.pt_slot0); d.pt_slot0 = t; U(t, L.pt_nslots); d.pt_nslots = t; U(t, L.owned_points); d.owned = t; }
  { std::vector<uint8_t> live(L.frame_used); uint8_t* t; U(t, live); d.frame_live = t; h->frame_live_dev = t; }
  A(d.ptacc, (size_t)d.P * 36); A(d.Uinv, (size_t)d.P * 9);
  if (hipMemset(d.ptacc, 0, (size_t)std::max(1u, d.P) * 36 * sizeof(double)) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP); A(d.lamP, (size_t)d.P * 3); A(d.sigP, (size_t)d.P * 3);
  A(d.Wv, (size_t)L.n_groups * 18);
  A(d.Av, (size_t)L.n_groups * 6);
  { uint8_t* t; U(t, L.pt_special); d.pt_special = t; }
  if (L.use_constraints) {
    uint32_t* t; U(t, L.c_i); d.c_i = t; U(t, L.c_j); d.c_j = t; U(t, L.my_constraints); d.my_cons = t;
    U(t, L.pt_cons0); d.pt_cons0 = t; U(t, L.pt_cons_list); d.pt_cons_list = t;
    double* f; U(f, L.c_dist); d.c_dist = f; U(f, L.c_sigma); d.c_sigma = f;
    A(d.Wpart, (size_t)L.M * 9);
  }
  const size_t n_band = (size_t)d.F * (d.bw + 1) * 36, n_arrow = (size_t)(d.NA + 1) * d.ld;
  h->red_count = n_band + n_arrow + 3 * (size_t)d.n_red + SCAL_N;
  A(h->red_block, h->red_count + ST_N + 8);   // ... | scal | step scalars | candidate partial sums: the host reads each pair with ONE copy
  d.Sband = h->red_block; d.Sarrow = d.Sband + n_band; d.rhsacc = d.Sarrow + n_arrow; d.gB = d.rhsacc + d.n_red; d.hdiag = d.gB + d.n_red; d.scal = d.hdiag + d.n_red;
  A(d.sig_red, d.n_red); A(d.lam_red, d.n_red); A(d.delta_red, d.n_red); A(d.Linv, (size_t)d.F * 36 + 36);
  // multi-GPU: slab exchange of the reduced block (every rank's partial block lives in one frame range)
  h->force_exchange = getenv("LIFCAL_FORCE_EXCHANGE") != nullptr;
  if ((opt.world_size > 1 || h->force_exchange) && d.Q == 0 && d.use_poses && getenv("LIFCAL_DENSE_ALLREDUCE") == nullptr) {
    Xch& x = h->xch;
    x.world = (uint32_t)opt.world_size; x.rank = (uint32_t)opt.rank; x.BS = (d.bw + 1) * 36; x.NA = d.NA; x.F6 = 6 * d.F;
    uint32_t maxn = 1;
    for (uint32_t n : L.rk_nfr) maxn = std::max(maxn, n);
    x.maxn = maxn;
    const uint64_t off_arrow = (uint64_t)maxn * x.BS, off_vec = off_arrow + (uint64_t)maxn * x.NA * 6, off_tail = off_vec + (uint64_t)maxn * 18;
    const uint64_t SL = off_tail + (uint64_t)x.NA * x.NA + 3 * x.NA + SCAL_N;
    // worth it only if the gathered slabs are smaller than what a ring all-reduce moves (2x the block)
    if (SL < (1ull << 31) && (uint64_t)x.world * SL <= 2 * (uint64_t)h->red_count) {
      x.off_arrow = (uint32_t)off_arrow; x.off_vec = (uint32_t)off_vec; x.off_tail = (uint32_t)off_tail; x.SL = (uint32_t)SL;
      uint32_t* t; U(t, L.rk_flo); x.flo = t; U(t, L.rk_nfr); x.nfr = t;
      double* b; A(b, (size_t)SL); x.send = b; A(b, (size_t)x.world * SL); x.recv = b;
      h->xch_ok = true;
    }
  }
  A(d.dbg, std::max((size_t)std::max(1u, d.n_blocks) * 32, (size_t)d.n_fwg * 64));
  A(d.dP, 3 * (size_t)d.P); A(h->ls_buf, 8); A(h->dirmax_buf, 65); A(d.lm, LM_N);
  d.step = d.scal + SCAL_N; h->partial = d.step + ST_N;   // (behind the all-reduced block, not part of it)
  A(h->hdiag_tmp, d.n_red); A(h->stats_buf, 8); A(h->stats_slots, 4 + 2 * 64); A(h->pts_gather, 3 * (size_t)d.P);
  // Cholesky panel: LDS when it fits (<= 64 KiB by default launch limits), else a global scratch
  const size_t panel_rows = 6 * (size_t)d.bw + d.NA + 1;
  const size_t lds_need = (80 + panel_rows * 6) * sizeof(double);
  if (lds_need <= 64 * 1024) { h->chol_lds = lds_need; d.panel_g = nullptr; }
  else { h->chol_lds = 80 * sizeof(double); A(d.panel_g, panel_rows * 6); }
  {
    const BandLds bl(d.bw, d.NA);
    h->bandw_lds = (size_t)bl.total * sizeof(double);
    h->backw_lds = (size_t)std::max(1u, d.n_red) * sizeof(double);
    h->bandw_ok = h->bandw_lds <= 160 * 1024 && h->backw_lds <= 160 * 1024 && getenv("LIFCAL_DISABLE_BANDW") == nullptr;
    if (h->bandw_ok) {
      A(h->Lpanel, (size_t)std::max(1u, d.F) * (6 * (size_t)d.bw + d.NA + 1) * 6);
      if (hipFuncSetAttribute((const void*)k_band_chol_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bandw_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
      if (hipFuncSetAttribute((const void*)k_band_backsolve_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->backw_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
      // long enough for two chains to pay: eliminate from both ends (LIFCAL_TWISTED=0: the single chain)
      h->twisted = d.use_poses && d.bw >= 1 && d.F >= 3 * d.bw + 8 && !(getenv("LIFCAL_TWISTED") && atoi(getenv("LIFCAL_TWISTED")) == 0);
      if (h->twisted) {
        h->tw_m = (d.F - d.bw) / 2;
        const size_t nd = (size_t)d.bw * (d.bw + 1) / 2 * 36 + (size_t)(d.NA + 1) * d.bw * 6 + (size_t)(d.NA + 1) * (d.NA + 1);
        A(h->dumpA, nd); A(h->dumpB, nd);
        if (hipFuncSetAttribute((const void*)k_band_chol_seg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bandw_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
        if (hipFuncSetAttribute((const void*)k_band_backsolve_seg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->backw_lds) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
      }
    }
  }
  {
    // block odd-even reduction of the band + arrow system: log2(F / bw) levels on up to F / (2 bw) workgroups instead of a chain of
    // F / 2 steps on two; pays from ~32 super-blocks on (LIFCAL_CR=1 / 0 forces / forbids it where it is applicable)
    const char* ev = getenv("LIFCAL_CR");
    const int want = ev ? atoi(ev) : -1;
    if (d.use_poses && want != 0 && cr_eligible(d.F, d.bw, d.NA) && (want == 1 || cr_geometry(d.F, d.bw, d.NA).nb >= 32) && cr_plan(h->cr, d.F, d.bw, d.NA)) {
      A(h->cr.ws.P, cr_ws_doubles_P(h->cr.ws)); A(h->cr.ws.D, cr_ws_doubles_D(h->cr.ws)); A(h->cr.ws.A, cr_ws_doubles_A(h->cr.ws));
      A(h->cr.ws.U, cr_ws_doubles_U(h->cr.ws)); A(h->cr.ws.x, cr_ws_doubles_x(h->cr.ws));
      h->use_cr = true;
    }
  }
#undef A
#undef U
  cclk.lap("create: alloc + upload");
  if (hipHostMalloc((void**)&h->h_scal, (SCAL_N + 2 * ST_N + 16) * sizeof(double)) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
  // mapped + coherent: k_lm_control writes the state straight into it, the host polls the round number
  if (hipHostMalloc((void**)&h->h_lm, 2 * LM_N * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);   // (second half: staging of the initial state)
  if (hipHostGetDevicePointer((void**)&h->h_lm_dev, h->h_lm, 0) != hipSuccess) return fail(LIFCAL_BA_ERR_HIP);
  if (int rc2 = upload_parameters(h)) return fail(rc2);
  cclk.lap("create: parameters");
  *out = h;
  return 0;
}

void lifcal_ba_destroy(lifcal_ba_handle* h) {
  if (!h) return;
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm && !h->comm_borrowed && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->h_scal) (void)hipHostFree(h->h_scal);
  if (h->h_lm) (void)hipHostFree(h->h_lm);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
  if (h->prof_span0) { (void)hipEventDestroy(h->prof_span0); (void)hipEventDestroy(h->prof_span1); }
  if (h->stream && !stream_pool_give(h->opt.device, h->stream)) (void)hipStreamDestroy(h->stream);
  delete h;
}

int lifcal_ba_set_fixed_frames(lifcal_ba_handle* h, const uint8_t* fixed) {
  if (!h) return LIFCAL_BA_ERR_INVALID_ARG;
  if (!h->use_sweep3) { g_last_error = "lifcal_ba_set_fixed_frames needs k_sweep3 (unset LIFCAL_SWEEP_KERNEL)"; return LIFCAL_BA_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(h->opt.device));
  std::vector<uint8_t> live(h->plan.frame_used);
  if (fixed) for (uint32_t f = 0; f < h->d.F; ++f) if (fixed[f]) live[f] = 0;
  // a solve boundary: sweeps still queued on the (non-blocking) stream read the old mask to the end before it changes
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (!live.empty()) HIP_TRY(hipMemcpy(h->frame_live_dev, live.data(), live.size(), hipMemcpyHostToDevice));
  h->sigma_valid = false;   // the Jacobi scaling is fixed at the first sweep of a solve: a new column set starts a new solve
  return 0;
}

int lifcal_ba_upload_parameters(lifcal_ba_handle* h) { return h ? upload_parameters(h) : LIFCAL_BA_ERR_INVALID_ARG; }
int lifcal_ba_download_parameters(lifcal_ba_handle* h) { return h ? download_parameters(h) : LIFCAL_BA_ERR_INVALID_ARG; }

int lifcal_ba_set_allreduce(lifcal_ba_handle* h, lifcal_ba_allreduce_fn fn, void* ctx) {
  if (!h) return LIFCAL_BA_ERR_INVALID_ARG;
  h->hook = fn; h->hook_ctx = ctx;
  return 0;
}

int lifcal_ba_set_allgather(lifcal_ba_handle* h, lifcal_ba_allgather_fn fn, void* ctx) {
  if (!h) return LIFCAL_BA_ERR_INVALID_ARG;
  h->ghook = fn; h->ghook_ctx = ctx;
  return 0;
}

int lifcal_ba_comm_unique_id(void* out128) {
  if (!out128) return LIFCAL_BA_ERR_INVALID_ARG;
  if (!g_rccl.load()) { g_last_error = "could not load librccl"; return LIFCAL_BA_ERR_COMM; }
  NcclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != 0) { g_last_error = "ncclGetUniqueId failed"; return LIFCAL_BA_ERR_COMM; }
  std::memcpy(out128, &id, sizeof(id));
  return 0;
}

int lifcal_ba_comm_init_rccl(lifcal_ba_handle* h, const void* unique_id128) {
  if (!h || !unique_id128) return LIFCAL_BA_ERR_INVALID_ARG;
  if (!g_rccl.load()) { g_last_error = "could not load librccl"; return LIFCAL_BA_ERR_COMM; }
  NcclUniqueId id; std::memcpy(&id, unique_id128, sizeof(id));
  if (hipSetDevice(h->opt.device) != hipSuccess) return LIFCAL_BA_ERR_HIP;
  if (g_rccl.CommInitRank(&h->comm, h->opt.world_size, id, h->opt.rank) != 0) { g_last_error = "ncclCommInitRank failed"; h->comm = nullptr; return LIFCAL_BA_ERR_COMM; }
  return 0;
}

#ifdef LIFCAL_STAMPS
// diagnostic build only: copies the per-block phase cycle counters of the last k_sweep2 launch (32 per block: 16 of thread 0, 16 of thread 256)
extern "C" int lifcal_ba_debug_stamps(lifcal_ba_handle* h, unsigned long long* out, uint32_t max_blocks) {
  if (!h || !out) return LIFCAL_BA_ERR_INVALID_ARG;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->use_sweep4) {   // k_front4: 64 counters per front workgroup (16 per role)
    const uint32_t n = std::min(max_blocks / 2, h->d.n_fwg);
    HIP_TRY(hipMemcpy(out, h->d.dbg, (size_t)n * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return (int)n;
  }
  const uint32_t n = std::min(max_blocks, h->d.n_blocks);
  HIP_TRY(hipMemcpy(out, h->d.dbg, (size_t)n * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return (int)n;
}
#endif

int lifcal_ba_get_info(lifcal_ba_handle* h, lifcal_ba_info* out) {
  if (!h || !out) return LIFCAL_BA_ERR_INVALID_ARG;
  out->n_obs_local = h->plan.n_obs_local; out->n_points_local = (uint32_t)h->plan.owned_points.size();
  out->n_groups = h->plan.n_pairs; out->n_tiles = h->plan.n_tiles; out->n_lenses = h->plan.n_lenses;
  out->n_reduced = h->plan.n_red_canon; out->n_promoted = h->plan.Q; out->n_chunks = h->plan.n_blocks; out->max_window_frames = h->d.bw + 1;
  out->n_tiles = h->plan.n_tiles + h->plan.pass_tiles() * h->plan.n_passes;
  out->device_bytes = h->bytes; out->stream = (void*)h->stream;
  return 0;
}

int lifcal_ba_sweep_enqueue(lifcal_ba_handle* h, double radius) {
  if (!h || !(radius > 0.0)) return LIFCAL_BA_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->opt.device));
  return launch_sweep(h, radius);
}

int lifcal_ba_profile_begin_sampled(lifcal_ba_handle* h, uint32_t max_sweeps, uint32_t stride) {
  if (!h || stride == 0) return LIFCAL_BA_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->opt.device));
  const size_t samples = ((size_t)max_sweeps + stride - 1) / stride;
  while (h->prof_events.size() < samples * 6) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); h->prof_events.push_back(e); }
  if (!h->prof_span0) { HIP_TRY(hipEventCreate(&h->prof_span0)); HIP_TRY(hipEventCreate(&h->prof_span1)); }
  h->prof_cap = max_sweeps; h->prof_used = 0; h->prof_seen = 0; h->prof_stride = stride; h->prof_on = true; h->prof_closed = false;
  return 0;
}
int lifcal_ba_profile_begin(lifcal_ba_handle* h, uint32_t max_sweeps) { return lifcal_ba_profile_begin_sampled(h, max_sweeps, 1u); }

int lifcal_ba_profile_end(lifcal_ba_handle* h, lifcal_ba_profile* out) {
  if (!h || !out) return LIFCAL_BA_ERR_INVALID_ARG;
  const uint32_t n = h->prof_used, seen = h->prof_seen;
  if (seen && !h->prof_closed) HIP_TRY(hipEventRecord(h->prof_span1, h->stream));   // fewer sweeps than reserved: close the span now
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::memset(out, 0, sizeof(*out));
  for (uint32_t i = 0; i < n; ++i) {   // per sampled sweep: the dominant kernel's own start / stop stamps (events 1, 2)
    float bms = 0;
    hipEvent_t* e = &h->prof_events[(size_t)i * 6];
    HIP_TRY(hipEventElapsedTime(&bms, e[1], e[2]));
    out->ms_accumulate += bms;
    if (h->opt.world_size > 1 || h->force_exchange) { float xms = 0; if (hipEventElapsedTime(&xms, e[3], e[4]) == hipSuccess) out->ms_exchange += xms; }
  }
  if (seen) {
    float t = 0;   // first kernel of the first sweep -> end of the last sweep of the span (includes the gaps between sweeps)
    HIP_TRY(hipEventElapsedTime(&t, h->prof_span0, h->prof_span1));
    out->ms_total = t / seen;
    if (n) { out->ms_accumulate /= n; out->ms_exchange /= n; }
    out->special_points = (double)h->d.n_special;           // (their kernels run outside the dominant kernel's time stamps)
    out->ms_schur = out->ms_total - out->ms_accumulate;     // everything outside the dominant kernel: tables, finalize, special points, exchange, launch gaps
  }
  out->n_sweeps = seen;
  out->n_sampled = n;
  h->prof_on = false;
  return 0;
}

int lifcal_ba_sweep(lifcal_ba_handle* h, double radius, lifcal_ba_sweep_out* out) {
  if (!h || !out || !(radius > 0.0)) return LIFCAL_BA_ERR_INVALID_ARG;
  Dev& d = h->d;
  HIP_TRY(hipSetDevice(h->opt.device));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  if (int rc = launch_sweep(h, radius)) return rc;
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  double cost, gmax, bad;
  if (int rc = read_sweep_scalars(h, &cost, &gmax, &bad)) return rc;
  float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  out->cost = cost; out->gradient_max_norm = gmax; out->seconds = ms * 1e-3;
  out->n_reduced = h->plan.n_red_canon; out->n_promoted = h->plan.Q;
  h->last_cost = cost; h->last_gmax = gmax;
  const uint32_t F6 = 6 * d.F, Q3 = 3 * d.Q, nci = d.n_red, ncan = h->plan.n_red_canon;
  // internal index (poses | promoted | camera) -> canonical index (camera 17 | poses | promoted)
  auto canon = [&](uint32_t t) -> uint32_t { if (t < F6) return 17 + t; if (t < F6 + Q3) return 17 + F6 + (t - F6); return t - F6 - Q3; };
  if (out->S || out->rhs || out->gradient_reduced) {
    std::vector<double> band((size_t)d.F * (d.bw + 1) * 36), arrow((size_t)(d.NA + 1) * d.ld), gB(nci);
    HIP_TRY(hipMemcpy(band.data(), d.Sband, band.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(arrow.data(), d.Sarrow, arrow.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(gB.data(), d.gB, gB.size() * 8, hipMemcpyDeviceToHost));
    if (out->S) {
      std::fill(out->S, out->S + (size_t)ncan * ncan, 0.0);
      for (uint32_t k = 0; k < ncan; ++k) out->S[(size_t)k * ncan + k] = 1.0;  // dead camera slots 9..16 etc.
      auto put = [&](uint32_t r, uint32_t c, double v) { const uint32_t a = canon(r), b = canon(c); out->S[(size_t)a * ncan + b] = v; out->S[(size_t)b * ncan + a] = v; };
      for (uint32_t f = 0; f < d.F; ++f)
        for (uint32_t dd = 0; dd <= std::min(d.bw, f); ++dd)
          for (uint32_t i = 0; i < 6; ++i)
            for (uint32_t j = 0; j < 6; ++j) {
              if (dd == 0 && j > i) continue;
              put(6 * f + i, 6 * (f - dd) + j, band[((size_t)f * (d.bw + 1) + dd) * 36 + i * 6 + j]);
            }
      for (uint32_t a = 0; a < d.NA; ++a)
        for (uint32_t c = 0; c <= F6 + a; ++c) put(F6 + a, c, arrow[(size_t)a * d.ld + c]);
    }
    if (out->rhs) { std::fill(out->rhs, out->rhs + ncan, 0.0); for (uint32_t t = 0; t < nci; ++t) out->rhs[canon(t)] = arrow[(size_t)d.NA * d.ld + t]; }
    if (out->gradient_reduced) { std::fill(out->gradient_reduced, out->gradient_reduced + ncan, 0.0); for (uint32_t t = 0; t < nci; ++t) out->gradient_reduced[canon(t)] = gB[t]; }
  }
  if (out->point_gradient || out->point_hessian_inv) {
    std::vector<double> acc((size_t)d.P * 36), ui((size_t)d.P * 9);
    HIP_TRY(hipMemcpy(acc.data(), d.ptacc, acc.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ui.data(), d.Uinv, ui.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t q = 0; q < d.P; ++q) {
      const bool mine = d.use_points && h->plan.promoted[q] < 0 && h->plan.owner[q] == h->opt.rank && (h->plan.pt_nslots[q] > 0 || (h->plan.pt_cons0.size() > q + 1 && h->plan.pt_cons0[q + 1] > h->plan.pt_cons0[q]));
      if (out->point_gradient) for (int k = 0; k < 3; ++k) out->point_gradient[3 * (size_t)q + k] = mine ? acc[(size_t)q * 36 + 6 + k] : 0.0;
      if (out->point_hessian_inv) for (int k = 0; k < 9; ++k) out->point_hessian_inv[9 * (size_t)q + k] = mine ? ui[(size_t)q * 9 + k] : 0.0;
    }
    if (out->point_gradient && d.use_points) {  // promoted points: gradient lives in the reduced block
      std::vector<double> gB(nci);
      HIP_TRY(hipMemcpy(gB.data(), d.gB, gB.size() * 8, hipMemcpyDeviceToHost));
      for (uint32_t qq = 0; qq < d.Q; ++qq) for (int k = 0; k < 3; ++k) out->point_gradient[3 * (size_t)h->plan.promoted_ids[qq] + k] = gB[F6 + 3 * qq + k];
    }
  }
  if (!std::isfinite(cost)) return LIFCAL_BA_ERR_NUMERIC;
  return 0;
}

// The LM loop with the decisions on the device (k_lm_control / k_lm_commit, kernels.hpp): per iteration the host enqueues
//   linear solve | candidate + its cost | control (writes the state into mapped host memory) | commit | the NEXT sweep (at whatever
//   point / radius control chose)
// and polls the round number of the mirrored state — behind it the next sweep is already running, so the queue never drains
// while the host decides and launches (round 2: two blocking read-backs per iteration, ~250 of ~550 us per iteration were
// host-induced idle time), and no copy kernel or event record sits between the kernels of an iteration (each a barrier packet:
// ~5 us of idle queue; the split of the loop's time comes from s_memrealtime stamps taken by the kernels themselves).  The sweep
// enqueued behind the terminating decision is the only wasted work.  Unbounded problems on one rank.
static int solve_device_loop(lifcal_ba_handle* h, lifcal_ba_summary* s, double t_start) {
  Dev& d = h->d;
  const lifcal_ba_options& o = h->opt;
  double lm0[LM_N];
  for (int i = 0; i < LM_N; ++i) lm0[i] = 0.0;
  lm0[LM_RADIUS] = o.initial_radius; lm0[LM_DECREASE] = 2.0; lm0[LM_STEP_OK] = 1.0; lm0[LM_FRESH] = 1.0; lm0[LM_INITIAL_COST] = -1.0;
  std::memcpy(h->h_lm + LM_N, lm0, sizeof(lm0));   // (staging half of the mapped buffer: no synchronisation before the loop starts)
  HIP_TRY(hipMemcpyAsync(d.lm, h->h_lm + LM_N, sizeof(lm0), hipMemcpyHostToDevice, h->stream));
  const LmOpts lo{o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance, o.min_relative_decrease, o.max_radius, o.min_radius, o.max_iterations};
  double t0 = now_s();
  if (int rc = launch_sweep(h, -1.0)) return rc;
  const uint32_t commit_grid = std::max(1u, std::min(1024u, (3 * d.P + 6 * d.F + 255) / 256));
  volatile double* mirror = h->h_lm;
  mirror[LM_SEQ] = 0.0;
  for (int round = 0; round < o.max_iterations + 8; ++round) {
    if (int rc = launch_linear_solve(h)) return rc;
    if (int rc = launch_candidate(h)) return rc;
    hipLaunchKernelGGL(k_lm_control, dim3(1), dim3(64), 0, h->stream, d, lo, (const double*)h->partial, h->h_lm_dev, (double)(round + 1));
    hipLaunchKernelGGL(k_lm_commit, dim3(commit_grid), dim3(256), 0, h->stream, d);
    HIP_TRY(hipGetLastError());
    if (int rc = launch_sweep(h, -1.0)) return rc;         // speculative: runs while the host looks at the state
    // wait for this round's state: one word of mapped host memory, written last by k_lm_control.  A queue that died would never
    // write it: every ~2 ms of waiting the stream is asked for its status (an error ends the wait, "not ready" continues it)
    const double t_wait = now_s();
    double t_check = t_wait;
    while (mirror[LM_SEQ] != (double)(round + 1)) {
      __builtin_ia32_pause();
      const double t = now_s();
      if (t - t_check > 2e-3) {
        t_check = t;
        const hipError_t q = hipStreamQuery(h->stream);
        if (q != hipSuccess && q != hipErrorNotReady) { g_last_error = std::string("device LM loop: ") + hipGetErrorString(q); return LIFCAL_BA_ERR_HIP; }
        if (q == hipSuccess && mirror[LM_SEQ] != (double)(round + 1)) { g_last_error = "device LM loop: the state mirror was not written"; return LIFCAL_BA_ERR_HIP; }
      }
    }
    if (mirror[LM_TERMINATION] != 0.0) break;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double* lm = h->h_lm;
  if (lm[LM_TERMINATION] < 0.0) { g_last_error = "non-finite cost at the initial point"; return LIFCAL_BA_ERR_NUMERIC; }
  s->initial_cost = lm[LM_INITIAL_COST]; s->final_cost = lm[LM_X_COST]; s->final_radius = lm[LM_RADIUS]; s->final_gradient_max_norm = lm[LM_GMAX];
  s->iterations = (int32_t)lm[LM_ITER]; s->successful_steps = (int32_t)lm[LM_SUCCESSFUL]; s->unsuccessful_steps = (int32_t)lm[LM_UNSUCCESSFUL];
  s->termination = lm[LM_TERMINATION] != 0.0 ? (int32_t)lm[LM_TERMINATION] : LIFCAL_BA_TERM_MAX_ITERATIONS;
  if (int rc = download_parameters(h)) return rc;
  s->seconds_total = now_s() - t_start;
  s->seconds_linear_solve = 1e-8 * lm[LM_TICKS_LINEAR];                    // linear solve + candidate evaluation: 100 MHz ticks from k_finalize's start to k_lm_control's
  s->seconds_sweep = std::max(0.0, (now_s() - t0) - s->seconds_linear_solve);   // everything else of the loop: sweeps, control, the read-backs
  return 0;
}

int lifcal_ba_solve(lifcal_ba_handle* h, lifcal_ba_summary* s) {
  if (!h || !s) return LIFCAL_BA_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->opt.device));
  const lifcal_ba_options& o = h->opt;
  const double t_start = now_s();
  std::memset(s, 0, sizeof(*s));
  if (int rc = upload_parameters(h)) return rc;
  // decisions on the device where the host has none of its own to make (LIFCAL_HOST_LM=1: the host loop below, e.g. for LIFCAL_TRACE)
  if (!h->constrained && o.world_size == 1 && o.precision == 0 && !h->trace && !o.verbose && getenv("LIFCAL_HOST_LM") == nullptr)
    return solve_device_loop(h, s, t_start);
  double radius = o.initial_radius, decrease_factor = 2.0;
  double x_cost, gmax, bad;
  double t0 = now_s();
  if (int rc = launch_sweep(h, radius)) return rc;
  if (int rc = read_sweep_scalars(h, &x_cost, &gmax, &bad)) return rc;
  if (o.precision == 1) { if (int rc = cost64_current(h, &x_cost)) return rc; }
  s->seconds_sweep += now_s() - t0;
  if (!std::isfinite(x_cost)) { g_last_error = "non-finite cost at the initial point"; return LIFCAL_BA_ERR_NUMERIC; }
  s->initial_cost = x_cost;
  int iteration = 0, invalid_steps = 0;
  bool step_successful = true, system_ready = true;
  if (o.verbose) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e    0.00e+00 %10.2e   0.00e+00   0.00e+00 %9.2e\n", 0, x_cost, gmax, radius);
  s->termination = LIFCAL_BA_TERM_NONE;
  if (gmax <= o.gradient_tolerance) s->termination = LIFCAL_BA_TERM_GRADIENT_TOLERANCE;
  while (s->termination == LIFCAL_BA_TERM_NONE) {
    if (iteration >= o.max_iterations) { s->termination = LIFCAL_BA_TERM_MAX_ITERATIONS; break; }
    if (step_successful && gmax <= o.gradient_tolerance) { s->termination = LIFCAL_BA_TERM_GRADIENT_TOLERANCE; break; }
    if (radius < o.min_radius) { s->termination = LIFCAL_BA_TERM_MIN_RADIUS; break; }
    ++iteration;
    if (!system_ready) {  // same point, new radius: the fused sweep is cheap enough to simply run again
      t0 = now_s();
      if (int rc = launch_sweep(h, radius)) return rc;
      s->seconds_sweep += now_s() - t0;
    }
    system_ready = false;
    t0 = now_s();
    if (int rc = launch_linear_solve(h)) return rc;
    if (int rc = launch_candidate(h)) return rc;
    StepScalars st;
    if (int rc = read_step_scalars(h, &st)) return rc;
    s->seconds_linear_solve += now_s() - t0;
    // model_cost_change = -g^T d - 1/2 d^T J^T J d with (J^T J + Lambda) d = -g  =>  1/2 (d^T Lambda d - g^T d)
    const double model_cost_change = 0.5 * (st.ddd - st.gtd);
    if (h->trace) fprintf(stderr, "[lifcal_ba r%d] it %d radius %.17g x_cost %.17g cand %.17g gtd %.17g ddd %.17g step2 %.17g chol_fail %g bad %g\n", o.rank, iteration, radius, x_cost, st.cand_cost, st.gtd, st.ddd, st.step2, st.chol_fail, bad);
    const bool valid = st.chol_fail == 0.0 && bad == 0.0 && std::isfinite(model_cost_change) && model_cost_change > 0.0;
    if (!valid) {
      if (++invalid_steps >= 5) { s->termination = LIFCAL_BA_TERM_INVALID_STEPS; break; }
      radius *= 0.5; step_successful = false; ++s->unsuccessful_steps;
      bad = 0.0;
      continue;
    }
    invalid_steps = 0;
    double cand_cost = st.cand_cost;
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    double step2 = st.step2, x2 = st.x2;
    if (h->constrained) {
      // ceres TrustRegionMinimizer::DoLineSearch: Armijo along the projected step, CUBIC interpolation, at most 20 trials;
      // phi(1) is the candidate cost already evaluated, the gradient at a trial point is only needed when the test fails
      const double g0 = st.gtd;
      const double suff = 1e-4;
      bool armijo_ok = std::isfinite(st.cand_cost) && st.cand_cost <= x_cost + suff * g0 * 1.0;
      if (!armijo_ok) {
        LsSample init, prev, cur;
        init.x = 0; init.value = x_cost; init.gradient = g0; init.value_valid = init.gradient_valid = true;
        if (int rc = eval_trial(h, 1.0, radius, &cur)) return rc;
        // max |delta| over ALL columns for the minimum-step test (ceres LineSearch::min_step_size / max_abs(direction)):
        // the camera + pose part is replicated, the point part is spread over the ranks -> per-rank slots, all-reduced.
        // (A rank-local maximum here once let one rank leave the search while the other entered the next trial's all-reduce.)
        double dir_max = 0;
        {
          Dev& d = h->d;
          HIP_TRY(hipMemsetAsync(h->dirmax_buf, 0, 65 * sizeof(double), h->stream));
          const uint32_t n = std::max(std::max(d.n_red, d.n_owned), 1u);
          hipLaunchKernelGGL(k_dir_max, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, (unsigned long long*)h->dirmax_buf);
          HIP_TRY(hipGetLastError());
          if (int rc = do_allreduce(h, h->dirmax_buf, 65)) return rc;
          double hm[65];
          HIP_TRY(hipMemcpyAsync(hm, h->dirmax_buf, sizeof(hm), hipMemcpyDeviceToHost, h->stream));
          HIP_TRY(hipStreamSynchronize(h->stream));
          for (int r = 0; r < o.world_size; ++r) dir_max = std::max(dir_max, hm[r]);
          dir_max = std::max(dir_max, hm[64]);
        }
        int ls_iter = 0; bool ls_ok = true;
        if (h->trace) fprintf(stderr, "[lifcal_ba r%d] it %d line search: dir_max %.17g phi(1) %.17g phi'(1) %.17g\n", o.rank, iteration, dir_max, cur.value, cur.gradient);
        while (!cur.value_valid || cur.value > x_cost + suff * g0 * cur.x) {
          if (++ls_iter >= 20) { ls_ok = false; break; }
          const double lo_b = 1e-3 * cur.x, hi_b = 0.6 * cur.x;
          double tnew;
          if (!cur.value_valid) tnew = std::min(std::max(cur.x * 0.5, lo_b), hi_b);
          else { std::vector<LsSample> smp{init, cur}; if (prev.value_valid) smp.push_back(prev); tnew = ls_minimize(smp, lo_b, hi_b); }
          if (tnew * dir_max < 1e-9) { ls_ok = false; break; }
          prev = cur;
          if (int rc = eval_trial(h, tnew, radius, &cur)) return rc;
          if (h->trace) fprintf(stderr, "[lifcal_ba r%d] it %d line search trial %d: t %.17g phi %.17g phi' %.17g\n", o.rank, iteration, ls_iter, tnew, cur.value, cur.gradient);
        }
        const double t_opt = ls_ok ? cur.x : 1.0;
        // candidate at the chosen step length, its cost and the step norm (the model cost change stays that of the full step)
        if (int rc = launch_apply_step(h, t_opt)) return rc;
        HIP_TRY(hipMemsetAsync(h->partial, 0, 8 * sizeof(double), h->stream));
        if (int rc = launch_tables(h, h->d.cam_c, h->d.views_c, h->d.camc_c, h->d.ft_c, h->d.lt_c, false, true)) return rc;
        {
          Dev& d = h->d;
          const double* pts_eval = d.use_points ? d.pts_c : d.pts;
          for (const TileSet* ts : {&h->ts1, &h->ts2}) {
            if (!ts->n_tiles) continue;
            const uint32_t grid = std::max(1u, std::min((ts->n_tiles + 3) / 4, 1024u));
#define CALL_COST2(NR, TAN, ADJ) hipLaunchKernelGGL((k_cost<NR, TAN, ADJ>), dim3(grid), dim3(256), 0, h->stream, d, *ts, (const CamConsts*)d.camc_c, (const double*)d.ft_c, (const double*)d.lt_c, pts_eval, h->partial + 4)
            DISPATCH_CFG(h, CALL_COST2);
#undef CALL_COST2
            if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, grid, 1u, h->partial + 4);
          }
          if (d.M_local) hipLaunchKernelGGL(k_constraints, dim3(d.deterministic ? 1 : (d.M_local + 63) / 64), dim3(d.deterministic ? 1 : 64), 0, h->stream, d, 1, pts_eval, h->partial + 4);
          HIP_TRY(hipGetLastError());
          if (int rc = do_allreduce(h, h->partial, 8)) return rc;
          if (int rc = do_allreduce(h, h->ls_buf, 2)) return rc;
          double hb[8], hp[8];
          HIP_TRY(hipMemcpyAsync(hb, h->ls_buf, sizeof(hb), hipMemcpyDeviceToHost, h->stream));
          HIP_TRY(hipMemcpyAsync(hp, h->partial, sizeof(hp), hipMemcpyDeviceToHost, h->stream));
          HIP_TRY(hipStreamSynchronize(h->stream));
          cand_cost = std::isfinite(hp[4]) ? hp[4] : std::numeric_limits<double>::max();
          step2 = hb[0]; x2 = hb[1];
        }
        system_ready = false;   // the trial sweeps overwrote the blocks of the current point
        if (getenv("LIFCAL_DEBUG_LS")) fprintf(stderr, "[lifcal_ba] line search: %d backtracks, t = %.6g\n", ls_iter, t_opt);
      }
    }
    const double step_norm = std::sqrt(step2), x_norm = std::sqrt(x2);
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { s->termination = LIFCAL_BA_TERM_PARAMETER_TOLERANCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= o.function_tolerance * x_cost) { s->termination = LIFCAL_BA_TERM_FUNCTION_TOLERANCE; break; }
    const double rel = (cand_cost >= std::numeric_limits<double>::max()) ? std::numeric_limits<double>::lowest() : cost_change / model_cost_change;
    if (rel > o.min_relative_decrease) {
      swap_current_candidate(h);
      t0 = now_s();
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
      radius = std::min(o.max_radius, radius);
      decrease_factor = 2.0;
      if (int rc = launch_sweep(h, radius)) return rc;
      if (int rc = read_sweep_scalars(h, &x_cost, &gmax, &bad)) return rc;
      if (o.precision == 1) x_cost = cand_cost;   // the fp64 cost of the point just accepted (see cost64_current)
      s->seconds_sweep += now_s() - t0;
      system_ready = true; step_successful = true; ++s->successful_steps;
    } else {
      radius = radius / decrease_factor; decrease_factor *= 2.0; step_successful = false; ++s->unsuccessful_steps;
    }
    if (o.verbose) printf("%4d % .6e   % .2e %10.2e  %9.2e  %9.2e %9.2e\n", iteration, x_cost, cost_change, gmax, step_norm, rel, radius);
  }
  if (h->trace) fprintf(stderr, "[lifcal_ba r%d] done: it %d termination %d cost %.17g\n", o.rank, iteration, (int)s->termination, x_cost);
  if (int rc = download_parameters(h)) return rc;
  s->iterations = iteration; s->final_cost = x_cost; s->final_radius = radius; s->final_gradient_max_norm = gmax;
  s->seconds_total = now_s() - t_start;
  return 0;
}

int lifcal_ba_reproj_stats(lifcal_ba_handle* h, double thr, lifcal_ba_stats* out) {
  if (!h || !out) return LIFCAL_BA_ERR_INVALID_ARG;
  Dev& d = h->d;
  HIP_TRY(hipSetDevice(h->opt.device));
  // reference :1028-1039: parameters are used as stored (no sign folding), scale goes through a float cast
  if (int rc = launch_tables(h, d.cam, d.views, h->camc_stats, d.ft_c, d.lt_c, false, false)) return rc;
  HIP_TRY(hipMemsetAsync(h->stats_buf, 0, 8 * sizeof(double), h->stream));
  for (const TileSet* ts : {&h->ts1, &h->ts2}) {
    if (!ts->n_tiles) continue;
    const uint32_t grid = std::max(1u, std::min((ts->n_tiles + 3) / 4, 1024u));
#define CALL_STATS(NR, TAN, ADJ) hipLaunchKernelGGL((k_stats<NR, TAN, ADJ>), dim3(grid), dim3(256), 0, h->stream, d, *ts, (const CamConsts*)h->camc_stats, (const double*)d.ft_c, (const double*)d.lt_c, (const double*)d.pts, thr * thr, h->stats_buf, (unsigned long long*)(h->stats_buf + 4))
    DISPATCH_CFG(h, CALL_STATS);
#undef CALL_STATS
    if (d.deterministic) hipLaunchKernelGGL(k_det_sum, dim3(1), dim3(64), 0, h->stream, (const double*)d.det_slots, grid, 4u, h->stats_buf);
  }
  HIP_TRY(hipGetLastError());
  double hb[8];
  HIP_TRY(hipMemcpyAsync(hb, h->stats_buf, sizeof(hb), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  double mx, my; std::memcpy(&mx, &hb[4], 8); std::memcpy(&my, &hb[5], 8);
  if (h->opt.world_size > 1) {
    // reference :1083-1084 takes the maxima over ALL observations: sums are all-reduced, the two maxima travel in one slot
    // pair per rank (the other ranks' slots are 0, so the sum all-reduce moves them exactly) and the host takes the largest
    std::vector<double> slots(4 + 2 * (size_t)h->opt.world_size, 0.0);
    for (int k = 0; k < 4; ++k) slots[k] = hb[k];
    slots[4 + 2 * (size_t)h->opt.rank] = mx; slots[5 + 2 * (size_t)h->opt.rank] = my;
    HIP_TRY(hipMemcpyAsync(h->stats_slots, slots.data(), slots.size() * 8, hipMemcpyHostToDevice, h->stream));
    if (int rc = do_allreduce(h, h->stats_slots, slots.size())) return rc;
    HIP_TRY(hipMemcpyAsync(slots.data(), h->stats_slots, slots.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int k = 0; k < 4; ++k) hb[k] = slots[k];
    mx = my = 0.0;
    for (int r = 0; r < h->opt.world_size; ++r) { mx = std::max(mx, slots[4 + 2 * (size_t)r]); my = std::max(my, slots[5 + 2 * (size_t)r]); }
  }
  const double n = hb[2];
  out->std_x = std::sqrt(hb[0] / n); out->std_y = std::sqrt(hb[1] / n); out->mae_x = mx; out->mae_y = my;
  out->num_points = (uint32_t)n; out->num_inliers = (uint32_t)hb[3];
  return 0;
}

// x_proj / y_proj of reference storeRawImagePointsCsv (src/CameraCalibration.cpp:1504-1538): the model's projection of every
// observation at the stored parameters, evaluated like calcReprojectionError evaluates it (:1028-1039)
int lifcal_ba_project_observations(lifcal_ba_handle* h, double* x_proj, double* y_proj) {
  if (!h || !x_proj || !y_proj) return LIFCAL_BA_ERR_INVALID_ARG;
  Dev& d = h->d;
  const Plan& L = h->plan;
  const size_t n = h->prob.n_obs;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(h->opt.device));
  if (int rc = launch_tables(h, d.cam, d.views, h->camc_stats, d.ft_c, d.lt_c, false, false)) return rc;
  uint32_t *src1 = nullptr, *src2 = nullptr; double* out = nullptr;
  auto release = [&]() { for (void* q : {(void*)src1, (void*)src2, (void*)out}) if (q) (void)hipFree(q); };
  auto up = [&](uint32_t** q, const std::vector<uint32_t>& v) -> hipError_t {
    if (v.empty()) return hipSuccess;
    hipError_t e = hipMalloc((void**)q, v.size() * 4);
    return e == hipSuccess ? hipMemcpyAsync(*q, v.data(), v.size() * 4, hipMemcpyHostToDevice, h->stream) : e;
  };
  hipError_t e = up(&src1, L.ell_src);
  if (e == hipSuccess) e = up(&src2, L.v2_src);
  if (e == hipSuccess) e = hipMalloc((void**)&out, 2 * n * 8);
  if (e == hipSuccess) e = hipMemsetAsync(out, 0xff, 2 * n * 8, h->stream);   // NaN where another rank owns the observation
  if (e == hipSuccess) {
    const TileSet* sets[2] = {&h->ts1, &h->ts2};
    const uint32_t* srcs[2] = {src1, src2};
    for (int k = 0; k < 2; ++k) {
      const TileSet* ts = sets[k];
      if (!ts->n_tiles || !srcs[k]) continue;
      const uint32_t grid = std::max(1u, std::min((ts->n_tiles + 3) / 4, 1024u));
#define CALL_PROJ(NR, TAN, ADJ) hipLaunchKernelGGL((k_project_obs<NR, TAN, ADJ>), dim3(grid), dim3(256), 0, h->stream, d, *ts, srcs[k], (const CamConsts*)h->camc_stats, (const double*)d.ft_c, (const double*)d.lt_c, (const double*)d.pts, out, out + n)
      DISPATCH_CFG(h, CALL_PROJ);
#undef CALL_PROJ
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(x_proj, out, n * 8, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(y_proj, out + n, n * 8, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  release();
  if (e != hipSuccess) { g_last_error = std::string("lifcal_ba_project_observations: ") + hipGetErrorString(e); return LIFCAL_BA_ERR_HIP; }
  return 0;
}

}  // extern "C"

// micro-lens grid, lens maps, epipolar web and projectPointsToRawImage (include/lifcal_mla.h)
#include "mla.hpp"

// LiFCal's result files (include/lifcal_io.h)
#include "writers.hpp"

// COLMAP sparse-model ingestion (include/lifcal_colmap.h)
#include "colmap.hpp"

// frame-windowed solve for long sequences (include/lifcal_ba.h)
#include "windowed.hpp"
