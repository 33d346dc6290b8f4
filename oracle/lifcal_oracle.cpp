// oracle/lifcal_oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of LiFCal's bundle-adjustment hot path.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this; the product (lifcal_amd/) never does.
//
//   problem assembly    <- reference src/CameraCalibration.cpp:858-953 (performBundleAdjustment)
//   residual + Jacobian <- reference src/BundleAdjustment/BundleAdjustment.h:120-222 evaluated with
//                          dual numbers (oracle/jet.hpp) exactly as ceres::AutoDiffCostFunction does
//   loss                <- ceres::CauchyLoss(0.5) + Corrector (call sites :892,:899,:909)
//   solver              <- ceres::Solve with DENSE_SCHUR, f_tol 1e-6, p_tol 1e-8, 200 iterations (:955-965)
//   stats               <- reference src/CameraCalibration.cpp:1026-1103 (calcReprojectionError)
//
// Ceres Solver 2.1.0 (reference installation/Dockerfile:105) and Eigen 3 are third-party
// dependencies that are NOT under /root/reference and NOT installed; their published algorithms
// (TrustRegionMinimizer, LevenbergMarquardtStrategy, SchurEliminator, DenseSchurComplementSolver,
// Corrector, ParameterBlock::Plus box projection, ArmijoLineSearch) are restated here.
//
// PARITY UNPINNED: the reference has no tests, fixtures or golden vectors for this path and cannot
// be compiled in this image, so nothing external pins this restatement.  It is cross-checked by an
// independent arbitrary-precision restatement (tests/test_oracle_model.py) and by invariants only.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../include/lifcal_ba.h"
#include "lifcal_oracle.h"
#include "model.hpp"
#include "analytic.hpp"

namespace {

using lo::Config;
using lo::Jet;
using lo::ObsFunctor;

constexpr int NC = LIFCAL_BA_MAX_CAMERA_PARAMETERS;  // 17

template <class F>
void parallel_for(int n_threads, int64_t n, F&& body) {
  if (n_threads <= 1 || n < 2 * n_threads) { body(0, n, 0); return; }
  std::vector<std::thread> th;
  const int64_t chunk = (n + n_threads - 1) / n_threads;
  for (int t = 0; t < n_threads; ++t) {
    const int64_t lo_ = t * chunk, hi_ = std::min<int64_t>(n, lo_ + chunk);
    if (lo_ >= hi_) break;
    th.emplace_back([&, lo_, hi_, t] { body(lo_, hi_, t); });
  }
  for (auto& x : th) x.join();
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// frames whose pose is held constant (lo_set_fixed_frames; mirrors lifcal_ba_set_fixed_frames): ceres SetParameterBlockConstant on
// views + 6 f — the residual blocks of that frame stay, their pose Jacobian does not exist.  Test-infrastructure global.
std::vector<uint8_t> g_fixed_frames;

// ---------------------------------------------------------------------------------------------
// Problem structure
// ---------------------------------------------------------------------------------------------
struct Structure {
  const lifcal_ba_problem* p;
  Config cfg;
  int F, P, N, M;
  bool use_poses, use_points, use_constraints;
  std::vector<int> promoted;        // per point: index among promoted points or -1
  std::vector<int> promoted_ids;
  int n_prom, n_red;
  std::vector<uint8_t> cam_free;    // 17: column takes part in the solve
  std::vector<int> pt_begin, pt_obs;  // CSR point -> observation indices (input order)
  std::vector<std::vector<int>> pt_cons;  // constraints touching each point
  std::vector<uint8_t> pt_used, fr_used, fr_fixed;

  explicit Structure(const lifcal_ba_problem* pr) : p(pr), cfg(pr->config) {
    F = pr->n_frames; P = pr->n_points; N = pr->n_obs; M = pr->n_constraints;
    use_poses = cfg.refine_poses;
    // reference :879-912: points are parameter blocks only in the <2,17,6,3> arity
    use_points = cfg.refine_poses && cfg.refine_points;
    use_constraints = use_points && pr->use_constraints && M > 0 && pr->c_i && pr->c_j;
    promoted.assign(P, -1);
    if (use_constraints) {
      std::vector<uint8_t> flag(P, 0);
      for (int c = 0; c < M; ++c) flag[pr->c_j[c]] = 1;  // one endpoint of every pair leaves the e-block set
      for (int i = 0; i < P; ++i) if (flag[i]) { promoted[i] = (int)promoted_ids.size(); promoted_ids.push_back(i); }
    }
    n_prom = (int)promoted_ids.size();
    n_red = NC + 6 * F + 3 * n_prom;
    cam_free.assign(NC, 0);
    for (int j = 0; j < cfg.n_camera; ++j) cam_free[j] = ((pr->fixed_mask >> j) & 1u) ? 0 : 1;
    pt_begin.assign(P + 1, 0);
    pt_used.assign(P, 0); fr_used.assign(F, 0);
    for (int i = 0; i < N; ++i) { pt_begin[pr->pt[i] + 1]++; pt_used[pr->pt[i]] = 1; fr_used[pr->fr[i]] = 1; }
    fr_fixed.assign(F, 0);
    for (int f = 0; f < F && f < (int)g_fixed_frames.size(); ++f) if (g_fixed_frames[f]) { fr_fixed[f] = 1; fr_used[f] = 0; }   // constant pose: not a column
    for (int i = 0; i < P; ++i) pt_begin[i + 1] += pt_begin[i];
    pt_obs.resize(N);
    std::vector<int> fill(pt_begin.begin(), pt_begin.end() - 1);
    for (int i = 0; i < N; ++i) pt_obs[fill[pr->pt[i]]++] = i;
    pt_cons.resize(P);
    if (use_constraints)
      for (int c = 0; c < M; ++c) {
        pt_cons[pr->c_i[c]].push_back(c); pt_cons[pr->c_j[c]].push_back(c);
        pt_used[pr->c_i[c]] = 1; pt_used[pr->c_j[c]] = 1;
      }
  }
  // analytic arm: unique micro-lens centres (exact bit patterns) and the lens of every observation, built on first use
  mutable std::vector<int> lens_of_obs; mutable std::vector<double> lens_xy;
  void build_lenses() const {
    if (!lens_of_obs.empty() || N == 0) return;
    lens_of_obs.resize(N);
    struct Key { uint64_t a, b; bool operator==(const Key& o) const { return a == o.a && b == o.b; } };
    struct Hash { size_t operator()(const Key& k) const { return (size_t)(k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0x7F4A7C15ull + (k.a << 6))); } };
    std::unordered_map<Key, int, Hash> map;
    map.reserve(1 << 16);
    for (int i = 0; i < N; ++i) {
      Key k; std::memcpy(&k.a, &p->mcx[i], 8); std::memcpy(&k.b, &p->mcy[i], 8);
      auto it = map.find(k);
      if (it == map.end()) { it = map.emplace(k, (int)(lens_xy.size() / 2)).first; lens_xy.push_back(p->mcx[i]); lens_xy.push_back(p->mcy[i]); }
      lens_of_obs[i] = it->second;
    }
  }
  int view_col(int f) const { return NC + 6 * f; }
  int prom_col(int k) const { return NC + 6 * F + 3 * k; }
};

int validate(const lifcal_ba_problem* p) {
  if (!p || !p->cam || (p->n_obs && (!p->u || !p->v || !p->mcx || !p->mcy || !p->pt || !p->fr))) return LIFCAL_BA_ERR_INVALID_ARG;
  if ((p->n_frames && !p->views) || (p->n_points && !p->pts)) return LIFCAL_BA_ERR_INVALID_ARG;
  for (uint32_t i = 0; i < p->n_obs; ++i)
    if (p->pt[i] >= p->n_points || p->fr[i] >= p->n_frames) return LIFCAL_BA_ERR_OUT_OF_RANGE;
  if (p->n_constraints && p->use_constraints && p->c_i && p->c_j)
    for (uint32_t c = 0; c < p->n_constraints; ++c) {
      if (p->c_i[c] >= p->n_points || p->c_j[c] >= p->n_points) return LIFCAL_BA_ERR_OUT_OF_RANGE;
      if (p->c_i[c] == p->c_j[c]) return LIFCAL_BA_ERR_INVALID_ARG;
    }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Evaluation (ceres::ProgramEvaluator + AutoDiffCostFunction + CauchyLoss/Corrector)
// ---------------------------------------------------------------------------------------------
struct Eval {
  double cost = 0.0;
  std::vector<double> r;   // 2N loss-corrected residuals
  std::vector<double> Jc;  // N x 2 x 17
  std::vector<double> Jv;  // N x 2 x 6
  std::vector<double> Jp;  // N x 2 x 3
  std::vector<double> cr;  // M constraint residuals
  std::vector<double> cJ;  // M x 6  (d/dP_i | d/dP_j)
};

ObsFunctor make_functor(const Structure& s, int i, const double* views, const double* pts) {
  const lifcal_ba_problem* p = s.p;
  ObsFunctor f(p->config, p->u[i], p->v[i], p->spx, p->spy, p->scale, p->mcx[i], p->mcy[i]);
  if (!s.use_points) f.set_fixed_point(pts + 3 * p->pt[i]);
  if (!s.use_poses) f.set_fixed_view(views + 6 * p->fr[i]);
  return f;
}

template <int NJ>
void eval_jet(const ObsFunctor& f, const double* cam, const double* view, const double* pt,
              double r[2], double* Jc, double* Jv, double* Jp) {
  Jet<NJ> jc[NC], jv[6], jp[3], res[2];
  for (int k = 0; k < NC; ++k) jc[k] = Jet<NJ>(cam[k], k);
  if (view) for (int k = 0; k < 6; ++k) jv[k] = Jet<NJ>(view[k], NC + k);
  if (pt) for (int k = 0; k < 3; ++k) jp[k] = Jet<NJ>(pt[k], NC + 6 + k);
  f(jc, view ? jv : (const Jet<NJ>*)nullptr, pt ? jp : (const Jet<NJ>*)nullptr, res);
  for (int a = 0; a < 2; ++a) {
    r[a] = res[a].a;
    for (int k = 0; k < NC; ++k) Jc[a * NC + k] = res[a].v[k];
    if (view) for (int k = 0; k < 6; ++k) Jv[a * 6 + k] = res[a].v[NC + k];
    if (pt) for (int k = 0; k < 3; ++k) Jp[a * 3 + k] = res[a].v[NC + 6 + k];
  }
}

// one residual block: value only (double) or value + Jacobian (dual numbers)
void eval_obs(const Structure& s, int i, const double* cam, const double* views, const double* pts,
              bool want_jac, double r[2], double* Jc, double* Jv, double* Jp) {
  ObsFunctor f = make_functor(s, i, views, pts);
  const double* view = s.use_poses ? views + 6 * s.p->fr[i] : nullptr;
  const double* pt = s.use_points ? pts + 3 * s.p->pt[i] : nullptr;
  if (!want_jac) { f(cam, view, pt, r); return; }
  if (pt) eval_jet<26>(f, cam, view, pt, r, Jc, Jv, Jp);
  else if (view) eval_jet<23>(f, cam, view, nullptr, r, Jc, Jv, Jp);
  else eval_jet<17>(f, cam, nullptr, nullptr, r, Jc, Jv, Jp);
}

// ceres::CauchyLoss(a): b = a^2, c = 1/b; rho = b log(1 + s c), rho' = max(min_double, 1/(1+sc)), rho'' < 0
// ceres::Corrector: rho'' <= 0 -> residual and Jacobian are both scaled by sqrt(rho'), no second-order term.
inline void cauchy(double a, double s, double& rho0, double& rho1) {
  const double b = a * a, c = 1.0 / b;
  const double sum = 1.0 + s * c, inv = 1.0 / sum;
  rho0 = b * std::log(sum);
  rho1 = std::max(std::numeric_limits<double>::min(), inv);
}

double evaluate(const Structure& s, const double* cam, const double* views, const double* pts,
                double loss_scale, int threads, Eval* out /* null: cost only */, bool analytic = false) {
  const int N = s.N;
  const bool jac = out != nullptr;
  // analytic arm (oracle/analytic.hpp): camera constants once, one table entry per unique lens and per frame
  lo::ACam acam; std::vector<lo::ALens> alens; std::vector<lo::AFrame> aframes;
  analytic = analytic && jac;
  if (analytic) {
    s.build_lenses();
    lo::acam_prepare(cam, s.cfg, s.p->spx, s.p->spy, s.p->scale, acam);
    alens.resize(s.lens_xy.size() / 2); aframes.resize(s.F);
    parallel_for(threads, (int64_t)alens.size(), [&](int64_t lo_, int64_t hi_, int) { for (int64_t l = lo_; l < hi_; ++l) lo::alens_eval(acam, s.lens_xy[2 * l], s.lens_xy[2 * l + 1], true, alens[l]); });
    for (int f = 0; f < s.F; ++f) lo::aframe_eval(views + 6 * (size_t)f, aframes[f]);
  }
  if (jac) {
    out->r.assign(2 * (size_t)N, 0.0);
    out->Jc.assign((size_t)N * 2 * NC, 0.0);
    out->Jv.assign(s.use_poses ? (size_t)N * 12 : 0, 0.0);
    out->Jp.assign(s.use_points ? (size_t)N * 6 : 0, 0.0);
  }
  std::vector<double> partial(std::max(threads, 1), 0.0);
  parallel_for(threads, N, [&](int64_t lo_, int64_t hi_, int t) {
    double acc = 0.0;
    double jv_tmp[12], jp_tmp[6], jc_tmp[2 * NC];
    for (int64_t i = lo_; i < hi_; ++i) {
      double r[2];
      double* Jc = jac ? &out->Jc[(size_t)i * 2 * NC] : jc_tmp;
      double* Jv = (jac && s.use_poses) ? &out->Jv[(size_t)i * 12] : jv_tmp;
      double* Jp = (jac && s.use_points) ? &out->Jp[(size_t)i * 6] : jp_tmp;
      if (analytic) {
        const lo::AFrame& fr = aframes[s.p->fr[i]];
        const double* P = pts + 3 * (size_t)s.p->pt[i];
        double pc[3], Jpc[2][3], Jth[2][17];
        for (int a = 0; a < 3; ++a) pc[a] = fr.R[a][0] * P[0] + fr.R[a][1] * P[1] + fr.R[a][2] * P[2] + fr.t[a];
        lo::aobs_eval(acam, alens[s.lens_of_obs[i]], pc, s.p->u[i], s.p->v[i], r, Jpc, Jth);
        for (int a = 0; a < 2; ++a) {
          for (int k = 0; k < NC; ++k) Jc[a * NC + k] = Jth[a][k];
          if (s.use_poses) {
            for (int k = 0; k < 3; ++k) {   // d(R P)/d angle_k = dR_k P
              double g[3];
              for (int b = 0; b < 3; ++b) g[b] = fr.dR[k][b][0] * P[0] + fr.dR[k][b][1] * P[1] + fr.dR[k][b][2] * P[2];
              Jv[a * 6 + k] = Jpc[a][0] * g[0] + Jpc[a][1] * g[1] + Jpc[a][2] * g[2];
              Jv[a * 6 + 3 + k] = Jpc[a][k];
            }
          }
          if (s.use_points) for (int k = 0; k < 3; ++k) Jp[a * 3 + k] = Jpc[a][0] * fr.R[0][k] + Jpc[a][1] * fr.R[1][k] + Jpc[a][2] * fr.R[2][k];
        }
      } else {
        eval_obs(s, (int)i, cam, views, pts, jac, r, Jc, Jv, Jp);
      }
      if (jac && s.use_poses && s.fr_fixed[s.p->fr[i]]) for (int k = 0; k < 12; ++k) Jv[k] = 0.0;   // constant pose
      const double sq = r[0] * r[0] + r[1] * r[1];
      if (s.cfg.robust) {
        double rho0, rho1; cauchy(loss_scale, sq, rho0, rho1);
        acc += 0.5 * rho0;
        if (jac) {
          const double sc = std::sqrt(rho1);
          r[0] *= sc; r[1] *= sc;
          for (int k = 0; k < 2 * NC; ++k) Jc[k] *= sc;
          if (s.use_poses) for (int k = 0; k < 12; ++k) Jv[k] *= sc;
          if (s.use_points) for (int k = 0; k < 6; ++k) Jp[k] *= sc;
        }
      } else {
        acc += 0.5 * sq;
      }
      if (jac) { out->r[2 * i] = r[0]; out->r[2 * i + 1] = r[1]; }
    }
    partial[t] = acc;
  });
  double cost = 0.0;
  for (double c : partial) cost += c;
  if (s.use_constraints) {
    if (jac) { out->cr.assign(s.M, 0.0); out->cJ.assign((size_t)s.M * 6, 0.0); }
    for (int c = 0; c < s.M; ++c) {
      const double* p1 = pts + 3 * s.p->c_i[c];
      const double* p2 = pts + 3 * s.p->c_j[c];
      if (jac) {
        Jet<6> a[3], b[3];
        for (int k = 0; k < 3; ++k) { a[k] = Jet<6>(p1[k], k); b[k] = Jet<6>(p2[k], 3 + k); }
        Jet<6> r = lo::distance_constraint<Jet<6>>(a, b, s.p->c_dist[c], s.p->c_sigma[c]);
        out->cr[c] = r.a;
        for (int k = 0; k < 6; ++k) out->cJ[(size_t)c * 6 + k] = r.v[k];
        cost += 0.5 * r.a * r.a;
      } else {
        const double r = lo::distance_constraint<double>(p1, p2, s.p->c_dist[c], s.p->c_sigma[c]);
        cost += 0.5 * r * r;
      }
    }
  }
  if (jac) out->cost = cost;
  return cost;
}

// ---------------------------------------------------------------------------------------------
// Column quantities: gradient J^T r, squared column norms, Jacobi scaling
// ---------------------------------------------------------------------------------------------
struct Columns {
  std::vector<double> cam, view, pt;  // 17, 6F, 3P
  void resize(const Structure& s, double v) { cam.assign(NC, v); view.assign(6 * (size_t)s.F, v); pt.assign(3 * (size_t)s.P, v); }
};

void gradient_and_norms(const Structure& s, const Eval& e, Columns* grad, Columns* sqn) {
  grad->resize(s, 0.0); sqn->resize(s, 0.0);
  for (int i = 0; i < s.N; ++i) {
    const int f = s.p->fr[i], p = s.p->pt[i];
    for (int a = 0; a < 2; ++a) {
      const double r = e.r[2 * (size_t)i + a];
      const double* jc = &e.Jc[((size_t)i * 2 + a) * NC];
      for (int k = 0; k < NC; ++k) { grad->cam[k] += jc[k] * r; sqn->cam[k] += jc[k] * jc[k]; }
      if (s.use_poses) {
        const double* jv = &e.Jv[((size_t)i * 2 + a) * 6];
        for (int k = 0; k < 6; ++k) { grad->view[6 * f + k] += jv[k] * r; sqn->view[6 * f + k] += jv[k] * jv[k]; }
      }
      if (s.use_points) {
        const double* jp = &e.Jp[((size_t)i * 2 + a) * 3];
        for (int k = 0; k < 3; ++k) { grad->pt[3 * p + k] += jp[k] * r; sqn->pt[3 * p + k] += jp[k] * jp[k]; }
      }
    }
  }
  if (s.use_constraints)
    for (int c = 0; c < s.M; ++c) {
      const int pi = s.p->c_i[c], pj = s.p->c_j[c];
      for (int k = 0; k < 3; ++k) {
        const double ji = e.cJ[(size_t)c * 6 + k], jj = e.cJ[(size_t)c * 6 + 3 + k];
        grad->pt[3 * pi + k] += ji * e.cr[c]; sqn->pt[3 * pi + k] += ji * ji;
        grad->pt[3 * pj + k] += jj * e.cr[c]; sqn->pt[3 * pj + k] += jj * jj;
      }
    }
  // fixed camera slots are not part of the tangent space (ceres::SubsetManifold)
  for (int k = 0; k < NC; ++k) if (!s.cam_free[k]) { grad->cam[k] = 0.0; sqn->cam[k] = 0.0; }
}

// ---------------------------------------------------------------------------------------------
// Dense Schur elimination (ceres::SchurEliminator + DenseSchurComplementSolver), in the
// Jacobi-scaled space exactly as ceres does it; e-blocks = non-promoted 3D points.
// System: (J^T J + D^2) y = J^T r ; step = -y.
// ---------------------------------------------------------------------------------------------
struct Linear {
  std::vector<double> lhs, rhs;        // reduced system, scaled space, rhs = F^T r - F^T E (E^T E)^-1 E^T r
  std::vector<double> ete_inv;         // 9 per point, scaled space, damped
  std::vector<double> eg;              // 3 per point: E^T r (scaled)
};

struct Spin { std::atomic_flag f = ATOMIC_FLAG_INIT; void lock() { while (f.test_and_set(std::memory_order_acquire)) {} } void unlock() { f.clear(std::memory_order_release); } };

bool invert_spd3(const double m[9], double inv[9]) {
  // LLT then solve identity (ceres InvertPSDMatrix<3> via Eigen LLT)
  double l00 = m[0]; if (!(l00 > 0.0)) return false; l00 = std::sqrt(l00);
  const double l10 = m[3] / l00, l20 = m[6] / l00;
  double l11 = m[4] - l10 * l10; if (!(l11 > 0.0)) return false; l11 = std::sqrt(l11);
  const double l21 = (m[7] - l20 * l10) / l11;
  double l22 = m[8] - l20 * l20 - l21 * l21; if (!(l22 > 0.0)) return false; l22 = std::sqrt(l22);
  for (int c = 0; c < 3; ++c) {
    double b[3] = {c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0};
    const double y0 = b[0] / l00, y1 = (b[1] - l10 * y0) / l11, y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
    const double x2 = y2 / l22, x1 = (y1 - l21 * x2) / l11, x0 = (y0 - l10 * x1 - l20 * x2) / l00;
    inv[0 * 3 + c] = x0; inv[1 * 3 + c] = x1; inv[2 * 3 + c] = x2;
  }
  return true;
}

// sigma: Jacobi scaling per column (1 when disabled); D2: LM diagonal squared per column (scaled space)
bool schur_eliminate(const Structure& s, const Eval& e, const Columns& sigma, const Columns& D2,
                     int threads, Linear* lin) {
  const int n = s.n_red;
  lin->lhs.assign((size_t)n * n, 0.0);
  lin->rhs.assign(n, 0.0);
  lin->ete_inv.assign(9 * (size_t)s.P, 0.0);
  lin->eg.assign(3 * (size_t)s.P, 0.0);
  std::vector<Spin> locks(4096);
  std::atomic<bool> ok(true);
  auto add_block = [&](int r0, int c0, int nr, int nc, const double* blk, int ld, double sign) {
    // lower+upper both kept: store at (r0+i, c0+j) and mirror when off-diagonal block
    Spin& l = locks[((size_t)(r0 / 3) * 1315423911u + (size_t)(c0 / 3)) & 4095];
    l.lock();
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) lin->lhs[(size_t)(r0 + i) * n + c0 + j] += sign * blk[i * ld + j];
    l.unlock();
  };
  Spin rhs_lock;
  parallel_for(threads, s.P, [&](int64_t lo_, int64_t hi_, int) {
    std::vector<double> Frow;        // scaled F row (dense over the chunk's columns)
    std::vector<int> cols;           // chunk-local column -> reduced column
    std::vector<double> buffer, fb;  // E^T F (3 x m), F^T r local
    std::vector<double> ftf;
    std::vector<int> fmap(s.F, -1), pmap;
    for (int64_t pt = lo_; pt < hi_; ++pt) {
      const int nobs = s.pt_begin[pt + 1] - s.pt_begin[pt];
      const auto& cons = s.pt_cons[pt];
      if (nobs == 0 && cons.empty()) continue;
      const bool elim = s.use_points && s.promoted[pt] < 0;
      // chunk-local columns: camera | frames seen | promoted points touched (incl. self if promoted)
      cols.clear();
      for (int k = 0; k < NC; ++k) cols.push_back(k);
      std::vector<int> frames;
      for (int q = 0; q < nobs; ++q) {
        const int f = s.p->fr[s.pt_obs[s.pt_begin[pt] + q]];
        if (fmap[f] < 0) { fmap[f] = (int)cols.size(); frames.push_back(f); for (int k = 0; k < 6; ++k) cols.push_back(s.view_col(f) + k); }
      }
      std::vector<std::pair<int, int>> proms;  // (point id, local col)
      auto prom_local = [&](int id) {
        for (auto& pr : proms) if (pr.first == id) return pr.second;
        const int lc = (int)cols.size(); proms.push_back({id, lc});
        for (int k = 0; k < 3; ++k) cols.push_back(s.prom_col(s.promoted[id]) + k);
        return lc;
      };
      if (s.use_points && !elim) prom_local((int)pt);
      // a constraint is a row of the chunk of its eliminated endpoint; if both endpoints are
      // promoted it has no e-block and is handled with the chunk of c_i.
      std::vector<int> my_cons;
      for (int c : cons) {
        const int pi = s.p->c_i[c], pj = s.p->c_j[c];
        const bool ei = s.promoted[pi] < 0, ej = s.promoted[pj] < 0;
        const int owner = ei ? pi : (ej ? pj : pi);
        if (owner != pt) continue;
        my_cons.push_back(c);
        if (!ei) prom_local(pi);
        if (!ej) prom_local(pj);
      }
      const int m = (int)cols.size();
      buffer.assign(3 * (size_t)m, 0.0); fb.assign(m, 0.0); ftf.assign((size_t)m * m, 0.0);
      double ete[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
      Frow.assign(m, 0.0);
      auto accumulate_row = [&](const double* Erow /*3 scaled or null*/, double r) {
        for (int a = 0; a < m; ++a) {
          const double fa = Frow[a];
          if (fa == 0.0) continue;
          fb[a] += fa * r;
          for (int b = 0; b < m; ++b) ftf[(size_t)a * m + b] += fa * Frow[b];
          if (Erow) for (int k = 0; k < 3; ++k) buffer[(size_t)k * m + a] += Erow[k] * fa;
        }
        if (Erow) for (int k = 0; k < 3; ++k) { g[k] += Erow[k] * r; for (int l = 0; l < 3; ++l) ete[k * 3 + l] += Erow[k] * Erow[l]; }
      };
      for (int q = 0; q < nobs; ++q) {
        const int i = s.pt_obs[s.pt_begin[pt] + q];
        const int f = s.p->fr[i];
        for (int a = 0; a < 2; ++a) {
          std::fill(Frow.begin(), Frow.end(), 0.0);
          const double* jc = &e.Jc[((size_t)i * 2 + a) * NC];
          for (int k = 0; k < NC; ++k) Frow[k] = s.cam_free[k] ? jc[k] * sigma.cam[k] : 0.0;
          if (s.use_poses) {
            const double* jv = &e.Jv[((size_t)i * 2 + a) * 6];
            for (int k = 0; k < 6; ++k) Frow[fmap[f] + k] = jv[k] * sigma.view[6 * f + k];
          }
          double Er[3];
          if (s.use_points) {
            const double* jp = &e.Jp[((size_t)i * 2 + a) * 3];
            for (int k = 0; k < 3; ++k) Er[k] = jp[k] * sigma.pt[3 * pt + k];
            if (!elim) { const int lc = prom_local((int)pt); for (int k = 0; k < 3; ++k) Frow[lc + k] = Er[k]; }
          }
          accumulate_row(elim ? Er : nullptr, e.r[2 * (size_t)i + a]);
        }
      }
      for (int c : my_cons) {
        const int pi = s.p->c_i[c], pj = s.p->c_j[c];
        std::fill(Frow.begin(), Frow.end(), 0.0);
        double Er[3]; bool has_e = false;
        for (int side = 0; side < 2; ++side) {
          const int id = side ? pj : pi;
          const double* j3 = &e.cJ[(size_t)c * 6 + 3 * side];
          if (id == pt && elim) { for (int k = 0; k < 3; ++k) Er[k] = j3[k] * sigma.pt[3 * id + k]; has_e = true; }
          else { const int lc = prom_local(id); for (int k = 0; k < 3; ++k) Frow[lc + k] = j3[k] * sigma.pt[3 * id + k]; }
        }
        accumulate_row(has_e ? Er : nullptr, e.cr[c]);
      }
      if (elim) {
        for (int k = 0; k < 3; ++k) ete[k * 3 + k] += D2.pt[3 * pt + k];
        double inv[9];
        if (!invert_spd3(ete, inv)) { ok = false; for (int k = 0; k < 9; ++k) inv[k] = 0.0; }
        for (int k = 0; k < 9; ++k) lin->ete_inv[9 * pt + k] = inv[k];
        for (int k = 0; k < 3; ++k) lin->eg[3 * pt + k] = g[k];
        // lhs -= buffer^T inv buffer ; rhs -= buffer^T inv g
        double ig[3];
        for (int k = 0; k < 3; ++k) ig[k] = inv[k * 3] * g[0] + inv[k * 3 + 1] * g[1] + inv[k * 3 + 2] * g[2];
        for (int a = 0; a < m; ++a) {
          double ib[3];
          for (int k = 0; k < 3; ++k) ib[k] = inv[k * 3] * buffer[a] + inv[k * 3 + 1] * buffer[m + a] + inv[k * 3 + 2] * buffer[2 * (size_t)m + a];
          fb[a] -= buffer[a] * ig[0] + buffer[m + a] * ig[1] + buffer[2 * (size_t)m + a] * ig[2];
          for (int b = 0; b < m; ++b)
            ftf[(size_t)b * m + a] -= buffer[b] * ib[0] + buffer[m + b] * ib[1] + buffer[2 * (size_t)m + b] * ib[2];
        }
      }
      // scatter the chunk's dense m x m contribution block-wise into lhs
      std::vector<std::pair<int, int>> blocks;  // (local start, width)
      blocks.push_back({0, NC});
      for (int f : frames) blocks.push_back({fmap[f], 6});
      for (auto& pr : proms) blocks.push_back({pr.second, 3});
      for (auto& br : blocks)
        for (auto& bc : blocks)
          add_block(cols[br.first], cols[bc.first], br.second, bc.second, &ftf[(size_t)br.first * m + bc.first], m, 1.0);
      rhs_lock.lock();
      for (int a = 0; a < m; ++a) lin->rhs[cols[a]] += fb[a];
      rhs_lock.unlock();
      for (int f : frames) fmap[f] = -1;
    }
  });
  // D^2 on the f-blocks; identity on columns that are not part of the problem
  for (int k = 0; k < NC; ++k) {
    if (s.cam_free[k]) lin->lhs[(size_t)k * n + k] += D2.cam[k];
    else { for (int j = 0; j < n; ++j) { lin->lhs[(size_t)k * n + j] = 0.0; lin->lhs[(size_t)j * n + k] = 0.0; } lin->lhs[(size_t)k * n + k] = 1.0; lin->rhs[k] = 0.0; }
  }
  for (int f = 0; f < s.F; ++f)
    for (int k = 0; k < 6; ++k) {
      const int c = s.view_col(f) + k;
      if (s.use_poses && s.fr_used[f]) lin->lhs[(size_t)c * n + c] += D2.view[6 * f + k];
      else { lin->lhs[(size_t)c * n + c] = 1.0; lin->rhs[c] = 0.0; }
    }
  for (int q = 0; q < s.n_prom; ++q)
    for (int k = 0; k < 3; ++k) { const int c = s.prom_col(q) + k; lin->lhs[(size_t)c * n + c] += D2.pt[3 * s.promoted_ids[q] + k]; }
  return ok;
}

// dense Cholesky solve (Eigen LLT in ceres DenseSchurComplementSolver); A is overwritten
bool cholesky_solve(std::vector<double>& A, int n, std::vector<double>& b) {
  for (int j = 0; j < n; ++j) {
    double* rj = &A[(size_t)j * n];
    double d = rj[j];
    for (int k = 0; k < j; ++k) d -= rj[k] * rj[k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d); rj[j] = d;
    const double di = 1.0 / d;
    for (int i = j + 1; i < n; ++i) {
      double* ri = &A[(size_t)i * n];
      double sum = ri[j];
      for (int k = 0; k < j; ++k) sum -= ri[k] * rj[k];
      ri[j] = sum * di;
    }
  }
  for (int i = 0; i < n; ++i) { double sum = b[i]; const double* ri = &A[(size_t)i * n]; for (int k = 0; k < i; ++k) sum -= ri[k] * b[k]; b[i] = sum / ri[i]; }
  for (int i = n - 1; i >= 0; --i) { double sum = b[i]; for (int k = i + 1; k < n; ++k) sum -= A[(size_t)k * n + i] * b[k]; b[i] = sum / A[(size_t)i * n + i]; }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Armijo line search used by ceres for bound-constrained trust-region problems
// (TrustRegionMinimizer::DoLineSearch; ArmijoLineSearch; polynomial interpolation, CUBIC)
// ---------------------------------------------------------------------------------------------
struct Sample { double x = 0, value = 0, gradient = 0; bool value_valid = false, gradient_valid = false; };

double polyval(const std::vector<double>& c, double x) { double v = 0; for (double a : c) v = v * x + a; return v; }

std::vector<double> interpolating_polynomial(const std::vector<Sample>& samples) {
  int m = 0; for (auto& s : samples) m += (s.value_valid ? 1 : 0) + (s.gradient_valid ? 1 : 0);
  const int deg = m - 1;
  std::vector<double> A((size_t)m * m, 0.0), b(m, 0.0);
  int row = 0;
  for (auto& s : samples) {
    if (s.value_valid) { for (int j = 0; j <= deg; ++j) A[(size_t)row * m + j] = std::pow(s.x, deg - j); b[row++] = s.value; }
    if (s.gradient_valid) { for (int j = 0; j < deg; ++j) A[(size_t)row * m + j] = (deg - j) * std::pow(s.x, deg - j - 1); b[row++] = s.gradient; }
  }
  // Gaussian elimination with full pivoting
  std::vector<int> perm(m); for (int i = 0; i < m; ++i) perm[i] = i;
  for (int k = 0; k < m; ++k) {
    int pr = k, pc = k; double best = 0;
    for (int i = k; i < m; ++i) for (int j = k; j < m; ++j) if (std::fabs(A[(size_t)i * m + j]) > best) { best = std::fabs(A[(size_t)i * m + j]); pr = i; pc = j; }
    if (best == 0) break;
    for (int j = 0; j < m; ++j) std::swap(A[(size_t)k * m + j], A[(size_t)pr * m + j]);
    std::swap(b[k], b[pr]);
    for (int i = 0; i < m; ++i) std::swap(A[(size_t)i * m + k], A[(size_t)i * m + pc]);
    std::swap(perm[k], perm[pc]);
    for (int i = k + 1; i < m; ++i) { const double f = A[(size_t)i * m + k] / A[(size_t)k * m + k]; for (int j = k; j < m; ++j) A[(size_t)i * m + j] -= f * A[(size_t)k * m + j]; b[i] -= f * b[k]; }
  }
  std::vector<double> y(m, 0.0), c(m, 0.0);
  for (int i = m - 1; i >= 0; --i) { double sum = b[i]; for (int j = i + 1; j < m; ++j) sum -= A[(size_t)i * m + j] * y[j]; y[i] = A[(size_t)i * m + i] != 0 ? sum / A[(size_t)i * m + i] : 0.0; }
  for (int i = 0; i < m; ++i) c[perm[i]] = y[i];
  return c;
}

void polynomial_root_real_parts(std::vector<double> c, std::vector<double>* roots) {
  roots->clear();
  while (!c.empty() && c.front() == 0.0) c.erase(c.begin());
  const int d = (int)c.size() - 1;
  if (d < 1) return;
  if (d == 1) { roots->push_back(-c[1] / c[0]); return; }
  if (d == 2) {
    const double D = c[1] * c[1] - 4 * c[0] * c[2];
    if (D >= 0) { const double sq = std::sqrt(D); const double q = -0.5 * (c[1] + (c[1] >= 0 ? sq : -sq)); roots->push_back(q / c[0]); if (q != 0) roots->push_back(c[2] / q); }
    else { roots->push_back(-c[1] / (2 * c[0])); roots->push_back(-c[1] / (2 * c[0])); }
    return;
  }
  // Durand–Kerner on the monic polynomial
  std::vector<std::complex<double>> z(d), a(d + 1);
  for (int i = 0; i <= d; ++i) a[i] = c[i] / c[0];
  double rad = 0; for (int i = 1; i <= d; ++i) rad = std::max(rad, std::abs(a[i])); rad = 1 + rad;
  for (int i = 0; i < d; ++i) z[i] = std::polar(rad * 0.7, 2 * M_PI * i / d + 0.4);
  for (int it = 0; it < 500; ++it) {
    double change = 0;
    for (int i = 0; i < d; ++i) {
      std::complex<double> pv = a[0]; for (int k = 1; k <= d; ++k) pv = pv * z[i] + a[k];
      std::complex<double> den = 1; for (int j = 0; j < d; ++j) if (j != i) den *= (z[i] - z[j]);
      if (std::abs(den) == 0) continue;
      const std::complex<double> dz = pv / den; z[i] -= dz; change = std::max(change, std::abs(dz));
    }
    if (change < 1e-15 * rad) break;
  }
  for (auto& r : z) roots->push_back(r.real());
}

double minimize_interpolating_polynomial(const std::vector<Sample>& samples, double x_min, double x_max) {
  const std::vector<double> poly = interpolating_polynomial(samples);
  double best_x = 0.5 * (x_min + x_max), best_v = polyval(poly, best_x);
  const double vmin = polyval(poly, x_min); if (vmin < best_v) { best_v = vmin; best_x = x_min; }
  const double vmax = polyval(poly, x_max); if (vmax < best_v) { best_v = vmax; best_x = x_max; }
  if (poly.size() <= 2) return best_x;
  std::vector<double> der; const int deg = (int)poly.size() - 1;
  for (int i = 0; i < deg; ++i) der.push_back(poly[i] * (deg - i));
  std::vector<double> roots; polynomial_root_real_parts(der, &roots);
  for (double r : roots) { if (r < x_min || r > x_max) continue; const double v = polyval(poly, r); if (v < best_v) { best_v = v; best_x = r; } }
  return best_x;
}

// ---------------------------------------------------------------------------------------------
// Solver state shared by sweep and solve
// ---------------------------------------------------------------------------------------------
struct Params {
  std::vector<double> cam, views, pts;
  void from(const lifcal_ba_problem* p) { cam.assign(p->cam, p->cam + NC); views.assign(p->views, p->views + 6 * (size_t)p->n_frames); pts.assign(p->pts, p->pts + 3 * (size_t)p->n_points); }
  void to(const lifcal_ba_problem* p) const { std::copy(cam.begin(), cam.end(), p->cam); std::copy(views.begin(), views.end(), p->views); std::copy(pts.begin(), pts.end(), p->pts); }
};

// ParameterBlock::Plus: x + delta then projection onto the box (ceres parameter_block.h)
void plus(const Structure& s, const Params& x, const Columns& delta, Params* out) {
  *out = x;
  for (int k = 0; k < NC; ++k) {
    if (s.cam_free[k]) out->cam[k] = x.cam[k] + delta.cam[k];
    if (s.p->lower && out->cam[k] < s.p->lower[k]) out->cam[k] = s.p->lower[k];
    if (s.p->upper && out->cam[k] > s.p->upper[k]) out->cam[k] = s.p->upper[k];
  }
  if (s.use_poses) for (size_t k = 0; k < x.views.size(); ++k) out->views[k] = x.views[k] + delta.view[k];
  if (s.use_points) for (size_t k = 0; k < x.pts.size(); ++k) out->pts[k] = x.pts[k] + delta.pt[k];
}

bool is_constrained(const Structure& s) {
  if (!s.p->lower && !s.p->upper) return false;
  for (int k = 0; k < NC; ++k) {
    if (s.p->lower && s.p->lower[k] > -std::numeric_limits<double>::max()) return true;
    if (s.p->upper && s.p->upper[k] < std::numeric_limits<double>::max()) return true;
  }
  return false;
}

// norm over the parameter blocks that are part of the ceres program
double ambient_norm2(const Structure& s, const Params& a, const Params* b) {
  double n2 = 0;
  for (int k = 0; k < NC; ++k) { const double d = a.cam[k] - (b ? b->cam[k] : 0.0); n2 += d * d; }
  if (s.use_poses) for (int f = 0; f < s.F; ++f) if (s.fr_used[f]) for (int k = 0; k < 6; ++k) { const double d = a.views[6 * f + k] - (b ? b->views[6 * f + k] : 0.0); n2 += d * d; }
  if (s.use_points) for (int p = 0; p < s.P; ++p) if (s.pt_used[p]) for (int k = 0; k < 3; ++k) { const double d = a.pts[3 * p + k] - (b ? b->pts[3 * p + k] : 0.0); n2 += d * d; }
  return n2;
}

double columns_dot(const Structure& s, const Columns& a, const Columns& b) {
  double d = 0;
  for (int k = 0; k < NC; ++k) d += a.cam[k] * b.cam[k];
  if (s.use_poses) for (size_t k = 0; k < a.view.size(); ++k) d += a.view[k] * b.view[k];
  if (s.use_points) for (size_t k = 0; k < a.pt.size(); ++k) d += a.pt[k] * b.pt[k];
  return d;
}

double columns_max_abs(const Structure& s, const Columns& a) {
  double m = 0;
  for (int k = 0; k < NC; ++k) m = std::max(m, std::fabs(a.cam[k]));
  if (s.use_poses) for (double v : a.view) m = std::max(m, std::fabs(v));
  if (s.use_points) for (double v : a.pt) m = std::max(m, std::fabs(v));
  return m;
}

void jacobi_scaling(const Structure& s, const Columns& sqn, bool enabled, Columns* sigma) {
  sigma->resize(s, 1.0);
  if (!enabled) return;
  for (int k = 0; k < NC; ++k) sigma->cam[k] = 1.0 / (1.0 + std::sqrt(sqn.cam[k]));
  for (size_t k = 0; k < sqn.view.size(); ++k) sigma->view[k] = 1.0 / (1.0 + std::sqrt(sqn.view[k]));
  for (size_t k = 0; k < sqn.pt.size(); ++k) sigma->pt[k] = 1.0 / (1.0 + std::sqrt(sqn.pt[k]));
}

// LevenbergMarquardtStrategy::ComputeStep: D^2 = clamp(diag((J sigma)^T (J sigma)), min, max) / radius
void lm_diagonal(const Columns& sqn, const Columns& sigma, double radius, double dmin, double dmax, Columns* D2) {
  auto f = [&](double h, double sg) { return std::min(std::max(h * sg * sg, dmin), dmax) / radius; };
  D2->cam.resize(sqn.cam.size()); D2->view.resize(sqn.view.size()); D2->pt.resize(sqn.pt.size());
  for (size_t k = 0; k < sqn.cam.size(); ++k) D2->cam[k] = f(sqn.cam[k], sigma.cam[k]);
  for (size_t k = 0; k < sqn.view.size(); ++k) D2->view[k] = f(sqn.view[k], sigma.view[k]);
  for (size_t k = 0; k < sqn.pt.size(); ++k) D2->pt[k] = f(sqn.pt[k], sigma.pt[k]);
}

// solve the damped system; returns the UNSCALED step delta (= -sigma * y) and the model cost change
bool compute_step(const Structure& s, const Eval& e, const Columns& sigma, const Columns& D2,
                  int threads, Linear* lin, Columns* delta, double* model_cost_change, bool keep_lhs) {
  if (!schur_eliminate(s, e, sigma, D2, threads, lin)) return false;
  std::vector<double> A = lin->lhs, y = lin->rhs;
  if (!cholesky_solve(A, s.n_red, y)) return false;
  (void)keep_lhs;
  delta->resize(s, 0.0);
  Columns ys; ys.resize(s, 0.0);  // scaled-space solution of (J^T J + D^2) y = J^T r
  for (int k = 0; k < NC; ++k) ys.cam[k] = s.cam_free[k] ? y[k] : 0.0;
  if (s.use_poses) for (int f = 0; f < s.F; ++f) for (int k = 0; k < 6; ++k) ys.view[6 * f + k] = s.fr_used[f] ? y[s.view_col(f) + k] : 0.0;
  if (s.use_points) {
    for (int q = 0; q < s.n_prom; ++q) for (int k = 0; k < 3; ++k) ys.pt[3 * s.promoted_ids[q] + k] = y[s.prom_col(q) + k];
    // back substitution: y_e = (E^T E)^-1 (E^T r - E^T F y_f)
    for (int pt = 0; pt < s.P; ++pt) {
      if (s.promoted[pt] >= 0 || !s.pt_used[pt]) continue;
      double t[3] = {lin->eg[3 * pt], lin->eg[3 * pt + 1], lin->eg[3 * pt + 2]};
      auto sub_row = [&](const double Er[3], double fy) { for (int k = 0; k < 3; ++k) t[k] -= Er[k] * fy; };
      for (int q = s.pt_begin[pt]; q < s.pt_begin[pt + 1]; ++q) {
        const int i = s.pt_obs[q], f = s.p->fr[i];
        for (int a = 0; a < 2; ++a) {
          double fy = 0;
          const double* jc = &e.Jc[((size_t)i * 2 + a) * NC];
          for (int k = 0; k < NC; ++k) if (s.cam_free[k]) fy += jc[k] * sigma.cam[k] * ys.cam[k];
          const double* jv = &e.Jv[((size_t)i * 2 + a) * 6];
          for (int k = 0; k < 6; ++k) fy += jv[k] * sigma.view[6 * f + k] * ys.view[6 * f + k];
          const double* jp = &e.Jp[((size_t)i * 2 + a) * 3];
          double Er[3]; for (int k = 0; k < 3; ++k) Er[k] = jp[k] * sigma.pt[3 * pt + k];
          sub_row(Er, fy);
        }
      }
      for (int c : s.pt_cons[pt]) {
        const int pi = s.p->c_i[c], pj = s.p->c_j[c];
        const int other = (pi == pt) ? pj : pi;
        const double* je = &e.cJ[(size_t)c * 6 + (pi == pt ? 0 : 3)];
        const double* jo = &e.cJ[(size_t)c * 6 + (pi == pt ? 3 : 0)];
        double fy = 0; for (int k = 0; k < 3; ++k) fy += jo[k] * sigma.pt[3 * other + k] * ys.pt[3 * other + k];
        double Er[3]; for (int k = 0; k < 3; ++k) Er[k] = je[k] * sigma.pt[3 * pt + k];
        sub_row(Er, fy);
      }
      const double* inv = &lin->ete_inv[9 * pt];
      for (int k = 0; k < 3; ++k) ys.pt[3 * pt + k] = inv[k * 3] * t[0] + inv[k * 3 + 1] * t[1] + inv[k * 3 + 2] * t[2];
    }
  }
  // trust_region_step = -y ; model_cost_change = -(J s)^T (r + J s / 2) evaluated row by row
  double mcc = 0.0;
  auto row_model = [&](double js, double r) { mcc += -js * (r + 0.5 * js); };
  for (int i = 0; i < s.N; ++i) {
    const int f = s.p->fr[i], pt = s.p->pt[i];
    for (int a = 0; a < 2; ++a) {
      double js = 0;
      const double* jc = &e.Jc[((size_t)i * 2 + a) * NC];
      for (int k = 0; k < NC; ++k) if (s.cam_free[k]) js -= jc[k] * sigma.cam[k] * ys.cam[k];
      if (s.use_poses) { const double* jv = &e.Jv[((size_t)i * 2 + a) * 6]; for (int k = 0; k < 6; ++k) js -= jv[k] * sigma.view[6 * f + k] * ys.view[6 * f + k]; }
      if (s.use_points) { const double* jp = &e.Jp[((size_t)i * 2 + a) * 3]; for (int k = 0; k < 3; ++k) js -= jp[k] * sigma.pt[3 * pt + k] * ys.pt[3 * pt + k]; }
      row_model(js, e.r[2 * (size_t)i + a]);
    }
  }
  if (s.use_constraints)
    for (int c = 0; c < s.M; ++c) {
      const int pi = s.p->c_i[c], pj = s.p->c_j[c];
      double js = 0;
      for (int k = 0; k < 3; ++k) js -= e.cJ[(size_t)c * 6 + k] * sigma.pt[3 * pi + k] * ys.pt[3 * pi + k] + e.cJ[(size_t)c * 6 + 3 + k] * sigma.pt[3 * pj + k] * ys.pt[3 * pj + k];
      row_model(js, e.cr[c]);
    }
  *model_cost_change = mcc;
  for (int k = 0; k < NC; ++k) delta->cam[k] = -ys.cam[k] * sigma.cam[k];
  for (size_t k = 0; k < ys.view.size(); ++k) delta->view[k] = -ys.view[k] * sigma.view[k];
  for (size_t k = 0; k < ys.pt.size(); ++k) delta->pt[k] = -ys.pt[k] * sigma.pt[k];
  return true;
}

}  // namespace

// =============================================================================================
// C API
// =============================================================================================
extern "C" {

int lo_project_point(const double pc[3], double spx, double spy, double fL, double bL0, double B,
                     const double c_raw[2], const double ml[2], const double* radial, int n_radial,
                     const double* tangential, int ml_center_adj, double out[2]) {
  lo::project_point<double>(out[0], out[1], pc, spx, spy, fL, bL0, B, c_raw, ml, radial, n_radial, tangential, ml_center_adj != 0);
  return 0;
}

int lo_rigid_transform(const double view[6], double RT[12]) {
  double m[3][4]; lo::rigid_transform<double>(view, view + 3, m);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) RT[i * 4 + j] = m[i][j];
  return 0;
}

// One residual block with its autodiff Jacobian.  `view`/`point` select the arity like
// OurCostFunctionBundle::Create: arity 3 -> <2,17,6,3>, 2 -> <2,17,6> (point constant),
// 1 -> <2,17> (view and point constant).  J layout: [2][26] = camera 17 | view 6 | point 3.
int lo_residual_block(uint32_t config, int arity, const double cam[17], const double view[6], const double point[3],
                      double u, double v, double mcx, double mcy, double spx, double spy, double scale,
                      double r[2], double J[52]) {
  ObsFunctor f(config, u, v, spx, spy, scale, mcx, mcy);
  if (arity < 3) f.set_fixed_point(point);
  if (arity < 2) f.set_fixed_view(view);
  double Jc[2 * NC] = {0}, Jv[12] = {0}, Jp[6] = {0};
  if (J) {
    if (arity == 3) eval_jet<26>(f, cam, view, point, r, Jc, Jv, Jp);
    else if (arity == 2) eval_jet<23>(f, cam, view, nullptr, r, Jc, Jv, Jp);
    else eval_jet<17>(f, cam, nullptr, nullptr, r, Jc, Jv, Jp);
    for (int a = 0; a < 2; ++a) {
      for (int k = 0; k < NC; ++k) J[a * 26 + k] = Jc[a * NC + k];
      for (int k = 0; k < 6; ++k) J[a * 26 + NC + k] = Jv[a * 6 + k];
      for (int k = 0; k < 3; ++k) J[a * 26 + NC + 6 + k] = Jp[a * 3 + k];
    }
  } else {
    f(cam, arity >= 2 ? view : (const double*)nullptr, arity == 3 ? point : (const double*)nullptr, r);
  }
  return 0;
}

int lo_constraint_block(const double p1[3], const double p2[3], double distance, double sigma, double* r, double J[6]) {
  Jet<6> a[3], b[3];
  for (int k = 0; k < 3; ++k) { a[k] = Jet<6>(p1[k], k); b[k] = Jet<6>(p2[k], 3 + k); }
  Jet<6> res = lo::distance_constraint<Jet<6>>(a, b, distance, sigma);
  *r = res.a;
  if (J) for (int k = 0; k < 6; ++k) J[k] = res.v[k];
  return 0;
}

int lo_set_fixed_frames(const uint8_t* fixed, uint32_t n_frames) {
  g_fixed_frames.clear();
  if (fixed) g_fixed_frames.assign(fixed, fixed + n_frames);
  return 0;
}

int lo_cost(const lifcal_ba_problem* p, double loss_scale, int threads, double* cost) {
  if (int rc = validate(p)) return rc;
  Structure s(p);
  *cost = evaluate(s, p->cam, p->views, p->pts, loss_scale, threads, nullptr);
  return 0;
}

// residuals (uncorrected, input order) at the current point
int lo_residuals(const lifcal_ba_problem* p, double* r2n) {
  if (int rc = validate(p)) return rc;
  Structure s(p);
  for (int i = 0; i < s.N; ++i) eval_obs(s, i, p->cam, p->views, p->pts, false, r2n + 2 * (size_t)i, nullptr, nullptr, nullptr);
  return 0;
}

int lo_reduced_size(const lifcal_ba_problem* p, uint32_t* n_reduced, uint32_t* n_promoted) {
  if (int rc = validate(p)) return rc;
  Structure s(p);
  *n_reduced = s.n_red; *n_promoted = s.n_prom;
  return 0;
}

// One Jacobian + Schur sweep at `radius` with the Jacobi scaling of the current point
// (= what ceres does in iteration 0/1).  Outputs follow include/lifcal_ba.h lifcal_ba_sweep_out:
// UNSCALED space, S delta = rhs.
static int sweep_impl(const lifcal_ba_problem* p, const lifcal_ba_options* o, double radius, int threads,
                      lifcal_ba_sweep_out* out, double* seconds_eval, double* seconds_schur, bool analytic) {
  if (int rc = validate(p)) return rc;
  Structure s(p);
  if (analytic) s.build_lenses();   // problem set-up like the CSR above, outside the timed sweep
  Eval e;
  const double t0 = now_s();
  evaluate(s, p->cam, p->views, p->pts, o->loss_scale, threads, &e, analytic);
  const double t1 = now_s();
  Columns grad, sqn, sigma, D2;
  gradient_and_norms(s, e, &grad, &sqn);
  jacobi_scaling(s, sqn, o->jacobi_scaling != 0, &sigma);
  lm_diagonal(sqn, sigma, radius, o->min_lm_diagonal, o->max_lm_diagonal, &D2);
  Linear lin;
  const bool ok = schur_eliminate(s, e, sigma, D2, threads, &lin);
  const double t2 = now_s();
  if (seconds_eval) *seconds_eval = t1 - t0;
  if (seconds_schur) *seconds_schur = t2 - t1;
  out->cost = e.cost;
  out->gradient_max_norm = columns_max_abs(s, grad);
  out->n_reduced = s.n_red; out->n_promoted = s.n_prom;
  out->seconds = t2 - t0;
  const int n = s.n_red;
  std::vector<double> sg(n, 1.0);
  for (int k = 0; k < NC; ++k) sg[k] = s.cam_free[k] ? sigma.cam[k] : 1.0;
  for (int f = 0; f < s.F; ++f) for (int k = 0; k < 6; ++k) sg[s.view_col(f) + k] = (s.use_poses && s.fr_used[f]) ? sigma.view[6 * f + k] : 1.0;
  for (int q = 0; q < s.n_prom; ++q) for (int k = 0; k < 3; ++k) sg[s.prom_col(q) + k] = sigma.pt[3 * s.promoted_ids[q] + k];
  if (out->S) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) out->S[(size_t)i * n + j] = lin.lhs[(size_t)i * n + j] / (sg[i] * sg[j]);
  if (out->rhs) for (int i = 0; i < n; ++i) out->rhs[i] = -lin.rhs[i] / sg[i];
  if (out->gradient_reduced) {
    for (int i = 0; i < n; ++i) out->gradient_reduced[i] = 0.0;
    for (int k = 0; k < NC; ++k) out->gradient_reduced[k] = grad.cam[k];
    if (s.use_poses) for (int f = 0; f < s.F; ++f) for (int k = 0; k < 6; ++k) out->gradient_reduced[s.view_col(f) + k] = grad.view[6 * f + k];
    for (int q = 0; q < s.n_prom; ++q) for (int k = 0; k < 3; ++k) out->gradient_reduced[s.prom_col(q) + k] = grad.pt[3 * s.promoted_ids[q] + k];
  }
  if (out->point_gradient) for (size_t k = 0; k < 3 * (size_t)s.P; ++k) out->point_gradient[k] = s.use_points ? grad.pt[k] : 0.0;
  if (out->point_hessian_inv)
    for (int pt = 0; pt < s.P; ++pt)
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
        out->point_hessian_inv[9 * (size_t)pt + 3 * a + b] = (s.use_points && s.promoted[pt] < 0) ? lin.ete_inv[9 * (size_t)pt + 3 * a + b] * sigma.pt[3 * pt + a] * sigma.pt[3 * pt + b] : 0.0;
  return ok ? 0 : LIFCAL_BA_ERR_NUMERIC;
}

int lo_sweep(const lifcal_ba_problem* p, const lifcal_ba_options* o, double radius, int threads,
             lifcal_ba_sweep_out* out, double* seconds_eval, double* seconds_schur) {
  return sweep_impl(p, o, radius, threads, out, seconds_eval, seconds_schur, false);
}
// the same sweep with the analytic Jacobian arm (oracle/analytic.hpp) instead of dual numbers
int lo_sweep_analytic(const lifcal_ba_problem* p, const lifcal_ba_options* o, double radius, int threads,
                      lifcal_ba_sweep_out* out, double* seconds_eval, double* seconds_schur) {
  return sweep_impl(p, o, radius, threads, out, seconds_eval, seconds_schur, true);
}

// ceres::Solve (TrustRegionMinimizer, LEVENBERG_MARQUARDT, DENSE_SCHUR); parameters updated in place
int lo_solve(const lifcal_ba_problem* p, const lifcal_ba_options* o, int threads, lifcal_ba_summary* sum) {
  if (int rc = validate(p)) return rc;
  const double t_start = now_s();
  Structure s(p);
  Params x, cand; x.from(p);
  const bool constrained = is_constrained(s);
  Columns zero; zero.resize(s, 0.0);
  if (constrained) { plus(s, x, zero, &cand); x = cand; }  // IterationZero: project onto the feasible set
  double x_norm = std::sqrt(ambient_norm2(s, x, nullptr));
  Eval e; Columns grad, sqn, sigma, D2, delta;
  double x_cost = evaluate(s, x.cam.data(), x.views.data(), x.pts.data(), o->loss_scale, threads, &e);
  if (!std::isfinite(x_cost)) return LIFCAL_BA_ERR_NUMERIC;
  gradient_and_norms(s, e, &grad, &sqn);
  jacobi_scaling(s, sqn, o->jacobi_scaling != 0, &sigma);  // fixed at iteration 0
  auto gradient_max_norm = [&]() {
    if (!constrained) return columns_max_abs(s, grad);
    Columns ng = grad; for (auto& v : ng.cam) v = -v; for (auto& v : ng.view) v = -v; for (auto& v : ng.pt) v = -v;
    Params proj; plus(s, x, ng, &proj);
    double m = 0;
    for (int k = 0; k < NC; ++k) m = std::max(m, std::fabs(x.cam[k] - proj.cam[k]));
    if (s.use_poses) for (size_t k = 0; k < x.views.size(); ++k) m = std::max(m, std::fabs(x.views[k] - proj.views[k]));
    if (s.use_points) for (size_t k = 0; k < x.pts.size(); ++k) m = std::max(m, std::fabs(x.pts[k] - proj.pts[k]));
    return m;
  };
  double gmax = gradient_max_norm();
  sum->initial_cost = x_cost; sum->iterations = 0; sum->successful_steps = 0; sum->unsuccessful_steps = 0;
  sum->termination = LIFCAL_BA_TERM_NONE; sum->seconds_sweep = 0; sum->seconds_linear_solve = 0;
  double radius = o->initial_radius, decrease_factor = 2.0;
  int invalid_steps = 0, iteration = 0;
  bool step_successful = true;  // iteration 0 counts as successful for the gradient check
  if (o->verbose) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e    0.00e+00 %10.2e   0.00e+00   0.00e+00 %9.2e\n", 0, x_cost, gmax, radius);
  if (gmax <= o->gradient_tolerance) sum->termination = LIFCAL_BA_TERM_GRADIENT_TOLERANCE;
  Linear lin;
  while (sum->termination == LIFCAL_BA_TERM_NONE) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iteration >= o->max_iterations) { sum->termination = LIFCAL_BA_TERM_MAX_ITERATIONS; break; }
    if (step_successful && gmax <= o->gradient_tolerance) { sum->termination = LIFCAL_BA_TERM_GRADIENT_TOLERANCE; break; }
    if (radius < o->min_radius) { sum->termination = LIFCAL_BA_TERM_MIN_RADIUS; break; }
    ++iteration;
    const double ts = now_s();
    lm_diagonal(sqn, sigma, radius, o->min_lm_diagonal, o->max_lm_diagonal, &D2);
    double model_cost_change = 0.0;
    bool valid = compute_step(s, e, sigma, D2, threads, &lin, &delta, &model_cost_change, false);
    sum->seconds_linear_solve += now_s() - ts;
    if (valid) valid = model_cost_change > 0.0;
    if (!valid) {  // HandleInvalidStep
      if (++invalid_steps >= 5) { sum->termination = LIFCAL_BA_TERM_INVALID_STEPS; break; }
      radius *= 0.5; step_successful = false; ++sum->unsuccessful_steps;
      continue;
    }
    invalid_steps = 0;
    if (constrained) {  // DoLineSearch (Armijo, cubic interpolation, is_silent)
      const double g0 = columns_dot(s, grad, delta);
      auto phi = [&](double t, Sample* smp) {
        Columns d = delta; for (auto& v : d.cam) v *= t; for (auto& v : d.view) v *= t; for (auto& v : d.pt) v *= t;
        Params xt; plus(s, x, d, &xt);
        Eval et; const double c = evaluate(s, xt.cam.data(), xt.views.data(), xt.pts.data(), o->loss_scale, threads, &et);
        Columns gt, st; gradient_and_norms(s, et, &gt, &st);
        smp->x = t; smp->value = c; smp->value_valid = std::isfinite(c);
        smp->gradient = columns_dot(s, gt, delta); smp->gradient_valid = smp->value_valid && std::isfinite(smp->gradient);
      };
      Sample init; init.x = 0; init.value = x_cost; init.gradient = g0; init.value_valid = init.gradient_valid = true;
      Sample prev, cur; phi(1.0, &cur);
      const double dir_max = columns_max_abs(s, delta);
      int ls_iter = 0; bool ls_ok = true;
      while (!cur.value_valid || cur.value > x_cost + 1e-4 * g0 * cur.x) {
        if (++ls_iter >= 20) { ls_ok = false; break; }
        double step;
        const double lo_b = 1e-3 * cur.x, hi_b = 0.6 * cur.x;
        if (!cur.value_valid) step = std::min(std::max(cur.x * 0.5, lo_b), hi_b);
        else { std::vector<Sample> smp{init, cur}; if (prev.value_valid) smp.push_back(prev); step = minimize_interpolating_polynomial(smp, lo_b, hi_b); }
        if (step * dir_max < 1e-9) { ls_ok = false; break; }
        prev = cur; phi(step, &cur);
      }
      if (ls_iter > 0 && getenv("LO_DEBUG_LS")) fprintf(stderr, "[oracle] line search: %d backtracks, t = %.6g ok=%d\n", ls_iter, cur.x, (int)ls_ok);
      if (ls_ok) { for (auto& v : delta.cam) v *= cur.x; for (auto& v : delta.view) v *= cur.x; for (auto& v : delta.pt) v *= cur.x; }
    }
    plus(s, x, delta, &cand);
    const double te = now_s();
    double cand_cost = evaluate(s, cand.cam.data(), cand.views.data(), cand.pts.data(), o->loss_scale, threads, nullptr);
    sum->seconds_sweep += now_s() - te;
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    const double step_norm = std::sqrt(ambient_norm2(s, x, &cand));
    if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { sum->termination = LIFCAL_BA_TERM_PARAMETER_TOLERANCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= o->function_tolerance * x_cost) { sum->termination = LIFCAL_BA_TERM_FUNCTION_TOLERANCE; break; }
    const double rel = (cand_cost >= std::numeric_limits<double>::max()) ? std::numeric_limits<double>::lowest() : cost_change / model_cost_change;
    if (rel > o->min_relative_decrease) {  // HandleSuccessfulStep
      x = cand; x_norm = std::sqrt(ambient_norm2(s, x, nullptr));
      const double tj = now_s();
      x_cost = evaluate(s, x.cam.data(), x.views.data(), x.pts.data(), o->loss_scale, threads, &e);
      gradient_and_norms(s, e, &grad, &sqn);
      sum->seconds_sweep += now_s() - tj;
      gmax = gradient_max_norm();
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
      radius = std::min(o->max_radius, radius);
      decrease_factor = 2.0; step_successful = true; ++sum->successful_steps;
    } else {
      radius = radius / decrease_factor; decrease_factor *= 2.0; step_successful = false; ++sum->unsuccessful_steps;
    }
    if (o->verbose) printf("%4d % .6e   % .2e %10.2e  %9.2e  %9.2e %9.2e\n", iteration, x_cost, cost_change, gmax, step_norm, rel, radius);
  }
  x.to(p);
  sum->iterations = iteration; sum->final_cost = x_cost; sum->final_radius = radius; sum->final_gradient_max_norm = gmax;
  sum->seconds_total = now_s() - t_start;
  return 0;
}

// reference calcReprojectionError (src/CameraCalibration.cpp:1026-1103): no sign folding of the
// camera parameters, (float) cast of the integer scale, error list in input order.
int lo_reproj_stats(const lifcal_ba_problem* p, double thr, lifcal_ba_stats* out, double* errors_2n) {
  if (int rc = validate(p)) return rc;
  Config cfg(p->config);
  const double sc = (double)(float)p->scale;
  const double spx = p->spx / sc, spy = p->spy / sc;
  double c_raw[2] = {(p->cam[3] + 0.5) * sc - 0.5, (p->cam[4] + 0.5) * sc - 0.5};
  const double* rad = cfg.n_radial > 0 ? p->cam + 5 : nullptr;
  const double* tan = cfg.tangential ? p->cam + 5 + cfg.n_radial : nullptr;
  double sx = 0, sy = 0, mx = 0, my = 0; uint32_t n = 0, inl = 0;
  for (uint32_t i = 0; i < p->n_obs; ++i) {
    double RT[3][4]; const double* view = p->views + 6 * (size_t)p->fr[i]; const double* P = p->pts + 3 * (size_t)p->pt[i];
    lo::rigid_transform<double>(view, view + 3, RT);
    double pc[3]; for (int k = 0; k < 3; ++k) pc[k] = RT[k][0] * P[0] + RT[k][1] * P[1] + RT[k][2] * P[2] + RT[k][3] * 1.0;
    double ml[2] = {p->mcx[i], p->mcy[i]}, x, y;
    lo::project_point<double>(x, y, pc, spx, spy, p->cam[0], p->cam[1], p->cam[2], c_raw, ml, rad, cfg.n_radial, tan, cfg.ml_center_adj);
    const double ex = x - p->u[i], ey = y - p->v[i];
    if (std::fabs(ex) > mx) mx = std::fabs(ex);
    if (std::fabs(ey) > my) my = std::fabs(ey);
    if (errors_2n) { errors_2n[2 * (size_t)i] = ex; errors_2n[2 * (size_t)i + 1] = ey; }
    if (ex * ex + ey * ey <= thr * thr) ++inl;
    sx += ex * ex; sy += ey * ey; ++n;
  }
  out->std_x = std::sqrt(sx / n); out->std_y = std::sqrt(sy / n); out->mae_x = mx; out->mae_y = my;
  out->num_points = n; out->num_inliers = inl;
  return 0;
}

int lo_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

// ------------------------------------------------------------------------------------------------
// reference src/CameraCalibration.cpp:456-499, CameraCalibration::initPlenopticParameters.
//   :469-471  a = ones(n, 2), b(n)                    :479 a(row,0) = virtual depth
//   :481-483  p_c = worldToCam * P.homogeneous();  b = fL z / (z - fL)
//   :485-490  rows with v < 2 or b < 0 are zeroed     :491 x = a.jacobiSvd(ThinU | ThinV).solve(b)
// Eigen (out of tree) computes the SVD of the n x 2 matrix with two-sided Jacobi sweeps after a QR step; for two columns
// that is one Givens rotation of the column pair, restated here as a one-sided (Hestenes) Jacobi SVD: rotate the two
// columns until they are orthogonal, singular values = column norms, x = V S^+ U^T b with S^+ dropping singular values
// <= 2 eps sigma_max (Eigen's default threshold: diagSize * epsilon).
// ------------------------------------------------------------------------------------------------
int lo_init_plenoptic(const lifcal_init_problem* p, lifcal_init_result* out) {
  if (!p || !out) return -1;
  const uint64_t n = p->n;
  std::vector<double> c0(n), c1(n), rhs(n);
  uint64_t used = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (p->fr[i] >= p->n_frames || p->pt[i] >= p->n_points) return -4;
    const double* M = p->world_to_cam + (size_t)p->fr[i] * 16;   // column-major Matrix4d
    const double* P = p->pts + 3 * (size_t)p->pt[i];
    double pc[4];
    for (int r = 0; r < 4; ++r) pc[r] = M[r] * P[0] + M[4 + r] * P[1] + M[8 + r] * P[2] + M[12 + r] * 1.0;
    const double z = pc[2];
    double a0 = p->vdepth[i], a1 = 1.0, b = (p->fL_init * z) / (z - p->fL_init);
    if ((p->vdepth[i] < 2) || (b < 0)) { a0 = 0; a1 = 0; b = 0; } else ++used;
    c0[i] = a0; c1[i] = a1; rhs[i] = b;
  }
  // one-sided Jacobi on the column pair; V accumulates the rotations
  double V[2][2] = {{1, 0}, {0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    long double al = 0, be = 0, ga = 0;
    for (uint64_t i = 0; i < n; ++i) { al += (long double)c0[i] * c0[i]; be += (long double)c1[i] * c1[i]; ga += (long double)c0[i] * c1[i]; }
    if (std::fabs((double)ga) <= 1e-300 || std::fabs((double)ga) <= 2.220446049250313e-16 * std::sqrt((double)al * (double)be)) break;
    const double zeta = (double)((be - al) / (2.0L * ga));
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
    const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
    for (uint64_t i = 0; i < n; ++i) { const double x0 = c0[i], x1 = c1[i]; c0[i] = cs * x0 - sn * x1; c1[i] = sn * x0 + cs * x1; }
    for (int r = 0; r < 2; ++r) { const double v0 = V[r][0], v1 = V[r][1]; V[r][0] = cs * v0 - sn * v1; V[r][1] = sn * v0 + cs * v1; }
  }
  long double s0 = 0, s1 = 0, u0b = 0, u1b = 0;
  for (uint64_t i = 0; i < n; ++i) { s0 += (long double)c0[i] * c0[i]; s1 += (long double)c1[i] * c1[i]; u0b += (long double)c0[i] * rhs[i]; u1b += (long double)c1[i] * rhs[i]; }
  const double sg0 = std::sqrt((double)s0), sg1 = std::sqrt((double)s1), smax = std::max(sg0, sg1), thr = 2.0 * 2.220446049250313e-16 * smax;
  double x[2] = {0, 0}; int rank = 0;
  // U_k = c_k / sigma_k, so V S^+ U^T b = sum_k V[:,k] (c_k . b) / sigma_k^2
  if (sg0 > thr && sg0 > 0) { const double w = (double)u0b / (double)s0; x[0] += V[0][0] * w; x[1] += V[1][0] * w; ++rank; }
  if (sg1 > thr && sg1 > 0) { const double w = (double)u1b / (double)s1; x[0] += V[0][1] * w; x[1] += V[1][1] * w; ++rank; }
  out->B_init = x[0]; out->bL0_init = x[1]; out->n_used = used; out->rank = rank; out->reserved = 0;
  return 0;
}


}  // extern "C"
