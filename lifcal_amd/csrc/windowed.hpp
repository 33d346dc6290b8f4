// windowed.hpp — lifcal_ba_solve_windowed: frame-windowed ("streaming") bundle adjustment for long sequences
// (include/lifcal_ba.h; BASELINE configs[4]: recalib, 2000 frames, pose + point refinement).  Host code; included at the end of
// lifcal_ba.hip.  No reference counterpart ("Config 5's streaming has no reference counterpart; implement as the same
// sharding over a sliding window of frames"): every window is an ordinary lifcal_ba problem — same kernels, same point sharding
// across ranks — built from the window's frames and the points they observe; only one window is resident on the device.
#pragma once
#include <algorithm>
#include <vector>

// every rank leaves a window with the same verdict: a rank-local failure (create, fixed frames, solve) travels as one all-reduced
// double through the borrowed collectives BEFORE any rank enters the next stage's collectives (a rank that returned alone left the
// others waiting in the window's all-reduces)
static int windowed_agree(lifcal_ba_handle* comm_template, const lifcal_ba_options& opt, double* dev_flag, int rc) {
  if (opt.world_size <= 1 || !comm_template) return rc;
  const double mine = rc ? 1.0 : 0.0;
  double all = 1.0;
  if (hipMemcpy(dev_flag, &mine, sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return LIFCAL_BA_ERR_HIP;
  if (int rc2 = do_allreduce(comm_template, dev_flag, 1)) return rc2;
  if (hipStreamSynchronize(comm_template->stream) != hipSuccess || hipMemcpy(&all, dev_flag, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return LIFCAL_BA_ERR_HIP;
  if (rc) return rc;
  if (all != 0.0) { g_last_error = "lifcal_ba_solve_windowed: another rank failed in this window"; return LIFCAL_BA_ERR_COMM; }
  return 0;
}

static int solve_windowed_impl(const lifcal_ba_problem* p, const lifcal_ba_options* o, uint32_t window_frames, uint32_t overlap_frames,
                               lifcal_ba_handle* comm_template, lifcal_ba_window_report* per_window, uint32_t* n_windows, double* dev_flag) {
  if (!p || !n_windows || window_frames == 0 || overlap_frames >= window_frames) return LIFCAL_BA_ERR_INVALID_ARG;
  if (int rc = lifcal::plan_validate(p)) return rc;
  lifcal_ba_options opt; if (o) opt = *o; else lifcal_ba_default_options(&opt);
  if (opt.world_size > 1 && !comm_template) { g_last_error = "lifcal_ba_solve_windowed: world_size > 1 needs a handle carrying the collectives"; return LIFCAL_BA_ERR_INVALID_ARG; }
  const uint32_t F = p->n_frames, N = p->n_obs, P = p->n_points;
  const uint32_t capacity = per_window ? *n_windows : 0;
  *n_windows = 0;
  // observations by frame (counting sort: input order is kept inside a frame)
  std::vector<uint32_t> f0(F + 1, 0), by_frame(N);
  for (uint32_t i = 0; i < N; ++i) ++f0[p->fr[i] + 1];
  for (uint32_t f = 0; f < F; ++f) f0[f + 1] += f0[f];
  { std::vector<uint32_t> fill(f0.begin(), f0.end() - 1); for (uint32_t i = 0; i < N; ++i) by_frame[fill[p->fr[i]]++] = i; }
  const uint32_t step = window_frames - overlap_frames;
  std::vector<int32_t> local_of(P, -1);
  uint32_t widx = 0;
  for (uint32_t a = 0; a < F; a += step, ++widx) {
    const uint32_t b = std::min(F, a + window_frames);
    const uint32_t nf = b - a, n = f0[b] - f0[a];
    // the window's points, in ascending global id
    std::vector<uint32_t> pts_ids;
    for (uint32_t k = f0[a]; k < f0[b]; ++k) { const uint32_t q = p->pt[by_frame[k]]; if (local_of[q] < 0) { local_of[q] = 0; pts_ids.push_back(q); } }
    std::sort(pts_ids.begin(), pts_ids.end());
    for (uint32_t j = 0; j < pts_ids.size(); ++j) local_of[pts_ids[j]] = (int32_t)j;
    std::vector<double> u(n), v(n), mcx(n), mcy(n), pts(3 * pts_ids.size());
    std::vector<uint32_t> pt(n), fr(n);
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t i = by_frame[f0[a] + k];
      u[k] = p->u[i]; v[k] = p->v[i]; mcx[k] = p->mcx[i]; mcy[k] = p->mcy[i];
      pt[k] = (uint32_t)local_of[p->pt[i]]; fr[k] = p->fr[i] - a;
    }
    for (size_t j = 0; j < pts_ids.size(); ++j) for (int c = 0; c < 3; ++c) pts[3 * j + c] = p->pts[3 * (size_t)pts_ids[j] + c];
    // distance constraints whose two points both belong to the window
    std::vector<uint32_t> ci, cj; std::vector<double> cd, cs;
    uint32_t dropped = 0;   // constraints with exactly one point in the window cannot be applied there: reported, not silent
    if (p->use_constraints && p->n_constraints && p->c_i && p->c_j)
      for (uint32_t c = 0; c < p->n_constraints; ++c) {
        const bool in_i = local_of[p->c_i[c]] >= 0, in_j = local_of[p->c_j[c]] >= 0;
        if (in_i && in_j) { ci.push_back((uint32_t)local_of[p->c_i[c]]); cj.push_back((uint32_t)local_of[p->c_j[c]]); cd.push_back(p->c_dist[c]); cs.push_back(p->c_sigma[c]); }
        else if (in_i != in_j) ++dropped;
      }
    lifcal_ba_problem sub = *p;
    sub.n_obs = n; sub.n_frames = nf; sub.n_points = (uint32_t)pts_ids.size(); sub.n_constraints = (uint32_t)ci.size();
    sub.u = u.data(); sub.v = v.data(); sub.mcx = mcx.data(); sub.mcy = mcy.data(); sub.pt = pt.data(); sub.fr = fr.data();
    sub.views = p->views + 6 * (size_t)a;    // in place: the caller's poses of this window
    sub.pts = pts.data();
    sub.c_i = ci.empty() ? nullptr : ci.data(); sub.c_j = cj.empty() ? nullptr : cj.data();
    sub.c_dist = cd.empty() ? nullptr : cd.data(); sub.c_sigma = cs.empty() ? nullptr : cs.data();
    lifcal_ba_window_report rep{};
    rep.first_frame = a; rep.n_frames = nf; rep.n_points = sub.n_points; rep.n_obs = n; rep.n_dropped_constraints = dropped;
    int rc = 0;
    if (n > 0) {
      lifcal_ba_handle* h = nullptr;
      rc = lifcal_ba_create(&sub, &opt, &h);
      if (rc != 0) h = nullptr;
      if (rc == 0) {
        if (comm_template) {   // the collectives of the caller's handle serve every window (the communicator stays the caller's)
          h->hook = comm_template->hook; h->hook_ctx = comm_template->hook_ctx;
          h->ghook = comm_template->ghook; h->ghook_ctx = comm_template->ghook_ctx;
          h->comm = comm_template->comm; h->comm_borrowed = true;
        }
        if (widx > 0 && overlap_frames > 0) {   // poses the previous window has refined stay as they are
          std::vector<uint8_t> fixed(nf, 0);
          for (uint32_t f = 0; f < std::min(nf, overlap_frames); ++f) fixed[f] = 1;
          rep.n_fixed_frames = std::min(nf, overlap_frames);
          rc = lifcal_ba_set_fixed_frames(h, fixed.data());
        }
      }
      rc = windowed_agree(comm_template, opt, dev_flag, rc);        // nobody enters the solve's collectives alone (all zero or all non-zero from here)
      if (rc == 0) {
        rc = lifcal_ba_solve(h, &rep.summary);
        rc = windowed_agree(comm_template, opt, dev_flag, rc);
      }
      if (h) lifcal_ba_destroy(h);
      if (rc == 0) for (size_t j = 0; j < pts_ids.size(); ++j) for (int c = 0; c < 3; ++c) p->pts[3 * (size_t)pts_ids[j] + c] = pts[3 * j + c];
    }
    for (uint32_t q : pts_ids) local_of[q] = -1;
    if (rc) return rc;
    if (per_window && widx < capacity) per_window[widx] = rep;
    *n_windows = widx + 1;
    if (b == F) break;
  }
  return 0;
}

extern "C" int lifcal_ba_solve_windowed(const lifcal_ba_problem* p, const lifcal_ba_options* o, uint32_t window_frames, uint32_t overlap_frames,
                                        lifcal_ba_handle* comm_template, lifcal_ba_window_report* per_window, uint32_t* n_windows) {
  double* dev_flag = nullptr;
  if (comm_template && o && o->world_size > 1 && hipMalloc((void**)&dev_flag, sizeof(double)) != hipSuccess) return LIFCAL_BA_ERR_NOMEM;
  int rc;
  try {
    rc = solve_windowed_impl(p, o, window_frames, overlap_frames, comm_template, per_window, n_windows, dev_flag);
  } catch (const std::bad_alloc&) {   // (no exception crosses the C ABI)
    g_last_error = "lifcal_ba_solve_windowed: out of host memory";
    rc = LIFCAL_BA_ERR_NOMEM;
  }
  if (dev_flag) (void)hipFree(dev_flag);
  return rc;
}
