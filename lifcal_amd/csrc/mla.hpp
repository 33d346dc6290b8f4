// mla.hpp — micro-lens grid, per-pixel lens maps, epipolar web and the projection of virtual-image points into the micro
// images (include/lifcal_mla.h).  Included at the end of lifcal_ba.hip (uses its g_last_error / error codes).
//
// What runs where: the lens list and the web are small and order-dependent (a chain of float / double operations whose
// rounding has to be reproduced), so the host builds them; the maps are per-pixel work (one lane per lens row for the
// validity discs, one lane per pixel for the nearest-lens ring search) and the projection is per-point work (count, prefix
// sum, fill), so the device does those.  All of it reproduces the reference's arithmetic bit for bit: its types, its
// evaluation order, no fused multiply-add (`#pragma clang fp contract(off)`; hipcc contracts by default on host and device).
#pragma once
#include <hipcub/hipcub.hpp>

#include "../../include/lifcal_mla.h"

namespace mla {

struct Line { double ex, ey, dist; };

// EpiPolarLine::EpiPolarLine (src/MicroLensGrid/EpiPolarLine.cpp:17-33): direction normalised unless its squared length is exactly one
inline Line make_line(double x, double y, double dist) {
#pragma clang fp contract(off)
  Line l{x, y, dist};
  const double n2 = l.ex * l.ex + l.ey * l.ey;
  if (n2 != 1.0) { const double n = std::sqrt(n2); l.ex /= n; l.ey /= n; }
  return l;
}
// EpiPolarLine::add (:40-48): vector sum of the two base lines
inline Line add_lines(const Line& a, const Line& b) {
#pragma clang fp contract(off)
  const double x = a.ex * a.dist + b.ex * b.dist;
  const double y = a.ey * a.dist + b.ey * b.dist;
  return make_line(x, y, std::sqrt(x * x + y * y));
}

struct Host {
  lifcal_mla_params prm;
  float im_center[2], offset_cv[2], valid_r, valid_r2;
  std::vector<float> cx, cy;
  std::vector<int32_t> type;
  std::vector<Line> web;            // flattened epiLineWeb
  std::vector<int32_t> web_group;
  int32_t n_groups = 0;
};

// MicroLensGrid::readInGrid, the values derived from the file (src/MicroLensGrid/MicroLensGrid.cpp:60-63, 107-111, 165-166)
inline void derive(Host& g, const lifcal_mla_params& p) {
#pragma clang fp contract(off)
  g.prm = p;
  g.im_center[0] = (float)p.width / 2.0f - 0.5f;
  g.im_center[1] = (float)p.height / 2.0f - 0.5f;
  const float lens_border = 1.0f;   // the reference overrides the file's value
  g.valid_r = p.lens_diameter * 0.5f - lens_border;
  g.valid_r2 = g.valid_r * g.valid_r;
  g.offset_cv[0] = p.offset[0] + g.im_center[0];
  g.offset_cv[1] = -p.offset[1] + g.im_center[1];
}

// MicroLensGrid::createGrid (src/MicroLensGrid/MicroLensGrid.cpp:186-270): two interleaved rectangular sub-grids; a lens is
// a function of its sub-grid coordinates alone, so the list is filled by index (column-major within each sub-grid, as the
// reference's nested loops produce it)
inline void build_lenses(Host& g) {
#pragma clang fp contract(off)
  const float d = g.prm.lens_diameter, bx = g.prm.lens_base_y[0], by = g.prm.lens_base_y[1];
  const float x_min = -g.im_center[0] - g.prm.offset[0] - d / 2.0f, x_max = g.im_center[0] - g.prm.offset[0] + d / 2.0f;
  const float y_min = -g.im_center[1] - g.prm.offset[1] - d / 2.0f, y_max = g.im_center[1] - g.prm.offset[1] + d / 2.0f;
  const float pitch_y = 2.0f * by * d;
  struct Sub { int x0, x1, y0, y1; } sub[2];
  sub[0] = {(int)std::ceil(x_min / d), (int)(x_max / d), (int)std::ceil(y_min / pitch_y), (int)(y_max / pitch_y)};
  sub[1] = {(int)std::ceil(x_min / d - bx - 1.0f), (int)(x_max / d - bx - 1.0f), (int)std::ceil(y_min / pitch_y - 0.5f), (int)(y_max / pitch_y - 0.5f)};
  const bool rot = g.prm.rotation_on_grid != 0;
  const float ca = rot ? std::cos(g.prm.rotation) : 0.0f, sa = rot ? std::sin(g.prm.rotation) : 0.0f;
  int64_t total = 0;
  for (const Sub& s : sub) total += (int64_t)(s.x1 - s.x0 + 1) * (s.y1 - s.y0 + 1);
  g.cx.clear(); g.cy.clear(); g.type.clear();
  if (total <= 0) return;
  g.cx.reserve((size_t)total); g.cy.reserve((size_t)total); g.type.reserve((size_t)total);
  for (int k = 0; k < 2; ++k) {
    const Sub& s = sub[k];
    for (int x = s.x0; x <= s.x1; ++x) {
      const int t = ((x % 3) + 3) % 3;
      const float gx = k == 0 ? (float)x * d : ((float)x + 1.0f + bx) * d;
      for (int y = s.y0; y <= s.y1; ++y) {
        const float gy = k == 0 ? (float)y * d * 2.0f * by : (((float)y * 2.0f + 1.0f) * by) * d;
        float lx, ly;
        if (rot) { lx = g.offset_cv[0] + (gx * ca - gy * sa); ly = g.offset_cv[1] - (gx * sa + gy * ca); }
        else { lx = g.offset_cv[0] + gx; ly = g.offset_cv[1] - gy; }
        g.cx.push_back(lx); g.cy.push_back(ly); g.type.push_back(t);
      }
    }
  }
}

// CameraCalibration::defineEpiPolarLines (src/CameraCalibration.cpp:521-632).  The lattice vectors are reached by chains of
// additions whose rounding decides the float-equality grouping, so the chains are the reference's: two diagonal zig-zags up
// to ten diameters, then every line extended along the row direction; insertion into groups sorted by length.
inline void build_web(Host& g) {
#pragma clang fp contract(off)
  const double d = g.prm.lens_diameter;
  const float max_dist = g.prm.lens_diameter * 10;
  const double h = std::sqrt(0.75);
  Line row = make_line(1, 0, d), up = make_line(0.5, h, d), up_neg = make_line(-0.5, -h, d), down = make_line(0.5, -h, d), down_neg = make_line(-0.5, h, d);
  if (g.prm.rotation_on_grid) {
    const double ca = (double)std::cos(g.prm.rotation), sa = (double)std::sin(g.prm.rotation);   // cos(float) is the float overload
    for (Line* l : {&row, &up, &up_neg, &down, &down_neg}) {
      const double x = l->ex, y = l->ey;
      l->ex = x * ca + y * sa;
      l->ey = -x * sa + y * ca;
    }
  }
  std::vector<Line> lines{up, down};
  for (int i = 0; lines.back().dist < max_dist; ++i) {
    const Line a = add_lines(lines[2 * i], (i % 2 == 0) ? down_neg : up);
    const Line b = add_lines(lines[2 * i + 1], (i % 2 == 0) ? up_neg : down);
    lines.push_back(a); lines.push_back(b);
  }
  lines.push_back(row);
  const size_t seeds = lines.size();
  for (size_t k = 0; k < seeds; ++k) {
    Line last = lines[k];
    while (last.dist < max_dist) { last = add_lines(last, row); lines.push_back(last); }
  }
  std::vector<std::vector<Line>> groups{{lines[0]}};
  for (size_t k = 1; k < lines.size(); ++k) {
    const Line& l = lines[k];
    if (l.ey == -1.0 || l.dist > max_dist) continue;
    size_t at = 0;
    int where = 0;   // 0: longer than every group, 1: joins groups[at], 2: goes before groups[at]
    for (; at < groups.size(); ++at) {
      if ((float)groups[at][0].dist == (float)l.dist) { where = 1; break; }
      if (groups[at][0].dist > l.dist) { where = 2; break; }
    }
    if (where == 1) groups[at].push_back(l);
    else if (where == 2) groups.insert(groups.begin() + at, std::vector<Line>{l});
    else groups.push_back({l});
  }
  g.web.clear(); g.web_group.clear();
  g.n_groups = (int32_t)groups.size();
  for (size_t gi = 0; gi < groups.size(); ++gi)
    for (const Line& l : groups[gi]) { g.web.push_back(l); g.web_group.push_back((int32_t)gi); }
}

// ------------------------------------------------------------------------------------------------ device side
struct Dev {
  int32_t width, height, n_lenses, n_lines;
  float lens_diameter, valid_r, valid_r2;
  const float* cx; const float* cy;
  int32_t* map_ml; int32_t* map_next;
  const float* w_dist; const float* w_ex; const float* w_ey;   // the web as the projection reads it: float casts of the doubles
  const double* w_first;                                        // length of the first line of the line's group (loop exit test)
};

constexpr int kDiscRows = 32;   // rows a validity disc can span: 2 (d/2 - 1) + 2 <= 32 up to 31-pixel lenses; larger lenses loop

// defineMlMaps, first half (src/MicroLensGrid/MicroLensGrid.cpp:354-377): a pixel belongs to the LAST lens in list order whose
// disc covers it; coverage of one pixel by one lens does not depend on other lenses, so lanes take (lens, row) pairs and the
// list order becomes an atomic max over lens indices.
__global__ __launch_bounds__(256) void k_mla_discs(Dev m) {
#pragma clang fp contract(off)
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lens = (int)(tid / kDiscRows), row = (int)(tid % kDiscRows);
  if (lens >= m.n_lenses) return;
  const double cx = m.cx[lens], cy = m.cy[lens];
  const double y_first = ceil(cy - m.valid_r);
  for (double yd = y_first + row; yd <= cy + m.valid_r; yd += kDiscRows) {
    if (!(yd >= -2147483648.0 && yd <= 2147483647.0)) break;
    const int y = (int)yd;
    if (y < 0 || y >= m.height) continue;
    const double y2 = (y - cy) * (y - cy);
    const double span2 = m.valid_r2 - y2;
    if (!(span2 >= 0)) continue;   // the reference's x loop has a false condition from the start in this case
    const double x_first = ceil(cx - sqrt(span2));
    if (!(x_first >= -2147483648.0 && x_first <= 2147483647.0)) continue;
    for (int x = (int)x_first; (x - cx) * (x - cx) <= span2; ++x) {
      if (x < 0 || x >= m.width) { if (x >= m.width) break; continue; }
      atomicMax(&m.map_ml[x + (int64_t)y * m.width], lens);
    }
  }
}

// defineMlMaps, second half (:379-420): pixels outside every disc take the nearest lens among those that own a pixel on the
// first square ring around them that holds any; ring scanned column by column, first found wins ties.  One lane per pixel.
__global__ __launch_bounds__(256) void k_mla_nearest(Dev m) {
#pragma clang fp contract(off)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)m.width * m.height) return;
  const int x = (int)(idx % m.width), y = (int)(idx / m.width);
  int32_t best = m.map_ml[idx];
  if (best == -1) {
    float dist2 = -1;
    for (int d = 1; best == -1 && d <= m.width + m.height; ++d) {
      for (int dx = -d; dx <= d; ++dx) {
        if (x + dx < 0) continue;
        if (x + dx >= m.width) break;
        const bool edge_col = dx == -d || dx == d;
        for (int dy = -d; dy <= d; dy += edge_col ? 1 : 2 * d) {   // inner columns touch the ring at its top and bottom only
          if (y + dy < 0) continue;
          if (y + dy >= m.height) break;
          const int32_t cand = m.map_ml[idx + dx + (int64_t)dy * m.width];
          if (cand != -1) {
            const float ccx = m.cx[cand], ccy = m.cy[cand];
            const float dn = (ccx - x) * (ccx - x) + (ccy - y) * (ccy - y);
            if (dn < dist2 || dist2 < 0) { best = cand; dist2 = dn; }
          }
        }
      }
    }
  }
  m.map_next[idx] = best;
}

struct ProjectArgs {
  uint64_t n;
  const double* x; const double* y; const double* vd;
  const uint32_t* fr; const uint32_t* pt;
  int32_t scale;
  uint64_t* counts;          // count pass: observations per image point
  const uint64_t* offsets;   // fill pass: exclusive prefix sum of counts
  double* u; double* v; double* mcx; double* mcy; uint32_t* src; uint32_t* ofr; uint32_t* opt;
};

constexpr int kMaxWebLds = 1024;

// projectPointsToRawImage (src/CameraCalibration.cpp:651-765), one lane per image point.  FILL = false counts, FILL = true
// writes at the point's offset, so the list comes out in the reference's push order.
template <bool FILL>
__global__ __launch_bounds__(256) void k_mla_project(Dev m, ProjectArgs a) {
#pragma clang fp contract(off)
  __shared__ float s_dist[kMaxWebLds], s_ex[kMaxWebLds], s_ey[kMaxWebLds];
  __shared__ double s_first[kMaxWebLds];
  const int n_lines = m.n_lines < kMaxWebLds ? m.n_lines : kMaxWebLds;   // create() refuses webs beyond kMaxWebLds
  for (int k = threadIdx.x; k < n_lines; k += blockDim.x) { s_dist[k] = m.w_dist[k]; s_ex[k] = m.w_ex[k]; s_ey[k] = m.w_ey[k]; s_first[k] = m.w_first[k]; }
  __syncthreads();
  const float w1 = (float)(m.width - 1), h1 = (float)(m.height - 1);
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.n; p += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t cnt = 0;
    const uint64_t base = FILL ? a.offsets[p] : 0;
    const float vdepth = (float)a.vd[p];
    const float xs = (float)a.x[p], ys = (float)a.y[p];
    const float radius = m.lens_diameter * 0.5f * vdepth + 2.0f;
    const float radius2 = radius * radius;
    const float xu = (float)a.scale * (xs + 0.5f) - 0.5f;
    const float yu = (float)a.scale * (ys + 0.5f) - 0.5f;
    // coordinates a float cannot carry into an int, or left of / above the image, are outside the reference's contract
    // (it indexes the map unchecked): such points yield no observation
    bool live = ((double)vdepth > 2.0 && (double)vdepth < 20.0) && xu > -1.0e9f && xu < 1.0e9f && yu > -1.0e9f && yu < 1.0e9f;
    int32_t nearest = -1;
    float ncx = 0, ncy = 0;
    if (live) {
      int xi = (int)(xu + 0.5f); if (xi >= m.width) xi = m.width - 1;
      int yi = (int)(yu + 0.5f); if (yi >= m.height) yi = m.height - 1;
      live = xi >= 0 && yi >= 0;
      if (live) nearest = m.map_next[xi + (int64_t)m.width * yi];
      live = live && nearest != -1;
    }
    if (live) {
      ncx = m.cx[nearest]; ncy = m.cy[nearest];
      const float dx = ncx - xu, dy = ncy - yu;
      live = !(dx * dx + dy * dy > radius2);
    }
    auto emit = [&](float lcx, float lcy) {   // :746-763
#pragma clang fp contract(off)
      const float xr = (xu - lcx) / vdepth + lcx;
      const float yr = (yu - lcy) / vdepth + lcy;
      if (!(xr >= 0 && xr <= w1 && yr >= 0 && yr <= h1)) return;
      const float ex = xr - lcx, ey = yr - lcy;
      if (ex * ex + ey * ey >= m.valid_r2) return;
      if (FILL) {
        const uint64_t o = base + cnt;
        a.u[o] = xr; a.v[o] = yr; a.mcx[o] = lcx; a.mcy[o] = lcy;
        if (a.src) a.src[o] = (uint32_t)p;
        if (a.ofr) a.ofr[o] = a.fr[p];
        if (a.opt) a.opt[o] = a.pt[p];
      }
      ++cnt;
    };
    if (live) {
      emit(ncx, ncy);
      for (int k = 0; k < n_lines; ++k) {
        if (s_first[k] > (double)radius) break;
        const float bl = s_dist[k];
        for (int sgn = 0; sgn < 2; ++sgn) {
          const float ex = sgn ? -s_ex[k] : s_ex[k], ey = sgn ? -s_ey[k] : s_ey[k];
          const float pcx = ncx + bl * ex, pcy = ncy + bl * ey;   // where the lattice puts the neighbour's centre
          const float dx = pcx - xu, dy = pcy - yu;
          if (dx * dx + dy * dy > radius2) continue;
          int ci = (int)((double)pcx + 0.5), cj = (int)((double)pcy + 0.5);
          ci = ci < 0 ? 0 : (ci >= m.width ? m.width - 1 : ci);
          cj = cj < 0 ? 0 : (cj >= m.height ? m.height - 1 : cj);
          const int32_t lens = m.map_ml[ci + (int64_t)cj * m.width];
          if (lens == -1) continue;
          emit(m.cx[lens], m.cy[lens]);
        }
      }
    }
    if (!FILL) a.counts[p] = cnt;
  }
}

}  // namespace mla

struct lifcal_mla_handle {
  mla::Host host;
  mla::Dev dev{};
  int32_t device = 0;
  std::vector<void*> allocs;
  // scratch of lifcal_mla_project, kept between calls and grown on demand (hipFree costs about 2 ms per call otherwise):
  // the k-th request of a call reuses the k-th buffer
  std::vector<std::pair<void*, size_t>> scratch;
};

namespace mla {

template <class T>
hipError_t upload(lifcal_mla_handle* h, const T** dst, const std::vector<T>& v) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess) return e;
  h->allocs.push_back(q);
  *dst = (const T*)q;
  return v.empty() ? hipSuccess : hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

inline int select_device(int32_t device, const char* who) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_last_error = std::string(who) + ": no HIP device (this path has no CPU fallback)"; return LIFCAL_BA_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { g_last_error = std::string(who) + ": hipSetDevice failed"; return LIFCAL_BA_ERR_NO_DEVICE; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string(who) + ": device is " + prop.gcnArchName + ", this library carries gfx950 code objects only"; return LIFCAL_BA_ERR_NO_DEVICE;
  }
  return 0;
}

}  // namespace mla

extern "C" {

int lifcal_mla_create(const lifcal_mla_params* p, int32_t device, lifcal_mla_handle** out) {
  if (!out) return LIFCAL_BA_ERR_INVALID_ARG;
  *out = nullptr;
  if (!p || p->width <= 0 || p->height <= 0 || (int64_t)p->width * p->height > (int64_t)1 << 30 || !(p->lens_diameter > 2.0f) || !(p->lens_diameter < 1.0e4f) ||
      !(p->lens_base_y[1] > 0.0f) || !std::isfinite(p->lens_base_y[0]) || !std::isfinite(p->rotation) || !std::isfinite(p->offset[0]) || !std::isfinite(p->offset[1])) {
    g_last_error = "lifcal_mla_create: bad grid parameters"; return LIFCAL_BA_ERR_INVALID_ARG;
  }
  lifcal_mla_handle* h = new (std::nothrow) lifcal_mla_handle();
  if (!h) return LIFCAL_BA_ERR_NOMEM;
  mla::Host& g = h->host;
  mla::derive(g, *p);
  mla::build_lenses(g);
  mla::build_web(g);
  if (g.cx.empty() || g.cx.size() > (size_t)1 << 26 || g.web.size() > (size_t)mla::kMaxWebLds) {
    g_last_error = "lifcal_mla_create: lens grid is empty or too large"; delete h; return LIFCAL_BA_ERR_INVALID_ARG;
  }
  if (int rc = mla::select_device(device, "lifcal_mla_create")) { delete h; return rc; }
  h->device = device;
  mla::Dev& d = h->dev;
  d.width = p->width; d.height = p->height; d.n_lenses = (int32_t)g.cx.size(); d.n_lines = (int32_t)g.web.size();
  d.lens_diameter = p->lens_diameter; d.valid_r = g.valid_r; d.valid_r2 = g.valid_r2;
  std::vector<float> wd, wx, wy; std::vector<double> wf;
  {
    double first = 0; int32_t grp = -1;
    for (size_t k = 0; k < g.web.size(); ++k) {
      if (g.web_group[k] != grp) { grp = g.web_group[k]; first = g.web[k].dist; }
      wd.push_back((float)g.web[k].dist); wx.push_back((float)g.web[k].ex); wy.push_back((float)g.web[k].ey); wf.push_back(first);
    }
  }
  const size_t npix = (size_t)p->width * p->height;
  hipError_t e = mla::upload(h, &d.cx, g.cx);
  if (e == hipSuccess) e = mla::upload(h, &d.cy, g.cy);
  if (e == hipSuccess) e = mla::upload(h, &d.w_dist, wd);
  if (e == hipSuccess) e = mla::upload(h, &d.w_ex, wx);
  if (e == hipSuccess) e = mla::upload(h, &d.w_ey, wy);
  if (e == hipSuccess) e = mla::upload(h, &d.w_first, wf);
  if (e == hipSuccess) { e = hipMalloc((void**)&d.map_ml, npix * 4); if (e == hipSuccess) h->allocs.push_back(d.map_ml); }
  if (e == hipSuccess) { e = hipMalloc((void**)&d.map_next, npix * 4); if (e == hipSuccess) h->allocs.push_back(d.map_next); }
  if (e == hipSuccess) e = hipMemset(d.map_ml, 0xff, npix * 4);
  if (e == hipSuccess) {
    const int64_t lanes = (int64_t)d.n_lenses * mla::kDiscRows;
    hipLaunchKernelGGL(mla::k_mla_discs, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(mla::k_mla_nearest, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, 0, d);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) { g_last_error = std::string("lifcal_mla_create: ") + hipGetErrorString(e); lifcal_mla_destroy(h); return LIFCAL_BA_ERR_HIP; }
  *out = h;
  return 0;
}

void lifcal_mla_destroy(lifcal_mla_handle* h) {
  if (!h) return;
  if (!h->allocs.empty() || !h->scratch.empty()) {
    (void)hipSetDevice(h->device);
    for (void* q : h->allocs) (void)hipFree(q);
    for (auto& sl : h->scratch) if (sl.first) (void)hipFree(sl.first);
  }
  delete h;
}

int lifcal_mla_info(const lifcal_mla_handle* h, int32_t* n_lenses, int32_t* n_web_groups, int32_t* n_web_lines) {
  if (!h) return LIFCAL_BA_ERR_INVALID_ARG;
  if (n_lenses) *n_lenses = (int32_t)h->host.cx.size();
  if (n_web_groups) *n_web_groups = h->host.n_groups;
  if (n_web_lines) *n_web_lines = (int32_t)h->host.web.size();
  return 0;
}

int lifcal_mla_get_lenses(const lifcal_mla_handle* h, float* cx, float* cy, int32_t* type) {
  if (!h) return LIFCAL_BA_ERR_INVALID_ARG;
  const size_t n = h->host.cx.size();
  if (cx) std::memcpy(cx, h->host.cx.data(), n * 4);
  if (cy) std::memcpy(cy, h->host.cy.data(), n * 4);
  if (type) std::memcpy(type, h->host.type.data(), n * 4);
  return 0;
}

int lifcal_mla_get_maps(lifcal_mla_handle* h, int32_t* map_ml, int32_t* map_next) {
  if (!h) return LIFCAL_BA_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = (size_t)h->dev.width * h->dev.height * 4;
  if (map_ml) HIP_TRY(hipMemcpy(map_ml, h->dev.map_ml, bytes, hipMemcpyDeviceToHost));
  if (map_next) HIP_TRY(hipMemcpy(map_next, h->dev.map_next, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int lifcal_mla_get_web(const lifcal_mla_handle* h, double* dist, double* ex, double* ey, int32_t* group) {
  if (!h) return LIFCAL_BA_ERR_INVALID_ARG;
  for (size_t k = 0; k < h->host.web.size(); ++k) {
    if (dist) dist[k] = h->host.web[k].dist;
    if (ex) ex[k] = h->host.web[k].ex;
    if (ey) ey[k] = h->host.web[k].ey;
    if (group) group[k] = h->host.web_group[k];
  }
  return 0;
}

int lifcal_mla_project(lifcal_mla_handle* h, int32_t depth_to_raw_im_scale, const lifcal_mla_points* pts, lifcal_mla_observations* obs) {
  if (!h || !pts || !obs || (pts->n && (!pts->x || !pts->y || !pts->vdepth)) || pts->n > 0x7ffffff0ull || depth_to_raw_im_scale < 1 ||
      (obs->capacity && (!obs->u || !obs->v || !obs->mcx || !obs->mcy)) || (obs->fr && !pts->fr) || (obs->pt && !pts->pt)) {
    g_last_error = "lifcal_mla_project: bad argument"; return LIFCAL_BA_ERR_INVALID_ARG;
  }
  obs->n_obs = 0;
  if (pts->n == 0) return 0;
  HIP_TRY(hipSetDevice(h->device));
  size_t next_scratch = 0;
  auto release = [&]() {};   // buffers stay with the handle
  auto dev_alloc = [&](void** q, size_t bytes) -> hipError_t {
    if (bytes == 0) bytes = 8;
    if (next_scratch == h->scratch.size()) h->scratch.push_back({nullptr, 0});
    std::pair<void*, size_t>& slot = h->scratch[next_scratch++];
    if (slot.second < bytes) {
      if (slot.first) (void)hipFree(slot.first);
      slot = {nullptr, 0};
      const size_t want = bytes + bytes / 4;   // some slack: the next call is likely to be of similar size
      hipError_t e = hipMalloc(&slot.first, want);
      if (e != hipSuccess) { slot.first = nullptr; *q = nullptr; return e; }
      slot.second = want;
    }
    *q = slot.first;
    return hipSuccess;
  };
  auto up = [&](const void** q, const void* src, size_t bytes) -> hipError_t {
    if (!src) { *q = nullptr; return hipSuccess; }
    void* w = nullptr; hipError_t e = dev_alloc(&w, bytes); *q = w;
    return e == hipSuccess ? hipMemcpy(w, src, bytes, hipMemcpyHostToDevice) : e;
  };
  lifcal::PlanClock clk;
  mla::ProjectArgs a{};
  a.n = pts->n; a.scale = depth_to_raw_im_scale;
  const uint64_t n = pts->n;
  hipError_t e = up((const void**)&a.x, pts->x, n * 8);
  if (e == hipSuccess) e = up((const void**)&a.y, pts->y, n * 8);
  if (e == hipSuccess) e = up((const void**)&a.vd, pts->vdepth, n * 8);
  if (e == hipSuccess) e = up((const void**)&a.fr, obs->fr ? pts->fr : nullptr, n * 4);
  if (e == hipSuccess) e = up((const void**)&a.pt, obs->pt ? pts->pt : nullptr, n * 4);
  uint64_t* counts = nullptr; uint64_t* offsets = nullptr;
  if (e == hipSuccess) e = dev_alloc((void**)&counts, (n + 1) * 8);
  if (e == hipSuccess) e = dev_alloc((void**)&offsets, (n + 1) * 8);
  if (e == hipSuccess) e = hipMemset(counts, 0, (n + 1) * 8);   // entry n stays 0: the scan's last output is the total
  clk.lap("project: upload");
  uint64_t total = 0;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16);
  if (e == hipSuccess) {
    a.counts = counts;
    hipLaunchKernelGGL(mla::k_mla_project<false>, dim3(grid), dim3(256), 0, 0, h->dev, a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    size_t scratch_bytes = 0; void* scratch = nullptr;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, scratch_bytes, counts, offsets, (int)(n + 1));
    if (e == hipSuccess) e = dev_alloc(&scratch, scratch_bytes);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(scratch, scratch_bytes, counts, offsets, (int)(n + 1));
    if (e == hipSuccess) e = hipMemcpy(&total, offsets + n, 8, hipMemcpyDeviceToHost);
  }
  clk.lap("project: count + scan");
  if (e != hipSuccess) { release(); g_last_error = std::string("lifcal_mla_project: ") + hipGetErrorString(e); return LIFCAL_BA_ERR_HIP; }
  obs->n_obs = total;
  if (total > obs->capacity) { release(); return LIFCAL_MLA_MORE; }
  if (total == 0) { release(); return 0; }
  a.offsets = offsets;
  e = dev_alloc((void**)&a.u, total * 8);
  if (e == hipSuccess) e = dev_alloc((void**)&a.v, total * 8);
  if (e == hipSuccess) e = dev_alloc((void**)&a.mcx, total * 8);
  if (e == hipSuccess) e = dev_alloc((void**)&a.mcy, total * 8);
  if (e == hipSuccess && obs->src) e = dev_alloc((void**)&a.src, total * 4);
  if (e == hipSuccess && obs->fr) e = dev_alloc((void**)&a.ofr, total * 4);
  if (e == hipSuccess && obs->pt) e = dev_alloc((void**)&a.opt, total * 4);
  clk.lap("project: output alloc");
  if (e == hipSuccess) {
    hipLaunchKernelGGL(mla::k_mla_project<true>, dim3(grid), dim3(256), 0, 0, h->dev, a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  clk.lap("project: fill");
  if (e == hipSuccess) e = hipMemcpy(obs->u, a.u, total * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(obs->v, a.v, total * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(obs->mcx, a.mcx, total * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(obs->mcy, a.mcy, total * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess && obs->src) e = hipMemcpy(obs->src, a.src, total * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && obs->fr) e = hipMemcpy(obs->fr, a.ofr, total * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && obs->pt) e = hipMemcpy(obs->pt, a.opt, total * 4, hipMemcpyDeviceToHost);
  clk.lap("project: download");
  release();
  clk.lap("project: free");
  if (e != hipSuccess) { g_last_error = std::string("lifcal_mla_project: ") + hipGetErrorString(e); return LIFCAL_BA_ERR_HIP; }
  return 0;
}

}  // extern "C"
