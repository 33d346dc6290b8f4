// colmap.hpp — COLMAP sparse-model ingestion (include/lifcal_colmap.h).  Host code only; included at the end of lifcal_ba.hip.
// File layouts: COLMAP's published model format (cameras / images / points3D as little-endian .bin or as .txt); the reference
// reads them through colmap::Reconstruction::Read (src/CalibrationData/CalibrationData.cpp:64-74), COLMAP is not in this image.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/lifcal_colmap.h"

struct lifcal_colmap_model {
  lifcal_colmap_info info{};
  std::vector<int32_t> frame_ids;
  std::vector<double> quat, trans;            // 4F (w x y z, normalised), 3F
  std::vector<uint64_t> point_ids;            // P, ascending
  std::vector<double> pts;                    // 3P
  std::vector<double> x, y;                   // inlier image points, frame-major
  std::vector<uint32_t> fr, pt;
};

namespace lifcal_colmap {

// COLMAP camera models: id -> number of parameters (SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV, OPENCV_FISHEYE,
// FULL_OPENCV, FOV, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE, THIN_PRISM_FISHEYE)
inline int model_num_params(int id) {
  static const int n[11] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};
  return (id >= 0 && id < 11) ? n[id] : -1;
}
inline int model_id_from_name(const std::string& s) {
  static const char* names[11] = {"SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV", "FOV",
                                  "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE", "THIN_PRISM_FISHEYE"};
  for (int i = 0; i < 11; ++i) if (s == names[i]) return i;
  return -1;
}

struct Camera { int model = -1; uint64_t width = 0, height = 0; std::vector<double> params; };
struct Point2 { double x, y; uint64_t id; };
struct Image { uint32_t id = 0, camera = 0; double q[4] = {1, 0, 0, 0}, t[3] = {0, 0, 0}; std::vector<Point2> pts; };
struct Raw { std::map<uint32_t, Camera> cameras; std::vector<Image> images; std::vector<std::pair<uint64_t, std::array<double, 3>>> points; };

inline bool slurp(const std::string& path, std::string* out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::string s; char buf[1 << 16]; size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
  std::fclose(f);
  out->swap(s);
  return true;
}
inline bool exists(const std::string& path) { FILE* f = std::fopen(path.c_str(), "rb"); if (!f) return false; std::fclose(f); return true; }

// ---- binary ----
struct Cursor {
  const char* p; const char* end; bool ok = true;
  template <class T> T get() { T v{}; if ((size_t)(end - p) < sizeof(T)) { ok = false; p = end; return v; } std::memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
  void skip(size_t n) { if ((size_t)(end - p) < n) { ok = false; p = end; } else p += n; }
  std::string cstr() { const char* z = (const char*)std::memchr(p, 0, (size_t)(end - p)); if (!z) { ok = false; p = end; return ""; } std::string s(p, z); p = z + 1; return s; }
};

inline bool read_binary(const std::string& dir, Raw* r, std::string* err) {
  std::string buf;
  if (!slurp(dir + "/cameras.bin", &buf)) { *err = "cannot read cameras.bin"; return false; }
  {
    Cursor c{buf.data(), buf.data() + buf.size()};
    const uint64_t n = c.get<uint64_t>();
    for (uint64_t i = 0; i < n && c.ok; ++i) {
      const uint32_t id = c.get<uint32_t>();
      Camera cam; cam.model = c.get<int32_t>(); cam.width = c.get<uint64_t>(); cam.height = c.get<uint64_t>();
      const int np = model_num_params(cam.model);
      if (np < 0) { *err = "cameras.bin: unknown camera model id " + std::to_string(cam.model); return false; }
      for (int k = 0; k < np; ++k) cam.params.push_back(c.get<double>());
      r->cameras[id] = cam;
    }
    if (!c.ok) { *err = "cameras.bin is truncated"; return false; }
  }
  if (!slurp(dir + "/images.bin", &buf)) { *err = "cannot read images.bin"; return false; }
  {
    Cursor c{buf.data(), buf.data() + buf.size()};
    const uint64_t n = c.get<uint64_t>();
    for (uint64_t i = 0; i < n && c.ok; ++i) {
      Image im; im.id = c.get<uint32_t>();
      for (int k = 0; k < 4; ++k) im.q[k] = c.get<double>();
      for (int k = 0; k < 3; ++k) im.t[k] = c.get<double>();
      im.camera = c.get<uint32_t>();
      (void)c.cstr();   // image name
      const uint64_t np = c.get<uint64_t>();
      if (!c.ok || np > (uint64_t)(c.end - c.p) / 24) { *err = "images.bin is truncated"; return false; }
      im.pts.resize(np);
      for (uint64_t k = 0; k < np; ++k) { im.pts[k].x = c.get<double>(); im.pts[k].y = c.get<double>(); im.pts[k].id = c.get<uint64_t>(); }
      r->images.push_back(std::move(im));
    }
    if (!c.ok) { *err = "images.bin is truncated"; return false; }
  }
  if (!slurp(dir + "/points3D.bin", &buf)) { *err = "cannot read points3D.bin"; return false; }
  {
    Cursor c{buf.data(), buf.data() + buf.size()};
    const uint64_t n = c.get<uint64_t>();
    for (uint64_t i = 0; i < n && c.ok; ++i) {
      const uint64_t id = c.get<uint64_t>();
      std::array<double, 3> xyz; for (int k = 0; k < 3; ++k) xyz[k] = c.get<double>();
      c.skip(3);                 // rgb
      (void)c.get<double>();     // reprojection error
      const uint64_t tl = c.get<uint64_t>();
      if (!c.ok || tl > (uint64_t)(c.end - c.p) / 8) { *err = "points3D.bin is truncated"; return false; }
      c.skip((size_t)tl * 8);    // track: (image id u32, point2D index u32)
      r->points.push_back({id, xyz});
    }
    if (!c.ok) { *err = "points3D.bin is truncated"; return false; }
  }
  return true;
}

// ---- text ----
inline void trim(std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
  s = s.substr(a, b - a);
}
struct Lines {
  const std::string& buf; size_t pos = 0;
  explicit Lines(const std::string& b) : buf(b) {}
  bool next(std::string* line) {
    if (pos >= buf.size()) return false;
    size_t e = buf.find('\n', pos);
    if (e == std::string::npos) e = buf.size();
    *line = buf.substr(pos, e - pos); pos = e + 1;
    trim(*line);
    return true;
  }
};
struct Tok {   // whitespace-separated fields of one line
  const char* p;
  explicit Tok(const std::string& s) : p(s.c_str()) {}
  bool word(std::string* w) { while (*p == ' ' || *p == '\t') ++p; if (!*p) return false; const char* b = p; while (*p && *p != ' ' && *p != '\t') ++p; w->assign(b, p); return true; }
  bool f64(double* v) { while (*p == ' ' || *p == '\t') ++p; if (!*p) return false; char* e; *v = std::strtod(p, &e); if (e == p) return false; p = e; return true; }
  bool u64(uint64_t* v) {   // "-1" is COLMAP's invalid id (2^64 - 1)
    while (*p == ' ' || *p == '\t') ++p; if (!*p) return false; char* e;
    if (*p == '-') { const long long s = std::strtoll(p, &e, 10); if (e == p) return false; *v = (uint64_t)s; }
    else { *v = std::strtoull(p, &e, 10); if (e == p) return false; }
    p = e; return true;
  }
};

inline bool read_text(const std::string& dir, Raw* r, std::string* err) {
  std::string buf, line;
  if (!slurp(dir + "/cameras.txt", &buf)) { *err = "cannot read cameras.txt"; return false; }
  {
    Lines L(buf);
    while (L.next(&line)) {
      if (line.empty() || line[0] == '#') continue;
      Tok t(line); uint64_t id, w, h; std::string model;
      if (!t.u64(&id) || !t.word(&model) || !t.u64(&w) || !t.u64(&h)) { *err = "cameras.txt: malformed line"; return false; }
      Camera cam; cam.model = model_id_from_name(model); cam.width = w; cam.height = h;
      if (cam.model < 0) { *err = "cameras.txt: unknown camera model " + model; return false; }
      double v; while (t.f64(&v)) cam.params.push_back(v);
      r->cameras[(uint32_t)id] = cam;
    }
  }
  if (!slurp(dir + "/images.txt", &buf)) { *err = "cannot read images.txt"; return false; }
  {
    Lines L(buf);
    while (L.next(&line)) {
      if (line.empty() || line[0] == '#') continue;
      Tok t(line); Image im; uint64_t id, cam;
      if (!t.u64(&id)) { *err = "images.txt: malformed image line"; return false; }
      for (int k = 0; k < 4; ++k) if (!t.f64(&im.q[k])) { *err = "images.txt: malformed image line"; return false; }
      for (int k = 0; k < 3; ++k) if (!t.f64(&im.t[k])) { *err = "images.txt: malformed image line"; return false; }
      if (!t.u64(&cam)) { *err = "images.txt: malformed image line"; return false; }
      im.id = (uint32_t)id; im.camera = (uint32_t)cam;
      // the points line follows unconditionally (it is empty for an image without points)
      if (L.next(&line)) {
        Tok q(line); Point2 p;
        while (q.f64(&p.x)) { if (!q.f64(&p.y) || !q.u64(&p.id)) { *err = "images.txt: malformed points line"; return false; } im.pts.push_back(p); }
      }
      r->images.push_back(std::move(im));
    }
  }
  if (!slurp(dir + "/points3D.txt", &buf)) { *err = "cannot read points3D.txt"; return false; }
  {
    Lines L(buf);
    while (L.next(&line)) {
      if (line.empty() || line[0] == '#') continue;
      Tok t(line); uint64_t id; std::array<double, 3> xyz;
      if (!t.u64(&id) || !t.f64(&xyz[0]) || !t.f64(&xyz[1]) || !t.f64(&xyz[2])) { *err = "points3D.txt: malformed line"; return false; }
      r->points.push_back({id, xyz});
    }
  }
  return true;
}

// Eigen::Quaterniond::toRotationMatrix
inline void quat_to_matrix(const double q[4], double R[3][3]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0][0] = 1.0 - (tyy + tzz); R[0][1] = txy - twz;         R[0][2] = txz + twy;
  R[1][0] = txy + twz;         R[1][1] = 1.0 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy;         R[2][1] = tyz + twx;         R[2][2] = 1.0 - (txx + tyy);
}

// Eigen::MatrixBase::eulerAngles(0, 1, 2) (Eigen 3.3 / 3.4 Geometry/EulerAngles.h): angles (a0, a1, a2) with
// R = Rx(a0) Ry(a1) Rz(a2), a0 in [0, pi], a1 and a2 in [-pi, pi]  — reference src/CalibrationData/CalibrationData.cpp:531
inline void euler_xyz(const double R[3][3], double a[3]) {
  const double kPi = 3.141592653589793238462643383279502884;
  double r0 = std::atan2(R[1][2], R[2][2]);
  const double c2 = std::sqrt(R[0][0] * R[0][0] + R[0][1] * R[0][1]);
  double r1;
  if (r0 > 0.0) { r0 -= kPi; r1 = std::atan2(-R[0][2], -c2); }   // (axes 0,1,2 are an even permutation: keep the first angle <= 0 before the final negation)
  else r1 = std::atan2(-R[0][2], c2);
  const double s1 = std::sin(r0), c1 = std::cos(r0);
  const double r2 = std::atan2(s1 * R[2][0] - c1 * R[1][0], c1 * R[1][1] - s1 * R[2][1]);
  a[0] = -r0; a[1] = -r1; a[2] = -r2;
}

}  // namespace lifcal_colmap

extern "C" {

int lifcal_colmap_read(const char* folder, lifcal_colmap_model** out) {
  using namespace lifcal_colmap;
  if (!folder || !out) return LIFCAL_BA_ERR_INVALID_ARG;
  *out = nullptr;
  const std::string dir(folder);
  Raw raw; std::string err; bool binary;
  // colmap::Reconstruction::Read: the binary triple if complete, else the text triple (CalibrationData.cpp:64-74)
  if (exists(dir + "/cameras.bin") && exists(dir + "/images.bin") && exists(dir + "/points3D.bin")) binary = true;
  else if (exists(dir + "/cameras.txt") && exists(dir + "/images.txt") && exists(dir + "/points3D.txt")) binary = false;
  else { g_last_error = "lifcal_colmap_read: some data files of the COLMAP model are missing in " + dir; return LIFCAL_BA_ERR_INVALID_ARG; }
  if (!(binary ? read_binary(dir, &raw, &err) : read_text(dir, &raw, &err))) { g_last_error = "lifcal_colmap_read: " + err; return LIFCAL_BA_ERR_INVALID_ARG; }
  // intrinsic orientation: camera 1, parameter count must match its model, params 0..7 (IntrinsicOrientation.cpp:51-71)
  auto ci = raw.cameras.find(1);
  if (ci == raw.cameras.end()) { g_last_error = "lifcal_colmap_read: no camera with id 1"; return LIFCAL_BA_ERR_INVALID_ARG; }
  const Camera& cam = ci->second;
  if ((int)cam.params.size() != model_num_params(cam.model) || cam.params.size() < 8) {
    g_last_error = "lifcal_colmap_read: camera 1 does not carry the 8 parameters fx fy cx cy k1 k2 p1 p2 (OPENCV model expected)";
    return LIFCAL_BA_ERR_INVALID_ARG;
  }
  if (raw.images.empty() || raw.points.empty()) { g_last_error = "lifcal_colmap_read: no images or no object points"; return LIFCAL_BA_ERR_INVALID_ARG; }
  lifcal_colmap_model* m = new (std::nothrow) lifcal_colmap_model();
  if (!m) return LIFCAL_BA_ERR_NOMEM;
  lifcal_colmap_info& I = m->info;
  I.binary = binary ? 1 : 0; I.camera_model_id = cam.model; I.width = (int32_t)cam.width; I.height = (int32_t)cam.height;
  I.fx = cam.params[0]; I.fy = cam.params[1]; I.cx = cam.params[2]; I.cy = cam.params[3];
  I.k1 = cam.params[4]; I.k2 = cam.params[5]; I.p1 = cam.params[6]; I.p2 = cam.params[7];
  I.f = (I.fx + I.fy) / 2;
  // object points, dense ids in ascending COLMAP id (duplicates: the first wins, ObjectPoints.cpp:33-44)
  std::stable_sort(raw.points.begin(), raw.points.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  std::unordered_map<uint64_t, uint32_t> dense;
  dense.reserve(raw.points.size() * 2);
  for (const auto& p : raw.points) {
    if (dense.count(p.first)) continue;
    dense[p.first] = (uint32_t)m->point_ids.size();
    m->point_ids.push_back(p.first);
    for (int k = 0; k < 3; ++k) m->pts.push_back(p.second[k]);
  }
  // frames in ascending image id (duplicates: the first wins, ExtrinsicOrientations.cpp:36-52)
  std::stable_sort(raw.images.begin(), raw.images.end(), [](const Image& a, const Image& b) { return a.id < b.id; });
  const uint64_t kInvalid = ~0ull;   // colmap::kInvalidPoint3DId: "pointID == -1" (Images.cpp:48)
  uint32_t last_id = 0; bool have_last = false;
  for (const Image& im : raw.images) {
    if (have_last && im.id == last_id) continue;
    last_id = im.id; have_last = true;
    const uint32_t f = (uint32_t)m->frame_ids.size();
    m->frame_ids.push_back((int32_t)im.id);
    const double n = std::sqrt(im.q[0] * im.q[0] + im.q[1] * im.q[1] + im.q[2] * im.q[2] + im.q[3] * im.q[3]);   // COLMAP normalises the quaternion it reads
    for (int k = 0; k < 4; ++k) m->quat.push_back(n > 0.0 ? im.q[k] / n : (k == 0 ? 1.0 : 0.0));
    for (int k = 0; k < 3; ++k) m->trans.push_back(im.t[k]);
    std::unordered_set<uint64_t> seen;
    for (const Point2& p : im.pts) {
      if (p.id == kInvalid) continue;                     // outlier
      if (!seen.insert(p.id).second) continue;            // the same 3D point twice in one image: the second one is neglected (Images.cpp:91-94)
      auto it = dense.find(p.id);
      if (it == dense.end()) {
        g_last_error = "lifcal_colmap_read: image " + std::to_string(im.id) + " references 3D point " + std::to_string(p.id) + " which is not in points3D";
        delete m; return LIFCAL_BA_ERR_OUT_OF_RANGE;
      }
      m->x.push_back(p.x); m->y.push_back(p.y); m->fr.push_back(f); m->pt.push_back(it->second);
    }
  }
  I.n_frames = (uint32_t)m->frame_ids.size(); I.n_points = (uint32_t)m->point_ids.size(); I.n_image_points = m->x.size();
  *out = m;
  return 0;
}

int lifcal_colmap_get_info(const lifcal_colmap_model* m, lifcal_colmap_info* info) {
  if (!m || !info) return LIFCAL_BA_ERR_INVALID_ARG;
  *info = m->info;
  return 0;
}

int lifcal_colmap_get_frames(const lifcal_colmap_model* m, int32_t* frame_ids, double* views, double* world_to_cam, double* quat_wxyz) {
  if (!m) return LIFCAL_BA_ERR_INVALID_ARG;
  const size_t F = m->frame_ids.size();
  for (size_t f = 0; f < F; ++f) {
    if (frame_ids) frame_ids[f] = m->frame_ids[f];
    if (quat_wxyz) for (int k = 0; k < 4; ++k) quat_wxyz[4 * f + k] = m->quat[4 * f + k];
    double R[3][3];
    lifcal_colmap::quat_to_matrix(&m->quat[4 * f], R);
    if (views) {
      lifcal_colmap::euler_xyz(R, views + 6 * f);
      for (int k = 0; k < 3; ++k) views[6 * f + 3 + k] = m->trans[3 * f + k];
    }
    if (world_to_cam) {   // column-major 4x4 [R t; 0 0 0 1]
      double* M = world_to_cam + 16 * f;
      for (int c = 0; c < 3; ++c) { for (int r = 0; r < 3; ++r) M[4 * c + r] = R[r][c]; M[4 * c + 3] = 0.0; }
      for (int r = 0; r < 3; ++r) M[12 + r] = m->trans[3 * f + r];
      M[15] = 1.0;
    }
  }
  return 0;
}

int lifcal_colmap_get_points(const lifcal_colmap_model* m, uint64_t* colmap_ids, double* pts) {
  if (!m) return LIFCAL_BA_ERR_INVALID_ARG;
  if (colmap_ids) std::copy(m->point_ids.begin(), m->point_ids.end(), colmap_ids);
  if (pts) std::copy(m->pts.begin(), m->pts.end(), pts);
  return 0;
}

int lifcal_colmap_get_image_points(const lifcal_colmap_model* m, double* x, double* y, uint32_t* fr, uint32_t* pt) {
  if (!m) return LIFCAL_BA_ERR_INVALID_ARG;
  if (x) std::copy(m->x.begin(), m->x.end(), x);
  if (y) std::copy(m->y.begin(), m->y.end(), y);
  if (fr) std::copy(m->fr.begin(), m->fr.end(), fr);
  if (pt) std::copy(m->pt.begin(), m->pt.end(), pt);
  return 0;
}

void lifcal_colmap_free(lifcal_colmap_model* m) { delete m; }

}  // extern "C"
