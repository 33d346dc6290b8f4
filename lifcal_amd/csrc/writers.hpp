// writers.hpp — LiFCal's result files (include/lifcal_io.h).  Host code only; included at the end of lifcal_ba.hip.
// The two XML files follow pugixml's default output (declaration, tab indentation, text-only elements on one line) and
// boost::lexical_cast<std::string>(double), which prints 17 significant digits in %g style.
#pragma once
#include <algorithm>
#include <cstdio>
#include <string>

#include "../../include/lifcal_io.h"

namespace lifcal_io {

inline std::string num(double v) { char b[64]; std::snprintf(b, sizeof b, "%.17g", v); return b; }   // boost::lexical_cast<std::string>(double)

struct Xml {   // the subset of pugixml's writer these files need
  std::string out = "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n";
  int depth = 0;
  void indent() { out.append((size_t)depth, '\t'); }
  void open(const std::string& tag, const std::string& attrs = "") { indent(); out += "<" + tag + attrs + ">\n"; ++depth; }
  void close(const std::string& tag) { --depth; indent(); out += "</" + tag + ">\n"; }
  void leaf(const std::string& tag, const std::string& text, const std::string& attrs = "") { indent(); out += "<" + tag + attrs + ">" + text + "</" + tag + ">\n"; }
  int save(const char* path) const {
    FILE* f = std::fopen(path, "wb");
    if (!f) return LIFCAL_BA_ERR_INVALID_ARG;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return (std::fclose(f) == 0 && ok) ? 0 : LIFCAL_BA_ERR_INVALID_ARG;
  }
};

// RigidBody::getTransformationMatrix (src/CameraModel.h:246-264): R = Rx(a0) Ry(a1) Rz(a2)
inline void rigid_matrix(const double* a, const double* t, double m[4][4]) {
  const double cx = std::cos(a[0]), sx = std::sin(a[0]), cy = std::cos(a[1]), sy = std::sin(a[1]), cz = std::cos(a[2]), sz = std::sin(a[2]);
  const double R[3][3] = {{cy * cz, -cy * sz, sy},
                          {sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy},
                          {-cx * sy * cz + sx * sz, cx * sy * sz + sx * cz, cx * cy}};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = i == j ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m[i][j] = R[i][j]; m[i][3] = t[i]; }
}

}  // namespace lifcal_io

extern "C" {

int lifcal_write_camera_model(const char* path, const lifcal_camera_model* m) {
  if (!path || !m || m->n_radial < 0 || m->n_radial > 8) return LIFCAL_BA_ERR_INVALID_ARG;
  using lifcal_io::num;
  lifcal_io::Xml x;
  x.open("Root");
  x.leaf("CalibrationModel", "Plenoptic");
  x.open("ImageSize", " units=\"pix\"");
  x.leaf("Width", std::to_string(m->image_width));
  x.leaf("Height", std::to_string(m->image_height));
  x.close("ImageSize");
  { char b[64]; std::snprintf(b, sizeof b, "%.5f", m->pixel_size); x.leaf("PixelSize", b, " units=\"mm\""); }
  x.open("PrincipalPoint", " units=\"pix\"");
  x.leaf("x", num(m->cx)); x.leaf("y", num(m->cy));
  x.close("PrincipalPoint");
  x.leaf("FocalLength", num(m->fL), " units=\"mm\"");
  x.leaf("MainLensMlaDistance", num(m->bL0), " units=\"mm\"");
  x.leaf("SensorMlaDistance", num(m->B), " units=\"mm\"");
  if (m->n_radial > 0) {
    x.open("RadialDistortion", " units=\"mm\"");
    for (int i = 0; i < m->n_radial; ++i) x.leaf("A" + std::to_string(i), num(m->radial[i]));
    x.close("RadialDistortion");
  }
  if (m->tangential) {
    x.open("TangentialDistortion", " units=\"mm\"");
    x.leaf("B0", num(m->tangential_dist[0])); x.leaf("B1", num(m->tangential_dist[1]));
    x.close("TangentialDistortion");
  }
  x.leaf("MicroLensCenterAdjustment", m->ml_center_adjustment ? "true" : "false");
  x.close("Root");
  return x.save(path);
}

int lifcal_write_extrinsic_orientations_xml(const char* path, uint32_t n_frames, const int32_t* frame_ids, const double* views) {
  if (!path || (n_frames && (!frame_ids || !views))) return LIFCAL_BA_ERR_INVALID_ARG;
  lifcal_io::Xml x;
  if (n_frames == 0) { x.out += "<Root />\n"; return x.save(path); }
  x.open("Root");
  for (uint32_t f = 0; f < n_frames; ++f) {
    x.open("Frame", " id=\"" + std::to_string(frame_ids[f]) + "\"");
    for (int part = 0; part < 2; ++part) {
      const char* tag = part == 0 ? "Rotation" : "Translation";
      x.open(tag);
      for (int i = 0; i < 3; ++i) x.leaf("Coeff", lifcal_io::num(views[6 * (size_t)f + 3 * part + i]), " i=\"" + std::to_string(i) + "\"");
      x.close(tag);
    }
    x.close("Frame");
  }
  x.close("Root");
  return x.save(path);
}

int lifcal_write_extrinsic_orientations_txt(const char* path, uint32_t n_frames, const int32_t* frame_ids, const double* views) {
  if (!path || (n_frames && (!frame_ids || !views))) return LIFCAL_BA_ERR_INVALID_ARG;
  std::vector<uint32_t> idx(n_frames);
  for (uint32_t i = 0; i < n_frames; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return frame_ids[a] < frame_ids[b]; });   // std::sort, as the reference (:1456)
  FILE* f = std::fopen(path, "w+");
  if (!f) return LIFCAL_BA_ERR_INVALID_ARG;
  for (uint32_t k = 0; k < n_frames; ++k) {
    const uint32_t i = idx[k];
    double m[4][4];
    lifcal_io::rigid_matrix(views + 6 * (size_t)i, views + 6 * (size_t)i + 3, m);
    std::fprintf(f, "%05d", frame_ids[i]);
    for (int y = 0; y < 4; ++y) for (int c = 0; c < 4; ++c) std::fprintf(f, " %16.10f", m[y][c]);
    std::fprintf(f, "\n");
  }
  return std::fclose(f) == 0 ? 0 : LIFCAL_BA_ERR_INVALID_ARG;
}

int lifcal_write_raw_image_points_csv(const char* path, uint64_t n_obs, uint32_t n_frames, const int32_t* frame_ids, const uint32_t* fr, const double* u,
                                      const double* v, const double* x_proj, const double* y_proj, const uint32_t* pt) {
  if (!path || (n_obs && (!frame_ids || !fr || !u || !v || !x_proj || !y_proj || !pt))) return LIFCAL_BA_ERR_INVALID_ARG;
  for (uint64_t k = 0; k < n_obs; ++k) if (fr[k] >= n_frames) { g_last_error = "lifcal_write_raw_image_points_csv: frame index out of range"; return LIFCAL_BA_ERR_OUT_OF_RANGE; }
  for (uint64_t k = 1; k < n_obs; ++k) if (fr[k] < fr[k - 1]) { g_last_error = "lifcal_write_raw_image_points_csv: observations are not in frame order"; return LIFCAL_BA_ERR_INVALID_ARG; }
  FILE* f = std::fopen(path, "w");
  if (!f) return LIFCAL_BA_ERR_INVALID_ARG;
  uint64_t first = 0;
  for (uint64_t k = 0; k < n_obs; ++k) {
    if (k && fr[k] != fr[k - 1]) first = k;
    std::fprintf(f, "%d,%d,%f,%f,%f,%f,%d\n", frame_ids[fr[k]], (int)(k - first), u[k], v[k], x_proj[k], y_proj[k], (int)pt[k]);
  }
  return std::fclose(f) == 0 ? 0 : LIFCAL_BA_ERR_INVALID_ARG;
}

int lifcal_write_protocol(const char* path, const lifcal_protocol* p) {
  if (!path || !p || p->model.n_radial < 0 || p->model.n_radial > 8) return LIFCAL_BA_ERR_INVALID_ARG;
  FILE* f = std::fopen(path, "w+");
  if (!f) return LIFCAL_BA_ERR_INVALID_ARG;
  const lifcal_camera_model& m = p->model;
  std::fprintf(f,
               "*******************************************************************************\n"
               "***   LiFCal: Online Light Field Camera Calibration via Bundle Adjustment   ***\n"
               "*******************************************************************************\n\n");
  std::fprintf(f, "*** Intrinsic Parameters ***\n");
  std::fprintf(f, "Pixel Size: %1.3f mm\n", m.pixel_size);
  std::fprintf(f, "\tfL   : %18.15f\n", m.fL);
  std::fprintf(f, "\tbL0  : %18.15f\n", m.bL0);
  std::fprintf(f, "\tB    : %18.15f\n", m.B);
  std::fprintf(f, "\tcx   : %18.15f\n", m.cx);
  std::fprintf(f, "\tcy   : %18.15f\n", m.cy);
  for (int i = 0; i < m.n_radial; ++i) std::fprintf(f, "\ta%d   : %18.15f\n", i, m.radial[i]);
  if (m.tangential) { std::fprintf(f, "\tb0   : %18.15f\n", m.tangential_dist[0]); std::fprintf(f, "\tb1   : %18.15f\n", m.tangential_dist[1]); }
  std::fprintf(f, "\n");
  if (m.ml_center_adjustment) std::fprintf(f, "\tDid micro lens center adjustment\n");
  std::fprintf(f, "*** Additional Settings ***\n\tDistortion defined on MLA plane.\n\n");
  std::fprintf(f, p->refine_poses ? "\tExtrinsic Orientations were refined.\n\n" : "\tExtrinsic Orientations from COLMAP were kept.\n\n");
  std::fprintf(f, p->refine_points ? "\t3D Object coordinates were refined.\n\n" : "\t3D Object coordinates from COLMAP were kept.\n\n");
  std::fprintf(f, p->robust_cost ? "\tRobust cost function was used for estimation.\n\n" : "\tSquared cost function was used for estimation.\n\n");
  std::fprintf(f, "*** Statistics ***\n\tReprojection errors:\n");
  std::fprintf(f, "\tstd. Dev. x:           %8.5f\n", p->std_x);
  std::fprintf(f, "\tstd. Dev. y:           %8.5f\n", p->std_y);
  std::fprintf(f, "\tmae x:                 %8.5f\n", p->mae_x);
  std::fprintf(f, "\tmae y:                 %8.5f\n", p->mae_y);
  return std::fclose(f) == 0 ? 0 : LIFCAL_BA_ERR_INVALID_ARG;
}

}  // extern "C"
