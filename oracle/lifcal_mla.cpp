// oracle/lifcal_mla.cpp — CPU restatement of the step that turns virtual-image points into micro-image observations
// (SURVEY.md 8f, rank f1).  TEST INFRASTRUCTURE ONLY, parity unpinned like the rest of oracle/ (see README.md): the product
// never includes, links or calls this file.  It is the sequential, line-by-line checker of the GPU path behind
// include/lifcal_mla.h (lifcal_amd/csrc/mla.hpp), compared bit for bit in tests/test_gpu_mla.py.
//
// Restated line by line, with the reference's types (float / double / int) and evaluation order:
//   MicroLensGrid::readInGrid      src/MicroLensGrid/MicroLensGrid.cpp:56-170   (derived quantities only, no XML)
//   MicroLensGrid::createGrid      src/MicroLensGrid/MicroLensGrid.cpp:186-270
//   MicroLensGrid::defineMlMaps    src/MicroLensGrid/MicroLensGrid.cpp:338-421
//   EpiPolarLine ctor / add        src/MicroLensGrid/EpiPolarLine.cpp:17-48
//   CameraCalibration::defineEpiPolarLines     src/CameraCalibration.cpp:521-632
//   CameraCalibration::projectPointsToRawImage src/CameraCalibration.cpp:637-769
// Built with -ffp-contract=off (oracle/Makefile): the reference's float expressions are not fused on its x86-64 target.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lifcal_oracle.h"

namespace {

struct MicroLens { unsigned idx, lensType; float centerX, centerY; };            // src/MicroLensGrid/MicroLens.h
struct Epl { double epiLine[2]; double baseLineDist; double minVirtualDepth; };  // src/MicroLensGrid/EpiPolarLine.h

Epl make_epl(double eplX, double eplY, double baseLineDist, float minVirtualDepth) {   // EpiPolarLine.cpp:17-33
  Epl e;
  e.epiLine[0] = eplX; e.epiLine[1] = eplY;
  double vecLength_2 = e.epiLine[0] * e.epiLine[0] + e.epiLine[1] * e.epiLine[1];
  if (vecLength_2 != 1.0f) {
    double vecLength = std::sqrt(vecLength_2);
    e.epiLine[0] /= vecLength; e.epiLine[1] /= vecLength;
  }
  e.baseLineDist = baseLineDist; e.minVirtualDepth = minVirtualDepth;
  return e;
}
Epl epl_add(const Epl& a, const Epl& other) {                                           // EpiPolarLine.cpp:40-48
  double x_new = a.epiLine[0] * a.baseLineDist + other.epiLine[0] * other.baseLineDist;
  double y_new = a.epiLine[1] * a.baseLineDist + other.epiLine[1] * other.baseLineDist;
  double baseLineDist_new = std::sqrt(x_new * x_new + y_new * y_new);
  return make_epl(x_new, y_new, baseLineDist_new, 2.0f);
}

struct Mla {
  int width = 0, height = 0;
  float imCenter[2], offset[2], offsetOpenCV[2];
  float lensDiameter = 0, rotation = 0, lensValidityRadius = 0, lensValidityRadius_2 = 0, lensBaseY[2];
  bool rotationOnGrid = true;
  std::vector<MicroLens> lenses;
  std::vector<int32_t> mapMlPointer, mapNextMl;          // lens index or -1 (the reference stores MicroLens*)
  std::vector<std::vector<Epl>> web;                     // epiLineWeb: groups of equal base-line length, ascending
};

void create_grid(Mla& g) {   // MicroLensGrid.cpp:186-270
  const bool doRotationOnGrid = g.rotationOnGrid;
  float xImMin = -g.imCenter[0] - g.offset[0] - g.lensDiameter / 2.0f;
  float xImMax = g.imCenter[0] - g.offset[0] + g.lensDiameter / 2.0f;
  float yImMin = -g.imCenter[1] - g.offset[1] - g.lensDiameter / 2.0f;
  float yImMax = g.imCenter[1] - g.offset[1] + g.lensDiameter / 2.0f;
  int xLensMinGrid1 = (int)std::ceil(xImMin / g.lensDiameter);
  int xLensMaxGrid1 = (int)(xImMax / g.lensDiameter);
  int yLensMinGrid1 = (int)std::ceil(yImMin / (2.0f * g.lensBaseY[1] * g.lensDiameter));
  int yLensMaxGrid1 = (int)(yImMax / (2.0f * g.lensBaseY[1] * g.lensDiameter));
  int xLensMinGrid2 = (int)std::ceil(xImMin / g.lensDiameter - g.lensBaseY[0] - 1.0f);
  int xLensMaxGrid2 = (int)(xImMax / g.lensDiameter - g.lensBaseY[0] - 1.0f);
  int yLensMinGrid2 = (int)std::ceil(yImMin / (2.0f * g.lensBaseY[1] * g.lensDiameter) - 0.5f);
  int yLensMaxGrid2 = (int)(yImMax / (2.0f * g.lensBaseY[1] * g.lensDiameter) - 0.5f);
  int nMicroLenses = (xLensMaxGrid1 - xLensMinGrid1 + 1) * (yLensMaxGrid1 - yLensMinGrid1 + 1);
  nMicroLenses += (xLensMaxGrid2 - xLensMinGrid2 + 1) * (yLensMaxGrid2 - yLensMinGrid2 + 1);
  g.lenses.assign((size_t)(nMicroLenses > 0 ? nMicroLenses : 0), MicroLens{0, 0, 0.0f, 0.0f});
  int lensId = 0;
  int lensType;
  float centerX = 0, centerY = 0, centerXTmp = 0, centerYTmp = 0, cosAlpha = 0, sinAlpha = 0;
  if (doRotationOnGrid) { cosAlpha = std::cos(g.rotation); sinAlpha = std::sin(g.rotation); }   // float overloads, as in the reference
  for (int x = xLensMinGrid1; x <= xLensMaxGrid1; x++) {
    lensType = x % 3; if (lensType < 0) lensType += 3;
    if (doRotationOnGrid) centerXTmp = (float)x * g.lensDiameter;
    else centerX = g.offsetOpenCV[0] + (float)x * g.lensDiameter;
    for (int y = yLensMinGrid1; y <= yLensMaxGrid1; y++, lensId++) {
      if (doRotationOnGrid) {
        centerYTmp = (float)y * g.lensDiameter * 2.0f * g.lensBaseY[1];
        centerX = g.offsetOpenCV[0] + (centerXTmp * cosAlpha - centerYTmp * sinAlpha);
        centerY = g.offsetOpenCV[1] - (centerXTmp * sinAlpha + centerYTmp * cosAlpha);
      } else centerY = g.offsetOpenCV[1] - (float)y * g.lensDiameter * 2.0f * g.lensBaseY[1];
      g.lenses[lensId] = MicroLens{(unsigned)lensId, (unsigned)lensType, centerX, centerY};
    }
  }
  for (int x = xLensMinGrid2; x <= xLensMaxGrid2; x++) {
    lensType = x % 3; if (lensType < 0) lensType += 3;
    if (doRotationOnGrid) centerXTmp = ((float)x + 1.0f + g.lensBaseY[0]) * g.lensDiameter;
    else centerX = g.offsetOpenCV[0] + ((float)x + 1.0f + g.lensBaseY[0]) * g.lensDiameter;
    for (int y = yLensMinGrid2; y <= yLensMaxGrid2; y++, lensId++) {
      if (doRotationOnGrid) {
        centerYTmp = (((float)y * 2.0f + 1.0f) * g.lensBaseY[1]) * g.lensDiameter;
        centerX = g.offsetOpenCV[0] + (centerXTmp * cosAlpha - centerYTmp * sinAlpha);
        centerY = g.offsetOpenCV[1] - (centerXTmp * sinAlpha + centerYTmp * cosAlpha);
      } else centerY = g.offsetOpenCV[1] - (((float)y * 2.0f + 1.0f) * g.lensBaseY[1]) * g.lensDiameter;
      g.lenses[lensId] = MicroLens{(unsigned)lensId, (unsigned)lensType, centerX, centerY};
    }
  }
}

void define_ml_maps(Mla& g) {   // MicroLensGrid.cpp:338-421
  const int width = g.width, height = g.height;
  g.mapMlPointer.assign((size_t)width * height, -1);
  g.mapNextMl.assign((size_t)width * height, -1);
  for (int i = 0; i < (int)g.lenses.size(); i++) {
    double centerX = g.lenses[i].centerX;
    double centerY = g.lenses[i].centerY;
    for (int y = (int)std::ceil(centerY - g.lensValidityRadius); y <= centerY + g.lensValidityRadius; y++) {
      if (y < 0 || y >= height) continue;
      double y_2 = (y - centerY) * (y - centerY);
      if (!(g.lensValidityRadius_2 - y_2 >= 0)) continue;   // the reference's loop condition is false for every x in this case
      for (int x = (int)std::ceil(centerX - std::sqrt(g.lensValidityRadius_2 - y_2)); (x - centerX) * (x - centerX) <= g.lensValidityRadius_2 - y_2; x++) {
        if (x < 0 || x >= width) continue;
        int idx = x + y * width;
        g.mapMlPointer[idx] = i;
        g.mapNextMl[idx] = i;
      }
    }
  }
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      int idx = x + y * width;
      if (g.mapNextMl[idx] != -1) continue;
      float dist_2 = -1;
      for (int d = 1;; d++) {
        for (int dx = -d; dx <= d; dx++) {
          if (x + dx < 0) continue;
          if (x + dx >= width) break;
          for (int dy = -d; dy <= d; dy++) {
            if (dx != -d && dx != d && dy != -d && dy != d) continue;
            if (y + dy < 0) continue;
            if (y + dy >= height) break;
            int32_t pML_new = g.mapMlPointer[idx + dx + dy * width];
            if (pML_new != -1) {
              float centerX = g.lenses[pML_new].centerX;
              float centerY = g.lenses[pML_new].centerY;
              float dist_2_new = (centerX - x) * (centerX - x) + (centerY - y) * (centerY - y);
              if (dist_2_new < dist_2 || dist_2 < 0) { g.mapNextMl[idx] = pML_new; dist_2 = dist_2_new; }
            }
          }
        }
        if (g.mapNextMl[idx] != -1) break;
        if (d > width + height) break;   // (not in the reference: an image without any lens would loop forever there)
      }
    }
  }
}

void define_epipolar_lines(Mla& g) {   // CameraCalibration.cpp:521-632
  float maxDist = g.lensDiameter * 10;
  Epl epl_0 = make_epl(1, 0, g.lensDiameter, 2.0f);
  Epl epl_1 = make_epl(0.5, std::sqrt(0.75), g.lensDiameter, 2.0f);
  Epl _epl_1 = make_epl(-0.5, -std::sqrt(0.75), g.lensDiameter, 2.0f);
  Epl epl_2 = make_epl(0.5, -std::sqrt(0.75), g.lensDiameter, 2.0f);
  Epl _epl_2 = make_epl(-0.5, std::sqrt(0.75), g.lensDiameter, 2.0f);
  if (g.rotationOnGrid) {
    // `double cosAlpha = cos(mlGrid->rotation)`: the argument is a float, so <math.h> under C++ picks the float overload
    // and the float result is widened.
    double cosAlpha = (double)std::cos(g.rotation);
    double sinAlpha = (double)std::sin(g.rotation);
    for (Epl* e : {&epl_0, &epl_1, &_epl_1, &epl_2, &_epl_2}) {
      double epx = e->epiLine[0], epy = e->epiLine[1];
      e->epiLine[0] = epx * cosAlpha + epy * sinAlpha;
      e->epiLine[1] = -epx * sinAlpha + epy * cosAlpha;
    }
  }
  int i = 0;
  std::vector<Epl> epi_Lines;
  epi_Lines.push_back(epl_1);
  epi_Lines.push_back(epl_2);
  while (epi_Lines.back().baseLineDist < maxDist) {
    Epl n1, n2;
    if (i % 2 == 0) { n1 = epl_add(epi_Lines[i * 2], _epl_2); n2 = epl_add(epi_Lines[i * 2 + 1], _epl_1); }
    else { n1 = epl_add(epi_Lines[i * 2], epl_1); n2 = epl_add(epi_Lines[i * 2 + 1], epl_2); }
    epi_Lines.push_back(n1);
    epi_Lines.push_back(n2);
    i++;
  }
  epi_Lines.push_back(epl_0);
  int initLenght = (int)epi_Lines.size();
  for (int k = 0; k < initLenght; k++) {
    Epl last = epi_Lines[k];
    while (last.baseLineDist < maxDist) { epi_Lines.push_back(epl_add(last, epl_0)); last = epi_Lines.back(); }
  }
  g.web.clear();
  g.web.push_back(std::vector<Epl>{epi_Lines[0]});
  for (size_t k = 1; k < epi_Lines.size(); k++) {
    if ((epi_Lines[k].epiLine[1] == -1.0f) || (epi_Lines[k].baseLineDist > maxDist)) continue;
    bool isSmaller = false, isEqual = false;
    size_t ii;
    for (ii = 0; (ii < g.web.size()) && (!(isSmaller || isEqual)); ii++) {
      if (((float)g.web[ii][0].baseLineDist) == ((float)epi_Lines[k].baseLineDist)) isEqual = true;
      else if (g.web[ii][0].baseLineDist > epi_Lines[k].baseLineDist) isSmaller = true;
    }
    if (isSmaller) g.web.insert(g.web.begin() + (ii - 1), std::vector<Epl>{epi_Lines[k]});
    else if (isEqual) g.web[ii - 1].push_back(epi_Lines[k]);
    else g.web.push_back(std::vector<Epl>{epi_Lines[k]});
  }
}

}  // namespace

extern "C" {


void* lo_mla_create(const lo_mla_params* p) {
  if (!p || p->width <= 0 || p->height <= 0) return nullptr;
  Mla* g = new Mla;
  g->width = p->width; g->height = p->height;                                   // MicroLensGrid.cpp:60-63
  g->imCenter[0] = ((float)g->width) / 2.0f - 0.5f; g->imCenter[1] = ((float)g->height) / 2.0f - 0.5f;
  g->rotationOnGrid = p->rotation_on_grid != 0;
  g->offset[0] = p->offset[0]; g->offset[1] = p->offset[1];
  g->lensDiameter = p->lens_diameter; g->rotation = p->rotation;
  float lensBorder = 1.0f;                                                       // :107 (the XML value is overridden)
  g->lensValidityRadius = g->lensDiameter * 0.5f - lensBorder;                   // :110-111
  g->lensValidityRadius_2 = g->lensValidityRadius * g->lensValidityRadius;
  g->lensBaseY[0] = p->lens_base_y[0]; g->lensBaseY[1] = p->lens_base_y[1];
  g->offsetOpenCV[0] = g->offset[0] + g->imCenter[0];                            // :165-166
  g->offsetOpenCV[1] = -g->offset[1] + g->imCenter[1];
  create_grid(*g);
  define_ml_maps(*g);
  define_epipolar_lines(*g);
  return g;
}
void lo_mla_destroy(void* h) { delete (Mla*)h; }

int lo_mla_counts(void* h, int32_t* n_lenses, int32_t* n_web_groups, int32_t* n_web_lines) {
  Mla* g = (Mla*)h; if (!g) return -1;
  *n_lenses = (int32_t)g->lenses.size(); *n_web_groups = (int32_t)g->web.size();
  int n = 0; for (auto& v : g->web) n += (int)v.size();
  *n_web_lines = n;
  return 0;
}
int lo_mla_lenses(void* h, float* cx, float* cy, int32_t* type) {
  Mla* g = (Mla*)h; if (!g) return -1;
  for (size_t i = 0; i < g->lenses.size(); ++i) { cx[i] = g->lenses[i].centerX; cy[i] = g->lenses[i].centerY; type[i] = (int32_t)g->lenses[i].lensType; }
  return 0;
}
int lo_mla_maps(void* h, int32_t* map_ml, int32_t* map_next) {
  Mla* g = (Mla*)h; if (!g) return -1;
  std::memcpy(map_ml, g->mapMlPointer.data(), g->mapMlPointer.size() * sizeof(int32_t));
  std::memcpy(map_next, g->mapNextMl.data(), g->mapNextMl.size() * sizeof(int32_t));
  return 0;
}
int lo_mla_web(void* h, double* dist, double* ex, double* ey, int32_t* group) {
  Mla* g = (Mla*)h; if (!g) return -1;
  size_t k = 0;
  for (size_t gi = 0; gi < g->web.size(); ++gi)
    for (const Epl& e : g->web[gi]) { dist[k] = e.baseLineDist; ex[k] = e.epiLine[0]; ey[k] = e.epiLine[1]; group[k] = (int32_t)gi; ++k; }
  return 0;
}

// CameraCalibration::projectPointsToRawImage for ONE frame (src/CameraCalibration.cpp:642-762): image points (x, y) of the
// virtual image (frames[i].imageCoordinates, doubles holding the values), their virtual depths; outputs in push order.
// Returns the number of observations written (at most `capacity`), or -(needed) if capacity is too small.
int64_t lo_mla_project_frame(void* h, int32_t depth_to_raw_im_scale, int64_t n, const double* px, const double* py, const double* vd,
                             int64_t capacity, double* xR_out, double* yR_out, double* cX_out, double* cY_out, int64_t* point_out) {
  Mla* g = (Mla*)h; if (!g) return 0;
  const int rawWidth = g->width, rawHeight = g->height;
  int64_t m = 0;
  std::vector<int32_t> microLenses;
  for (int64_t p = 0; p < n; p++) {
    float vdepth = (float)vd[p];
    if (!(vdepth > 2.0 && vdepth < 20.0)) continue;
    float x = (float)px[p];
    float y = (float)py[p];
    float radius = g->lensDiameter * 0.5f * vdepth + 2.0f;
    float radius_2 = radius * radius;
    float xUps_flt = ((float)depth_to_raw_im_scale) * (x + 0.5f) - 0.5f;
    float yUps_flt = ((float)depth_to_raw_im_scale) * (y + 0.5f) - 0.5f;
    // (not in the reference: coordinates a float cannot carry into an int are outside its contract; no observation)
    if (!(xUps_flt > -1.0e9f && xUps_flt < 1.0e9f && yUps_flt > -1.0e9f && yUps_flt < 1.0e9f)) continue;
    int xUps_int = (int)(xUps_flt + 0.5f);
    if (xUps_int >= rawWidth) xUps_int = rawWidth - 1;
    int yUps_int = (int)(yUps_flt + 0.5f);
    if (yUps_int >= rawHeight) yUps_int = rawHeight - 1;
    if (xUps_int < 0 || yUps_int < 0) continue;   // (the reference indexes the map unchecked here; negative coordinates are out of its contract)
    int idxUps = xUps_int + rawWidth * yUps_int;
    int32_t ml = g->mapNextMl[idxUps];
    if (ml == -1) continue;
    float centerX_next = g->lenses[ml].centerX;
    float centerY_next = g->lenses[ml].centerY;
    float distCenterX_next = centerX_next - xUps_flt;
    float distCenterY_next = centerY_next - yUps_flt;
    float distCenter_2_next = distCenterX_next * distCenterX_next + distCenterY_next * distCenterY_next;
    if (distCenter_2_next > radius_2) continue;
    microLenses.clear();
    microLenses.push_back(ml);
    for (const auto& grp : g->web) {
      if (grp[0].baseLineDist > radius) break;
      for (const Epl& base : grp) {
        for (int iEpl = 0; iEpl < 2; iEpl++) {
          float baseLineDist = (float)base.baseLineDist;
          float epx, epy;
          if (iEpl == 0) { epx = (float)base.epiLine[0]; epy = (float)base.epiLine[1]; }
          else { epx = (float)(-base.epiLine[0]); epy = (float)(-base.epiLine[1]); }
          float centerX = centerX_next + baseLineDist * epx;
          float centerY = centerY_next + baseLineDist * epy;
          float distCenterX = centerX - xUps_flt;
          float distCenterY = centerY - yUps_flt;
          float distCenter_2 = distCenterX * distCenterX + distCenterY * distCenterY;
          if (distCenter_2 > radius_2) continue;
          int centerX_int = (int)(centerX + 0.5);
          int centerY_int = (int)(centerY + 0.5);
          if (centerX_int < 0) centerX_int = 0;
          if (centerX_int >= rawWidth) centerX_int = rawWidth - 1;
          if (centerY_int < 0) centerY_int = 0;
          if (centerY_int >= rawHeight) centerY_int = rawHeight - 1;
          int idxMl = centerX_int + centerY_int * rawWidth;
          int32_t mlc = g->mapMlPointer[idxMl];
          if (mlc == -1) continue;
          microLenses.push_back(mlc);
        }
      }
    }
    for (size_t iMl = 0; iMl < microLenses.size(); iMl++) {
      float centerX = g->lenses[microLenses[iMl]].centerX;
      float centerY = g->lenses[microLenses[iMl]].centerY;
      float xR = (xUps_flt - centerX) / vdepth + centerX;
      float yR = (yUps_flt - centerY) / vdepth + centerY;
      if (!(xR >= 0 && xR <= rawWidth - 1 && yR >= 0 && yR <= rawHeight - 1)) continue;
      float distToCenterX = xR - centerX;
      float distToCenterY = yR - centerY;
      float distToCenter_2 = distToCenterX * distToCenterX + distToCenterY * distToCenterY;
      if (distToCenter_2 >= g->lensValidityRadius_2) continue;
      if (m < capacity) { xR_out[m] = xR; yR_out[m] = yR; cX_out[m] = centerX; cY_out[m] = centerY; point_out[m] = p; }
      ++m;
    }
  }
  return m <= capacity ? m : -m;
}

}  // extern "C"
