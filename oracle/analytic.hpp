// oracle/analytic.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Second CPU arm of the residual-block evaluation: the same residual as oracle/model.hpp (reference
// src/BundleAdjustment/BundleAdjustment.h:120-195 over src/CameraModel.h:86-264) with a hand-derived ANALYTIC Jacobian
//  instead of dual numbers, and with what depends on the camera only / on the micro lens only / on the
// frame only computed once per sweep instead of once per observation:
//   acam_prepare   sign folding of fL, bL0, B and c_raw (BundleAdjustment.h:123-133)
//   alens_eval     the 10-sweep undistortion of the micro-lens centre (CameraModel.h:92-125) and the tangents of that
//                  truncated iteration w.r.t. (c_raw, k, p) — what autodiff propagates through the loop
//   aframe_eval    R = Rx Ry Rz (CameraModel.h:246-264) and dR/d(angle)
//   aobs_eval      the closed-form rest of projectPoint (CameraModel.h:127-195)
// Purpose: (1) the "analytic Jacobian" arm of bench.py's CPU baseline (both arms are timed), (2) one more
// cross-check of the dual-number oracle (tests/test_oracle_analytic.py: both arms agree to round-off).
// Written from the model equations, independently of lifcal_amd/csrc/device_model.hpp.  PARITY UNPINNED like the rest of oracle/.
#pragma once
#include <cmath>

#include "model.hpp"

namespace lo {

struct ACam {
  int nr; bool tan, adj;
  double fL, bL0, B, cr[2], sp[2], k[2], p[2];
  double ch[9];                       // d(model parameter j) / d(camera slot j): -1 where the stored value is negative, +-scale for c
  double D, e, zC0, gamma, beta, a;   // fL - bL0, fL/D, fL bL0/D, fL B/D, B/D, bL0/(bL0+B)
  double de[3], dzC0[3], dgamma[3], dbeta[3], da[3];   // partials w.r.t. (fL, bL0, B)
};

inline void acam_prepare(const double* cam, const Config& cfg, double spx_in, double spy_in, double scale, ACam& c) {
  c.nr = cfg.n_radial; c.tan = cfg.tangential; c.adj = cfg.ml_center_adj;
  double th[3];
  for (int i = 0; i < 3; ++i) { th[i] = cam[i]; c.ch[i] = 1.0; if (th[i] < 0.0) { th[i] = -th[i]; c.ch[i] = -1.0; } }
  c.fL = th[0]; c.bL0 = th[1]; c.B = th[2];
  for (int i = 0; i < 2; ++i) {
    double v = (cam[3 + i] + 0.5) * scale - 0.5; c.ch[3 + i] = scale;
    if (v < 0.0) { v = -v; c.ch[3 + i] = -scale; }
    c.cr[i] = v;
  }
  for (int j = 5; j < 9; ++j) c.ch[j] = 1.0;
  c.sp[0] = spx_in / scale; c.sp[1] = spy_in / scale;
  c.k[0] = c.nr > 0 ? cam[5] : 0.0; c.k[1] = c.nr > 1 ? cam[6] : 0.0;
  c.p[0] = c.tan ? cam[5 + c.nr] : 0.0; c.p[1] = c.tan ? cam[6 + c.nr] : 0.0;
  const double D = c.fL - c.bL0, D2 = D * D;
  c.D = D; c.e = c.fL / D; c.zC0 = c.fL * c.bL0 / D; c.gamma = c.fL * c.B / D; c.beta = c.B / D;
  c.de[0] = -c.bL0 / D2;             c.de[1] = c.fL / D2;            c.de[2] = 0.0;
  c.dzC0[0] = -c.bL0 * c.bL0 / D2;   c.dzC0[1] = c.fL * c.fL / D2;   c.dzC0[2] = 0.0;
  c.dgamma[0] = -c.B * c.bL0 / D2;   c.dgamma[1] = c.fL * c.B / D2;  c.dgamma[2] = c.fL / D;
  c.dbeta[0] = -c.B / D2;            c.dbeta[1] = c.B / D2;          c.dbeta[2] = 1.0 / D;
  const double s = c.bL0 + c.B;
  c.a = c.bL0 / s; c.da[0] = 0.0; c.da[1] = c.B / (s * s); c.da[2] = -c.bL0 / (s * s);
}

// Delta = radial + tangential distortion at (x, y) (CameraModel.h:205-241), A = dDelta/d(x,y), explicit partials w.r.t. k_i, p_i
struct ADist { double dx, dy, a00, a01, a10, a11, ek[2][2], ep[2][2]; };

inline void adist(const ACam& c, double x, double y, ADist& d) {
  const double r2 = x * x + y * y, r4 = r2 * r2;
  double g = 0.0, gp = 0.0;
  if (c.nr >= 1) { g = c.k[0] * r2; gp = c.k[0]; }
  if (c.nr >= 2) { g += c.k[1] * r4; gp += 2.0 * c.k[1] * r2; }
  d.dx = x * g; d.dy = y * g;
  d.a00 = g + 2.0 * x * x * gp; d.a01 = 2.0 * x * y * gp; d.a10 = d.a01; d.a11 = g + 2.0 * y * y * gp;
  d.ek[0][0] = x * r2; d.ek[0][1] = y * r2; d.ek[1][0] = x * r4; d.ek[1][1] = y * r4;
  d.ep[0][0] = d.ep[0][1] = d.ep[1][0] = d.ep[1][1] = 0.0;
  if (c.tan) {
    d.dx += c.p[0] * (r2 + 2.0 * x * x) + 2.0 * c.p[1] * x * y;
    d.dy += c.p[1] * (r2 + 2.0 * y * y) + 2.0 * c.p[0] * x * y;
    d.a00 += 6.0 * c.p[0] * x + 2.0 * c.p[1] * y; d.a01 += 2.0 * c.p[0] * y + 2.0 * c.p[1] * x;
    d.a10 += 2.0 * c.p[1] * x + 2.0 * c.p[0] * y; d.a11 += 6.0 * c.p[1] * y + 2.0 * c.p[0] * x;
    d.ep[0][0] = r2 + 2.0 * x * x; d.ep[0][1] = 2.0 * x * y;
    d.ep[1][0] = 2.0 * x * y;      d.ep[1][1] = r2 + 2.0 * y * y;
  }
}

// lens parameters in this order: c_raw.x, c_raw.y, k_1, k_2, p_1, p_2 (inactive ones keep zero tangents)
struct ALens { double cd[2], cu[2], t[6][2]; };

inline void alens_eval(const ACam& c, double mx, double my, bool want_tangents, ALens& L) {
  L.cd[0] = (mx - c.cr[0]) * c.sp[0]; L.cd[1] = (my - c.cr[1]) * c.sp[1];
  double x = L.cd[0], y = L.cd[1];
  double d0[6][2];
  for (int a = 0; a < 6; ++a) { d0[a][0] = d0[a][1] = 0.0; }
  d0[0][0] = -c.sp[0]; d0[1][1] = -c.sp[1];
  for (int a = 0; a < 6; ++a) { L.t[a][0] = d0[a][0]; L.t[a][1] = d0[a][1]; }
  if (c.nr > 0 || c.tan) {
    ADist d;
    for (int it = 0; it < 10; ++it) {   // CameraModel.h:109-124
      adist(c, x, y, d);
      if (want_tangents)
        for (int a = 0; a < 6; ++a) {
          double ex = 0.0, ey = 0.0;
          if (a == 2 && c.nr >= 1) { ex = d.ek[0][0]; ey = d.ek[0][1]; }
          if (a == 3 && c.nr >= 2) { ex = d.ek[1][0]; ey = d.ek[1][1]; }
          if (a == 4 && c.tan) { ex = d.ep[0][0]; ey = d.ep[0][1]; }
          if (a == 5 && c.tan) { ex = d.ep[1][0]; ey = d.ep[1][1]; }
          const double nx = d0[a][0] - (d.a00 * L.t[a][0] + d.a01 * L.t[a][1]) - ex;
          const double ny = d0[a][1] - (d.a10 * L.t[a][0] + d.a11 * L.t[a][1]) - ey;
          L.t[a][0] = nx; L.t[a][1] = ny;
        }
      x = L.cd[0] - d.dx; y = L.cd[1] - d.dy;
    }
  }
  L.cu[0] = x; L.cu[1] = y;
}

struct AFrame { double R[3][3], dR[3][3][3], t[3]; };

inline void aframe_eval(const double* view, AFrame& f) {
  const double s0 = std::sin(view[0]), c0 = std::cos(view[0]), s1 = std::sin(view[1]), c1 = std::cos(view[1]);
  const double s2 = std::sin(view[2]), c2 = std::cos(view[2]);
  double (&R)[3][3] = f.R;
  R[0][0] = c1 * c2;                 R[0][1] = -c1 * s2;                R[0][2] = s1;
  R[1][0] = c0 * s2 + s0 * s1 * c2;  R[1][1] = c0 * c2 - s0 * s1 * s2;  R[1][2] = -s0 * c1;
  R[2][0] = s0 * s2 - c0 * s1 * c2;  R[2][1] = s0 * c2 + c0 * s1 * s2;  R[2][2] = c0 * c1;
  double (&A)[3][3] = f.dR[0];
  A[0][0] = 0.0;                      A[0][1] = 0.0;                      A[0][2] = 0.0;
  A[1][0] = -s0 * s2 + c0 * s1 * c2;  A[1][1] = -s0 * c2 - c0 * s1 * s2;  A[1][2] = -c0 * c1;
  A[2][0] = c0 * s2 + s0 * s1 * c2;   A[2][1] = c0 * c2 - s0 * s1 * s2;   A[2][2] = -s0 * c1;
  double (&Bm)[3][3] = f.dR[1];
  Bm[0][0] = -s1 * c2;       Bm[0][1] = s1 * s2;        Bm[0][2] = c1;
  Bm[1][0] = s0 * c1 * c2;   Bm[1][1] = -s0 * c1 * s2;  Bm[1][2] = s0 * s1;
  Bm[2][0] = -c0 * c1 * c2;  Bm[2][1] = c0 * c1 * s2;   Bm[2][2] = -c0 * s1;
  double (&C)[3][3] = f.dR[2];
  C[0][0] = -c1 * s2;                C[0][1] = -c1 * c2;                 C[0][2] = 0.0;
  C[1][0] = c0 * c2 - s0 * s1 * s2;  C[1][1] = -c0 * s2 - s0 * s1 * c2;  C[1][2] = 0.0;
  C[2][0] = s0 * c2 + c0 * s1 * s2;  C[2][1] = -s0 * s2 + c0 * s1 * c2;  C[2][2] = 0.0;
  f.t[0] = view[3]; f.t[1] = view[4]; f.t[2] = view[5];
}

// residual r[2], Jpc = dr/d(camera-frame point) [2][3], Jth = dr/d(camera slot) [2][17] (slots >= 5 + nr + 2 tan stay 0)
inline void aobs_eval(const ACam& c, const ALens& L, const double pc[3], double u, double v, double r[2], double Jpc[2][3], double Jth[2][17]) {
  const int nlive = 5 + c.nr + (c.tan ? 2 : 0);
  const double wsc = c.adj ? c.a : 1.0;
  const double w[2] = {L.cu[0] * wsc, L.cu[1] * wsc};
  const double Zq = pc[2] + c.zC0, iZq = 1.0 / Zq;
  const double q[2] = {(pc[0] + w[0] * c.e) * iZq, (pc[1] + w[1] * c.e) * iZq};
  double proj[2] = {c.gamma * q[0] - c.beta * w[0], c.gamma * q[1] - c.beta * w[1]};   // pMl (CameraModel.h:146-148)
  // d proj0 / d model parameter (9 columns: fL, bL0, B, c_raw.x, c_raw.y, k1, k2, p1, p2) and / d pc
  double dth[2][9], dpc[2][3];
  for (int i = 0; i < 3; ++i) {
    for (int a = 0; a < 2; ++a) {
      const double dw = c.adj ? c.da[i] * L.cu[a] : 0.0;
      const double dq = (dw * c.e + w[a] * c.de[i]) * iZq - q[a] * c.dzC0[i] * iZq;
      dth[a][i] = c.dgamma[i] * q[a] + c.gamma * dq - c.dbeta[i] * w[a] - c.beta * dw + (c.adj ? dw : 0.0);
    }
  }
  // lens parameters: they act through w = (a) c_u only (and through c_d = (m - c_raw) sp when proj0 = pMl + c_d);
  // d pMl / d w = gamma e / Zq - beta.  dth columns are indexed by MODEL parameter: 0-2 main lens, 3-4 c_raw, 5-6 k, 7-8 p
  const double mu = c.gamma * c.e * iZq - c.beta;
  for (int j = 3; j < 9; ++j) dth[0][j] = dth[1][j] = 0.0;
  for (int l = 0; l < 6; ++l) {
    const bool active = (l < 2) || (l < 4 && (l - 2) < c.nr) || (l >= 4 && c.tan);
    if (!active) continue;
    const int mcol = 3 + l;   // lens-parameter order (c_raw.x, c_raw.y, k1, k2, p1, p2) = model columns 3..8
    for (int a = 0; a < 2; ++a) {
      const double dw = wsc * L.t[l][a];
      double d = mu * dw;
      if (c.adj) d += dw;                                              // proj0 = pMl + w
      else if (l == a) d += -c.sp[a];                                  // proj0 = pMl + c_d, d c_d / d c_raw = -sp
      dth[a][mcol] = d;
    }
  }
  dpc[0][0] = c.gamma * iZq; dpc[0][1] = 0.0; dpc[0][2] = -c.gamma * q[0] * iZq;
  dpc[1][0] = 0.0; dpc[1][1] = c.gamma * iZq; dpc[1][2] = -c.gamma * q[1] * iZq;
  if (c.adj) { proj[0] += w[0]; proj[1] += w[1]; } else { proj[0] += L.cd[0]; proj[1] += L.cd[1]; }
  if (c.adj && (c.nr > 0 || c.tan)) {    // proj += Delta(proj) (CameraModel.h:152-176)
    ADist d; adist(c, proj[0], proj[1], d);
    const double b00 = 1.0 + d.a00, b01 = d.a01, b10 = d.a10, b11 = 1.0 + d.a11;
    for (int j = 0; j < 9; ++j) {
      double ex = 0.0, ey = 0.0;
      if (j == 5 && c.nr >= 1) { ex = d.ek[0][0]; ey = d.ek[0][1]; }
      if (j == 6 && c.nr >= 2) { ex = d.ek[1][0]; ey = d.ek[1][1]; }
      if (j == 7 && c.tan) { ex = d.ep[0][0]; ey = d.ep[0][1]; }
      if (j == 8 && c.tan) { ex = d.ep[1][0]; ey = d.ep[1][1]; }
      const double nx = b00 * dth[0][j] + b01 * dth[1][j] + ex, ny = b10 * dth[0][j] + b11 * dth[1][j] + ey;
      dth[0][j] = nx; dth[1][j] = ny;
    }
    for (int j = 0; j < 3; ++j) {
      const double nx = b00 * dpc[0][j] + b01 * dpc[1][j], ny = b10 * dpc[0][j] + b11 * dpc[1][j];
      dpc[0][j] = nx; dpc[1][j] = ny;
    }
    proj[0] += d.dx; proj[1] += d.dy;
  }
  r[0] = proj[0] / c.sp[0] + c.cr[0] - u;     // CameraModel.h:194-195, BundleAdjustment.h:191-192
  r[1] = proj[1] / c.sp[1] + c.cr[1] - v;
  for (int a = 0; a < 2; ++a) {
    for (int j = 0; j < 17; ++j) Jth[a][j] = 0.0;
    for (int j = 0; j < 3; ++j) Jpc[a][j] = dpc[a][j] / c.sp[a];
    for (int j = 0; j < 5; ++j) Jth[a][j] = dth[a][j] / c.sp[a];
    Jth[a][3 + a] += 1.0;                                              // "+ c_raw"
    for (int i = 0; i < c.nr; ++i) Jth[a][5 + i] = dth[a][5 + i] / c.sp[a];
    if (c.tan) { Jth[a][5 + c.nr] = dth[a][7] / c.sp[a]; Jth[a][6 + c.nr] = dth[a][8] / c.sp[a]; }
    for (int j = 0; j < nlive && j < 5; ++j) Jth[a][j] *= c.ch[j];
  }
}

}  // namespace lo
