// device_model.hpp — plenoptic camera model with a hand-derived analytic Jacobian (gfx950 device code).
//
// What it computes (same mathematics as the reference, different algorithm):
//   reference src/CameraModel.h:86-199   CameraModel::projectPoint<T>          -> lens_eval + obs_eval / obs_value
//   reference src/CameraModel.h:205-241  radialDistortion / tangentialDistortion -> distortion()
//   reference src/CameraModel.h:246-264  RigidBody::getTransformationMatrix<T>  -> frame_eval
//   reference src/BundleAdjustment/BundleAdjustment.h:120-195 operator_function -> cam_prepare (sign folding) + obs_eval
// The reference differentiates this with ceres::Jet<double,26> per observation (≈38 kflop); here
//   * everything that depends on the camera only is computed once (CamConsts),
//   * the 10-sweep undistortion of the micro-lens centre and its tangents w.r.t. (c_raw, k, p) is
//     tabulated per UNIQUE lens (lens_eval) — the tangent recurrence unrolls the same fixed-point
//     sweeps autodiff would, so the derivative is that of the truncated iteration, not the implicit one,
//   * the rotation and translation are tabulated per frame (frame_eval),
//   * per observation only the closed-form chain remains (obs_eval): residual r[2],
//     Jq = dr/d(camera-frame point) [2x3] and Jc = dr/d(camera slots) [2xNC].
// Pose and point Jacobians follow from Jq per (point, frame) group: J_t = Jq, J_a = Jq [ax_k x (R P)],
// J_P = Jq R (see group_transform in kernels.hpp).
#pragma once
#include <hip/hip_runtime.h>

namespace lifcal {

#define LIFCAL_DEV __device__ __forceinline__

constexpr int NCMAX = 9;       // 5 + 2 radial + 2 tangential live camera slots
constexpr int LENS_STRIDE = 16;  // doubles per lens-table entry
constexpr int FRAME_STRIDE = 16; // doubles per frame-table entry

// Camera-only quantities.  Model parameters theta = (fL, bL0, B, c_raw.x, c_raw.y, k1, k2, p1, p2);
// chm[j] = d theta_j / d camera[j] (sign folding / +-scale) times the free-column mask.
struct CamConsts {
  double fL, bL0, B, ifL;
  double craw[2];
  double sp[2], isp[2];
  double D, iD, e, zC0, gamma, beta;
  double de[3], dzC0[3], dgamma[3], dbeta[3];
  double a, da[3];
  double k[2], p[2];
  double chm[NCMAX];
  double loss_b, loss_c;
};

// reference BundleAdjustment.h:123-146 (fold = true) or CameraCalibration.cpp:1028-1039 (fold = false,
// scale cast through float as calcReprojectionError does)
LIFCAL_DEV void cam_prepare(const double* cam, double spx, double spy, double scale, int n_radial, bool tangential,
                            unsigned fixed_mask, double loss_scale, bool fold, CamConsts& c) {
  double sg[3] = {1.0, 1.0, 1.0};
  double th[3];
  for (int i = 0; i < 3; ++i) { th[i] = cam[i]; if (fold && th[i] < 0.0) { th[i] = -th[i]; sg[i] = -1.0; } }
  c.fL = th[0]; c.bL0 = th[1]; c.B = th[2]; c.ifL = 1.0 / c.fL;
  double dcr[2];
  for (int i = 0; i < 2; ++i) {
    double v = (cam[3 + i] + 0.5) * scale - 0.5; dcr[i] = scale;
    if (fold && v < 0.0) { v = -v; dcr[i] = -scale; }
    c.craw[i] = v;
  }
  c.sp[0] = spx / scale; c.sp[1] = spy / scale; c.isp[0] = 1.0 / c.sp[0]; c.isp[1] = 1.0 / c.sp[1];
  const double D = c.fL - c.bL0, iD = 1.0 / D, iD2 = iD * iD;
  c.D = D; c.iD = iD;
  c.e = c.fL * iD; c.zC0 = c.fL * c.bL0 * iD; c.gamma = c.fL * c.B * iD; c.beta = c.B * iD;
  c.de[0] = -c.bL0 * iD2;          c.de[1] = c.fL * iD2;           c.de[2] = 0.0;
  c.dzC0[0] = -c.bL0 * c.bL0 * iD2; c.dzC0[1] = c.fL * c.fL * iD2;   c.dzC0[2] = 0.0;
  c.dgamma[0] = -c.B * c.bL0 * iD2; c.dgamma[1] = c.fL * c.B * iD2;  c.dgamma[2] = c.fL * iD;
  c.dbeta[0] = -c.B * iD2;          c.dbeta[1] = c.B * iD2;          c.dbeta[2] = iD;
  const double s = c.bL0 + c.B, is = 1.0 / s;
  c.a = c.bL0 * is; c.da[0] = 0.0; c.da[1] = c.B * is * is; c.da[2] = -c.bL0 * is * is;
  c.k[0] = n_radial > 0 ? cam[5] : 0.0; c.k[1] = n_radial > 1 ? cam[6] : 0.0;
  c.p[0] = tangential ? cam[5 + n_radial] : 0.0; c.p[1] = tangential ? cam[6 + n_radial] : 0.0;
  const int nc = 5 + n_radial + (tangential ? 2 : 0);
  for (int j = 0; j < NCMAX; ++j) {
    double ch = 1.0;
    if (j < 3) ch = sg[j]; else if (j < 5) ch = dcr[j - 3];
    const bool live = (j < nc) && !((fixed_mask >> j) & 1u);
    c.chm[j] = live ? ch : 0.0;
  }
  c.loss_b = loss_scale * loss_scale; c.loss_c = 1.0 / c.loss_b;
}

// Distortion value Delta(x,y) (reference CameraModel.h:205-241), its 2x2 Jacobian A and the explicit
// partials w.r.t. k_i, p_i.  NR / TAN are compile-time so unused terms vanish.
template <int NR, bool TAN>
struct Distortion {
  double dx, dy, A00, A01, A10, A11, r2, r4;
  LIFCAL_DEV void eval(double x, double y, const CamConsts& c, bool want_jac) {
    r2 = x * x + y * y; r4 = r2 * r2;
    double g = 0.0, gp = 0.0;
    if (NR >= 1) { g = c.k[0] * r2; gp = c.k[0]; }
    if (NR >= 2) { g += c.k[1] * r4; gp += 2.0 * c.k[1] * r2; }
    dx = x * g; dy = y * g;
    if (want_jac) { const double xy2 = 2.0 * x * y * gp; A00 = g + 2.0 * x * x * gp; A01 = xy2; A10 = xy2; A11 = g + 2.0 * y * y * gp; }
    if (TAN) {
      dx += c.p[0] * (r2 + 2.0 * x * x) + 2.0 * c.p[1] * x * y;
      dy += c.p[1] * (r2 + 2.0 * y * y) + 2.0 * c.p[0] * x * y;
      if (want_jac) {
        A00 += 6.0 * c.p[0] * x + 2.0 * c.p[1] * y; A01 += 2.0 * c.p[0] * y + 2.0 * c.p[1] * x;
        A10 += 2.0 * c.p[1] * x + 2.0 * c.p[0] * y; A11 += 6.0 * c.p[1] * y + 2.0 * c.p[0] * x;
      }
    }
  }
  // explicit partial of Delta w.r.t. lens-parameter a (0: c_raw.x, 1: c_raw.y, 2..: k, then p)
  LIFCAL_DEV void explicit_partial(int a, double x, double y, double& ex, double& ey) const {
    ex = 0.0; ey = 0.0;
    if (NR >= 1 && a == 2) { ex = x * r2; ey = y * r2; }
    if (NR >= 2 && a == 3) { ex = x * r4; ey = y * r4; }
    if (TAN && a == 2 + NR) { ex = r2 + 2.0 * x * x; ey = 2.0 * x * y; }
    if (TAN && a == 3 + NR) { ex = 2.0 * x * y; ey = r2 + 2.0 * y * y; }
  }
};

// Lens-table entry: [0,1] lens centre, [2,3] undistorted centre c_u (mm, before the mlCenterAdj factor),
// [4+2a, 5+2a] d c_u / d lens-parameter a, a < NA = 2 + NR + 2*TAN.
template <int NR, bool TAN>
LIFCAL_DEV void lens_eval(const CamConsts& c, double mx, double my, bool want_tangents, double* out) {
  constexpr int NA = 2 + NR + (TAN ? 2 : 0);
  const double cdx = (mx - c.craw[0]) * c.sp[0], cdy = (my - c.craw[1]) * c.sp[1];
  double x = cdx, y = cdy;
  double tx[NA], ty[NA], d0x[NA], d0y[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) { d0x[a] = (a == 0) ? -c.sp[0] : 0.0; d0y[a] = (a == 1) ? -c.sp[1] : 0.0; tx[a] = d0x[a]; ty[a] = d0y[a]; }
  if (NR > 0 || TAN) {
    Distortion<NR, TAN> d;
    for (int it = 0; it < 10; ++it) {  // reference CameraModel.h:109-124
      d.eval(x, y, c, want_tangents);
      if (want_tangents) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          double ex, ey; d.explicit_partial(a, x, y, ex, ey);
          const double nx = d0x[a] - (d.A00 * tx[a] + d.A01 * ty[a]) - ex;
          const double ny = d0y[a] - (d.A10 * tx[a] + d.A11 * ty[a]) - ey;
          tx[a] = nx; ty[a] = ny;
        }
      }
      x = cdx - d.dx; y = cdy - d.dy;
    }
  }
  out[0] = mx; out[1] = my; out[2] = x; out[3] = y;
  if (want_tangents) {
#pragma unroll
    for (int a = 0; a < NA; ++a) { out[4 + 2 * a] = tx[a]; out[5 + 2 * a] = ty[a]; }
  }
}

// Frame-table entry: [0..8] R row-major (R = Rx Ry Rz), [9..11] t, [12] cos a0, [13] sin a0.
// d(R P)/d a0 = e_x x (R P), d/d a1 = (0, cos a0, sin a0) x (R P), d/d a2 = R[:,2] x (R P).
LIFCAL_DEV void frame_eval(const double* view, double* out) {
  double s0, c0, s1, c1, s2, c2;
  sincos(view[0], &s0, &c0); sincos(view[1], &s1, &c1); sincos(view[2], &s2, &c2);
  out[0] = c1 * c2;                 out[1] = -c1 * s2;                out[2] = s1;
  out[3] = c0 * s2 + s0 * s1 * c2;  out[4] = c0 * c2 - s0 * s1 * s2;  out[5] = -s0 * c1;
  out[6] = s0 * s2 - c0 * s1 * c2;  out[7] = s0 * c2 + c0 * s1 * s2;  out[8] = c0 * c1;
  out[9] = view[3]; out[10] = view[4]; out[11] = view[5];
  out[12] = c0; out[13] = s0; out[14] = 0.0; out[15] = 0.0;
}

// quantities shared by all observations of one (point, frame) group
struct GroupConsts {
  double X, Y, iZq;     // camera-frame point, 1/(Z + zC0)
  double kq[3], kw[3];  // d pMl / d(fL,bL0,B) = q * kq + w * kw + mu * dw
  double mu, gz;        // gamma e iZq - beta ; gamma iZq
};

LIFCAL_DEV void group_prepare(const CamConsts& c, double X, double Y, double Z, GroupConsts& g) {
  g.X = X; g.Y = Y; g.iZq = 1.0 / (Z + c.zC0);
  g.gz = c.gamma * g.iZq;
  g.mu = g.gz * c.e - c.beta;
#pragma unroll
  for (int i = 0; i < 3; ++i) { g.kq[i] = c.dgamma[i] - g.gz * c.dzC0[i]; g.kw[i] = g.gz * c.de[i] - c.dbeta[i]; }
}

// value-only projection residual (cost evaluation, reprojection statistics)
template <int NR, bool TAN, bool ADJ>
LIFCAL_DEV void obs_value(const CamConsts& c, const GroupConsts& g, double mx, double my, double cux, double cuy,
                          double u, double v, double& rx, double& ry) {
  const double wx = ADJ ? cux * c.a : cux, wy = ADJ ? cuy * c.a : cuy;
  const double qx = (g.X + wx * c.e) * g.iZq, qy = (g.Y + wy * c.e) * g.iZq;
  const double mlx = c.gamma * qx - c.beta * wx, mly = c.gamma * qy - c.beta * wy;
  double px, py;
  if (ADJ) {
    px = mlx + wx; py = mly + wy;
    if (NR > 0 || TAN) { Distortion<NR, TAN> d; d.eval(px, py, c, false); px += d.dx; py += d.dy; }
  } else {
    px = mlx + (mx - c.craw[0]) * c.sp[0]; py = mly + (my - c.craw[1]) * c.sp[1];
  }
  rx = px * c.isp[0] + c.craw[0] - u;
  ry = py * c.isp[1] + c.craw[1] - v;
}

// residual + analytic Jacobian of one observation.  L = lens-table entry (with tangents).
// Jc[row][j]: j indexes the live camera slots in camera[] order (fL,bL0,B,cx,cy,k..,p..).
template <int NR, bool TAN, bool ADJ>
LIFCAL_DEV void obs_eval(const CamConsts& c, const GroupConsts& g, const double* __restrict__ L, double u, double v,
                         double r[2], double Jq[2][3], double Jc[2][5 + NR + (TAN ? 2 : 0)]) {
  constexpr int NA = 2 + NR + (TAN ? 2 : 0);
  constexpr int NC = 3 + NA;
  const double mx = L[0], my = L[1], cux = L[2], cuy = L[3];
  const double wx = ADJ ? cux * c.a : cux, wy = ADJ ? cuy * c.a : cuy;
  const double qx = (g.X + wx * c.e) * g.iZq, qy = (g.Y + wy * c.e) * g.iZq;
  const double mlx = c.gamma * qx - c.beta * wx, mly = c.gamma * qy - c.beta * wy;
  // d(pMl)/d theta: columns 0..2 = (fL,bL0,B), 3.. = lens parameters
  double dx[NC], dy[NC];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dx[i] = qx * g.kq[i] + wx * g.kw[i];
    dy[i] = qy * g.kq[i] + wy * g.kw[i];
    if (ADJ) { dx[i] += g.mu * cux * c.da[i]; dy[i] += g.mu * cuy * c.da[i]; }
  }
  const double wscale = ADJ ? c.a : 1.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) { dx[3 + a] = g.mu * wscale * L[4 + 2 * a]; dy[3 + a] = g.mu * wscale * L[5 + 2 * a]; }
  // d(pMl)/d p_c
  double qxX = g.gz, qxZ = -g.gz * qx, qyY = g.gz, qyZ = -g.gz * qy;
  double px, py;
  double j00, j01, j02, j10, j11, j12;  // d proj / d (X,Y,Z)
  if (ADJ) {
    // proj0 = pMl + w ; proj = proj0 + Delta(proj0)   (reference CameraModel.h:152-176)
#pragma unroll
    for (int i = 0; i < 3; ++i) { dx[i] += cux * c.da[i]; dy[i] += cuy * c.da[i]; }
#pragma unroll
    for (int a = 0; a < NA; ++a) { dx[3 + a] += wscale * L[4 + 2 * a]; dy[3 + a] += wscale * L[5 + 2 * a]; }
    px = mlx + wx; py = mly + wy;
    if (NR > 0 || TAN) {
      Distortion<NR, TAN> d; d.eval(px, py, c, true);
      const double b00 = 1.0 + d.A00, b01 = d.A01, b10 = d.A10, b11 = 1.0 + d.A11;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        double ex = 0.0, ey = 0.0;
        if (i >= 3) d.explicit_partial(i - 3, px, py, ex, ey);
        const double nx = b00 * dx[i] + b01 * dy[i] + ex, ny = b10 * dx[i] + b11 * dy[i] + ey;
        dx[i] = nx; dy[i] = ny;
      }
      j00 = b00 * qxX; j01 = b01 * qyY; j02 = b00 * qxZ + b01 * qyZ;
      j10 = b10 * qxX; j11 = b11 * qyY; j12 = b10 * qxZ + b11 * qyZ;
      px += d.dx; py += d.dy;
    } else {
      j00 = qxX; j01 = 0.0; j02 = qxZ; j10 = 0.0; j11 = qyY; j12 = qyZ;
    }
    // out = proj / sp + c_raw : the c_raw term adds 1 to d/d c_raw
    dx[3] += c.sp[0]; dy[4] += c.sp[1];
  } else {
    // proj = pMl + c_d (reference :187-190); d c_d/d c_raw = -sp cancels the +1 of "+ c_raw" exactly
    px = mlx + (mx - c.craw[0]) * c.sp[0]; py = mly + (my - c.craw[1]) * c.sp[1];
    j00 = qxX; j01 = 0.0; j02 = qxZ; j10 = 0.0; j11 = qyY; j12 = qyZ;
  }
  r[0] = px * c.isp[0] + c.craw[0] - u;
  r[1] = py * c.isp[1] + c.craw[1] - v;
  Jq[0][0] = j00 * c.isp[0]; Jq[0][1] = j01 * c.isp[0]; Jq[0][2] = j02 * c.isp[0];
  Jq[1][0] = j10 * c.isp[1]; Jq[1][1] = j11 * c.isp[1]; Jq[1][2] = j12 * c.isp[1];
#pragma unroll
  for (int j = 0; j < NC; ++j) { Jc[0][j] = dx[j] * c.isp[0] * c.chm[j]; Jc[1][j] = dy[j] * c.isp[1] * c.chm[j]; }
}


// ---- lean variant for the LDS-window kernel (k_sweep2) ----
// Same mathematics as obs_eval, three per-observation savings:
//   * d pMl/d(fL,bL0,B) = q kq + c_u kc with kc = a kw + (mu+1) da folded per group (ADJ),
//   * d pMl/d(lens parameter) = gl * tangent with gl = (mu+1) a folded per group,
//   * the pixel scale 1/sp and the robust weight sqrt(rho') go into ONE pair of row factors; the sign/scale folding chm[j]
//     is NOT applied: Jc comes back in model-parameter columns and the caller scales its accumulated blocks once.
struct GroupConsts2 {
  double X, Y, iZq, gz;
  double kq[3], kc[3];   // ADJ: dx[i] = q kq[i] + c_u kc[i];  !ADJ: kc = kw (multiplies w = c_u)
  double gl;             // ADJ: (mu + 1) a;  !ADJ: mu
};

template <bool ADJ>
LIFCAL_DEV void group_prepare2(const CamConsts& c, double X, double Y, double Z, GroupConsts2& g) {
  g.X = X; g.Y = Y; g.iZq = 1.0 / (Z + c.zC0);
  g.gz = c.gamma * g.iZq;
  const double mu = g.gz * c.e - c.beta;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g.kq[i] = c.dgamma[i] - g.gz * c.dzC0[i];
    const double kw = g.gz * c.de[i] - c.dbeta[i];
    g.kc[i] = ADJ ? c.a * kw + (mu + 1.0) * c.da[i] : kw;
  }
  g.gl = ADJ ? (mu + 1.0) * c.a : mu;
}

// returns the Cauchy argument 1 + s/b (robust) or the squared residual norm (not robust) through `arg`;
// r, Jq, Jc come back already multiplied by sqrt(rho') when robust
template <int NR, bool TAN, bool ADJ>
LIFCAL_DEV void obs_eval2(const CamConsts& c, const GroupConsts2& g, const double* __restrict__ L, double u, double v, bool robust,
                          double r[2], double Jq[2][3], double Jc[2][5 + NR + (TAN ? 2 : 0)], double& arg) {
  constexpr int NA = 2 + NR + (TAN ? 2 : 0);
  constexpr int NC = 3 + NA;
  const double mx = L[0], my = L[1], cux = L[2], cuy = L[3];
  const double wx = ADJ ? cux * c.a : cux, wy = ADJ ? cuy * c.a : cuy;
  const double qx = (g.X + wx * c.e) * g.iZq, qy = (g.Y + wy * c.e) * g.iZq;
  const double mlx = c.gamma * qx - c.beta * wx, mly = c.gamma * qy - c.beta * wy;
  double dx[NC], dy[NC];
#pragma unroll
  for (int i = 0; i < 3; ++i) { dx[i] = qx * g.kq[i] + cux * g.kc[i]; dy[i] = qy * g.kq[i] + cuy * g.kc[i]; }
#pragma unroll
  for (int a = 0; a < NA; ++a) { dx[3 + a] = g.gl * L[4 + 2 * a]; dy[3 + a] = g.gl * L[5 + 2 * a]; }
  const double qxZ = -g.gz * qx, qyZ = -g.gz * qy;
  double px, py;
  double j00, j01, j02, j10, j11, j12;  // d proj / d (X,Y,Z)
  if (ADJ) {
    px = mlx + wx; py = mly + wy;
    if (NR > 0 || TAN) {
      Distortion<NR, TAN> d; d.eval(px, py, c, true);
      const double b00 = 1.0 + d.A00, b01 = d.A01, b10 = d.A10, b11 = 1.0 + d.A11;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        double ex = 0.0, ey = 0.0;
        if (i >= 3) d.explicit_partial(i - 3, px, py, ex, ey);
        const double nx = b00 * dx[i] + b01 * dy[i] + ex, ny = b10 * dx[i] + b11 * dy[i] + ey;
        dx[i] = nx; dy[i] = ny;
      }
      j00 = b00 * g.gz; j01 = b01 * g.gz; j02 = b00 * qxZ + b01 * qyZ;
      j10 = b10 * g.gz; j11 = b11 * g.gz; j12 = b10 * qxZ + b11 * qyZ;
      px += d.dx; py += d.dy;
    } else {
      j00 = g.gz; j01 = 0.0; j02 = qxZ; j10 = 0.0; j11 = g.gz; j12 = qyZ;
    }
    dx[3] += c.sp[0]; dy[4] += c.sp[1];
  } else {
    px = mlx + (mx - c.craw[0]) * c.sp[0]; py = mly + (my - c.craw[1]) * c.sp[1];
    j00 = g.gz; j01 = 0.0; j02 = qxZ; j10 = 0.0; j11 = g.gz; j12 = qyZ;
  }
  r[0] = px * c.isp[0] + c.craw[0] - u;
  r[1] = py * c.isp[1] + c.craw[1] - v;
  const double sq = r[0] * r[0] + r[1] * r[1];
  double s0 = c.isp[0], s1 = c.isp[1];
  if (robust) {   // ceres::CauchyLoss + Corrector with rho'' < 0: r and J scaled by sqrt(rho') = rsqrt(1 + s/b)
    arg = 1.0 + sq * c.loss_c;
    const double sc = rsqrt(arg);
    r[0] *= sc; r[1] *= sc; s0 *= sc; s1 *= sc;
  } else {
    arg = sq;
  }
  Jq[0][0] = j00 * s0; Jq[0][1] = j01 * s0; Jq[0][2] = j02 * s0;
  Jq[1][0] = j10 * s1; Jq[1][1] = j11 * s1; Jq[1][2] = j12 * s1;
#pragma unroll
  for (int j = 0; j < NC; ++j) { Jc[0][j] = dx[j] * s0; Jc[1][j] = dy[j] * s1; }
}

// ---- fp32 evaluation for options.precision = 1 (BASELINE configs[4]: "fp32 residuals / fp64 normal-eq accumulate") ----
// Same chain as obs_eval2 in single precision.  What keeps fp32 adequate on ~1000-pixel coordinates:
//   * the observation is stored relative to its micro-lens centre m (du = u - m.x, dv = v - m.y: a few pixels),
//   * the lens table carries, per unique lens and computed in fp64, w = (a) c_u [mm] and the small pixel offset
//     Lm = w / sp + c_raw - m (mlCenterAdj; 0 otherwise: c_d / sp + c_raw - m vanishes identically),
//   so r = pMl / sp + Lm + Delta / sp - du is a sum of SMALL terms;
//   * the one cancellation left, inside pMl = gamma q - beta w (point direction against lens direction, ~40 : 1, which costs
//     ~1e-4 px in plain fp32), is taken in fp64: pMl = -gamma hZ (w - w0) with hZ = (Z - fL) / ((Z + zC0) fL) and w0 = (X, Y) fL / (Z - fL)
//     the lens position that images the point onto the lens centre, both per group in fp64; per observation ONE fp64 subtraction
//     per axis (w comes from a small fp64 side table), the difference (a fraction of a millimetre) continues in fp32.
//   Measured on BASELINE configs[1]: converged intrinsics within 2e-5 of the fp64 arm without, a few 1e-7 with this step.
// Group constants are computed in fp64 (group_prepare2) and rounded once.
struct CamF { float a, e, gamma, beta, isp0, isp1, sp0, sp1, k0, k1, p0, p1, loss_c; };
LIFCAL_DEV CamF cam_to_float(const CamConsts& c) {
  CamF f;
  f.a = (float)c.a; f.e = (float)c.e; f.gamma = (float)c.gamma; f.beta = (float)c.beta;
  f.isp0 = (float)c.isp[0]; f.isp1 = (float)c.isp[1]; f.sp0 = (float)c.sp[0]; f.sp1 = (float)c.sp[1];
  f.k0 = (float)c.k[0]; f.k1 = (float)c.k[1]; f.p0 = (float)c.p[0]; f.p1 = (float)c.p[1]; f.loss_c = (float)c.loss_c;
  return f;
}
struct GroupConsts2F { float X, Y, iZq, gz, kq[3], kc[3], gl, mgx, mgy; double w0x, w0y; };   // mg = -gamma hZ / sp
template <bool ADJ>
LIFCAL_DEV void group_prepare2f(const CamConsts& c, double X, double Y, double Z, GroupConsts2F& g) {
  GroupConsts2 d; group_prepare2<ADJ>(c, X, Y, Z, d);
  g.X = (float)d.X; g.Y = (float)d.Y; g.iZq = (float)d.iZq; g.gz = (float)d.gz; g.gl = (float)d.gl;
  const double izf = 1.0 / (Z - c.fL), hZ = (Z - c.fL) * d.iZq * c.ifL;
  g.w0x = X * c.fL * izf; g.w0y = Y * c.fL * izf;
  g.mgx = (float)(-c.gamma * hZ * c.isp[0]); g.mgy = (float)(-c.gamma * hZ * c.isp[1]);
#pragma unroll
  for (int i = 0; i < 3; ++i) { g.kq[i] = (float)d.kq[i]; g.kc[i] = (float)(ADJ ? d.kc[i] / c.a : d.kc[i]); }   // multiplies w = a c_u instead of c_u
}

template <int NR, bool TAN>
struct DistortionF {
  float dx, dy, A00, A01, A10, A11, r2, r4;
  LIFCAL_DEV void eval(float x, float y, const CamF& c) {
    r2 = x * x + y * y; r4 = r2 * r2;
    float g = 0.f, gp = 0.f;
    if (NR >= 1) { g = c.k0 * r2; gp = c.k0; }
    if (NR >= 2) { g += c.k1 * r4; gp += 2.f * c.k1 * r2; }
    dx = x * g; dy = y * g;
    const float xy2 = 2.f * x * y * gp; A00 = g + 2.f * x * x * gp; A01 = xy2; A10 = xy2; A11 = g + 2.f * y * y * gp;
    if (TAN) {
      dx += c.p0 * (r2 + 2.f * x * x) + 2.f * c.p1 * x * y;
      dy += c.p1 * (r2 + 2.f * y * y) + 2.f * c.p0 * x * y;
      A00 += 6.f * c.p0 * x + 2.f * c.p1 * y; A01 += 2.f * c.p0 * y + 2.f * c.p1 * x;
      A10 += 2.f * c.p1 * x + 2.f * c.p0 * y; A11 += 6.f * c.p1 * y + 2.f * c.p0 * x;
    }
  }
  LIFCAL_DEV void explicit_partial(int a, float x, float y, float& ex, float& ey) const {
    ex = 0.f; ey = 0.f;
    if (NR >= 1 && a == 2) { ex = x * r2; ey = y * r2; }
    if (NR >= 2 && a == 3) { ex = x * r4; ey = y * r4; }
    if (TAN && a == 2 + NR) { ex = r2 + 2.f * x * x; ey = 2.f * x * y; }
    if (TAN && a == 3 + NR) { ex = 2.f * x * y; ey = r2 + 2.f * y * y; }
  }
};

// fp32 lens-table row (16 floats): [0,1] w = (a) c_u [mm], [2,3] Lm [px], [4+2l, 5+2l] d c_u / d lens-parameter l (as the fp64 row)
// r, Jq, Jc come back multiplied by sqrt(rho') when robust; arg = 1 + s/b (robust) or s (not robust), as obs_eval2
template <int NR, bool TAN, bool ADJ>
LIFCAL_DEV void obs_eval2f(const CamF& c, const GroupConsts2F& g, const float* __restrict__ L, double w64x, double w64y, float du, float dv, bool robust,
                           float r[2], float Jq[2][3], float Jc[2][5 + NR + (TAN ? 2 : 0)], float& arg) {
  constexpr int NA = 2 + NR + (TAN ? 2 : 0);
  constexpr int NC = 3 + NA;
  const float wx = L[0], wy = L[1];
  const float qx = (g.X + wx * c.e) * g.iZq, qy = (g.Y + wy * c.e) * g.iZq;
  const float mlx = c.gamma * qx - c.beta * wx, mly = c.gamma * qy - c.beta * wy;
  float dx[NC], dy[NC];
#pragma unroll
  for (int i = 0; i < 3; ++i) { dx[i] = qx * g.kq[i] + wx * g.kc[i]; dy[i] = qy * g.kq[i] + wy * g.kc[i]; }
#pragma unroll
  for (int a = 0; a < NA; ++a) { dx[3 + a] = g.gl * L[4 + 2 * a]; dy[3 + a] = g.gl * L[5 + 2 * a]; }
  const float qxZ = -g.gz * qx, qyZ = -g.gz * qy;
  float j00, j01, j02, j10, j11, j12;
  // residual: pMl / sp from the fp64 difference (see above); mlx / mly (plain fp32) only feed the distortion argument
  float r0 = g.mgx * (float)(w64x - g.w0x) - du, r1 = g.mgy * (float)(w64y - g.w0y) - dv;
  if (ADJ) {
    r0 += L[2]; r1 += L[3];
    if (NR > 0 || TAN) {
      const float px = mlx + wx, py = mly + wy;
      DistortionF<NR, TAN> d; d.eval(px, py, c);
      const float b00 = 1.f + d.A00, b01 = d.A01, b10 = d.A10, b11 = 1.f + d.A11;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        float ex = 0.f, ey = 0.f;
        if (i >= 3) d.explicit_partial(i - 3, px, py, ex, ey);
        const float nx = b00 * dx[i] + b01 * dy[i] + ex, ny = b10 * dx[i] + b11 * dy[i] + ey;
        dx[i] = nx; dy[i] = ny;
      }
      j00 = b00 * g.gz; j01 = b01 * g.gz; j02 = b00 * qxZ + b01 * qyZ;
      j10 = b10 * g.gz; j11 = b11 * g.gz; j12 = b10 * qxZ + b11 * qyZ;
      r0 += d.dx * c.isp0; r1 += d.dy * c.isp1;
    } else {
      j00 = g.gz; j01 = 0.f; j02 = qxZ; j10 = 0.f; j11 = g.gz; j12 = qyZ;
    }
    dx[3] += c.sp0; dy[4] += c.sp1;
  } else {
    j00 = g.gz; j01 = 0.f; j02 = qxZ; j10 = 0.f; j11 = g.gz; j12 = qyZ;
  }
  r[0] = r0; r[1] = r1;
  const float sq = r0 * r0 + r1 * r1;
  float s0 = c.isp0, s1 = c.isp1;
  if (robust) {
    arg = 1.f + sq * c.loss_c;
    const float sc = rsqrtf(arg);
    r[0] *= sc; r[1] *= sc; s0 *= sc; s1 *= sc;
  } else {
    arg = sq;
  }
  Jq[0][0] = j00 * s0; Jq[0][1] = j01 * s0; Jq[0][2] = j02 * s0;
  Jq[1][0] = j10 * s1; Jq[1][1] = j11 * s1; Jq[1][2] = j12 * s1;
#pragma unroll
  for (int j = 0; j < NC; ++j) { Jc[0][j] = dx[j] * s0; Jc[1][j] = dy[j] * s1; }
}

// ---- the same evaluation in PACKED fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two floats per lane and instruction) ----
// The model is a chain of (x, y) pairs — lens offset w, direction q, micro-image point, every Jacobian column (dx[i], dy[i]),
// the row factors (1/sp_x, 1/sp_y) — so the pairs live in <2 x float> values and one packed instruction does both components:
// about 85 vector instructions per observation instead of about 180.  Same operations in the same order as obs_eval2f (the
// results agree to the last bit wherever the compiler contracts the same multiply-adds).
typedef float f32x2 __attribute__((ext_vector_type(2)));
LIFCAL_DEV f32x2 pk(float a, float b) { f32x2 r; r.x = a; r.y = b; return r; }
LIFCAL_DEV f32x2 pk1(float a) { f32x2 r; r.x = a; r.y = a; return r; }
template <int NR, bool TAN, bool ADJ>
LIFCAL_DEV void obs_eval2f_pk(const CamF& c, const GroupConsts2F& g, const float* __restrict__ L, double w64x, double w64y, float du, float dv, bool robust,
                              float r[2], float Jq[2][3], float Jc[2][5 + NR + (TAN ? 2 : 0)], float& arg) {
  constexpr int NA = 2 + NR + (TAN ? 2 : 0);
  constexpr int NC = 3 + NA;
  const f32x2 w = pk(L[0], L[1]);
  const f32x2 q = (pk(g.X, g.Y) + w * c.e) * g.iZq;
  const f32x2 ml = q * c.gamma - w * c.beta;
  f32x2 d[NC];
#pragma unroll
  for (int i = 0; i < 3; ++i) d[i] = q * g.kq[i] + w * g.kc[i];
#pragma unroll
  for (int a = 0; a < NA; ++a) d[3 + a] = pk(L[4 + 2 * a], L[5 + 2 * a]) * g.gl;
  const f32x2 qZ = q * (-g.gz);
  f32x2 jd, jo, j2;   // (j00, j11), (j01, j10), (j02, j12)
  // residual: pMl / sp from the fp64 difference (see obs_eval2f); ml (plain fp32) only feeds the distortion argument
  f32x2 rr = pk(g.mgx, g.mgy) * pk((float)(w64x - g.w0x), (float)(w64y - g.w0y)) - pk(du, dv);
  if (ADJ) {
    rr += pk(L[2], L[3]);
    if (NR > 0 || TAN) {
      const f32x2 p = ml + w;
      const f32x2 pp = p * p;
      const float r2 = pp.x + pp.y, r4 = r2 * r2, xy = p.x * p.y;
      float gg = 0.f, gp = 0.f;
      if (NR >= 1) { gg = c.k0 * r2; gp = c.k0; }
      if (NR >= 2) { gg += c.k1 * r4; gp += 2.f * c.k1 * r2; }
      f32x2 dd = p * gg;                                  // Delta
      f32x2 Ad = pk1(gg) + pp * (2.f * gp);               // (A00, A11)
      float Ao = 2.f * xy * gp;                           // A01 = A10
      const f32x2 pT = pk(c.p0, c.p1), pS = pk(c.p1, c.p0);
      if (TAN) {
        dd += pT * (pk1(r2) + pp * 2.f) + pS * (2.f * xy);
        Ad += pT * p * 6.f + pS * pk(p.y, p.x) * 2.f;
        Ao += 2.f * c.p0 * p.y + 2.f * c.p1 * p.x;
      }
      const f32x2 bd = pk1(1.f) + Ad;                     // (b00, b11); b01 = b10 = Ao
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        f32x2 e = pk1(0.f);
        if (NR >= 1 && i == 5) e = p * r2;
        if (NR >= 2 && i == 6) e = p * r4;
        if (TAN && i == 5 + NR) e = pk(r2 + 2.f * pp.x, 2.f * xy);
        if (TAN && i == 6 + NR) e = pk(2.f * xy, r2 + 2.f * pp.y);
        d[i] = bd * d[i] + pk(d[i].y, d[i].x) * Ao + e;
      }
      jd = bd * g.gz; jo = pk1(Ao * g.gz);
      j2 = bd * qZ + pk(qZ.y, qZ.x) * Ao;
      rr += dd * pk(c.isp0, c.isp1);
    } else {
      jd = pk1(g.gz); jo = pk1(0.f); j2 = qZ;
    }
    d[3].x += c.sp0; d[4].y += c.sp1;
  } else {
    jd = pk1(g.gz); jo = pk1(0.f); j2 = qZ;
  }
  const f32x2 r2v = rr * rr;
  const float sq = r2v.x + r2v.y;
  f32x2 sv = pk(c.isp0, c.isp1);
  if (robust) {
    arg = 1.f + sq * c.loss_c;
    const float sc = rsqrtf(arg);
    rr *= sc; sv *= sc;
  } else {
    arg = sq;
  }
  r[0] = rr.x; r[1] = rr.y;
  const f32x2 a0 = jd * sv, a1 = jo * sv, a2 = j2 * sv;
  Jq[0][0] = a0.x; Jq[1][1] = a0.y; Jq[0][1] = a1.x; Jq[1][0] = a1.y; Jq[0][2] = a2.x; Jq[1][2] = a2.y;
#pragma unroll
  for (int j = 0; j < NC; ++j) { const f32x2 t = d[j] * sv; Jc[0][j] = t.x; Jc[1][j] = t.y; }
}

// the fp32 lens-table row from the fp64 one (k_tables), and w in fp64 for the side table
template <bool ADJ>
LIFCAL_DEV void lens_row_to_float(const CamConsts& c, const double* row, float* out, double* w64) {
  const double wx = ADJ ? row[2] * c.a : row[2], wy = ADJ ? row[3] * c.a : row[3];
  w64[0] = wx; w64[1] = wy;
  out[0] = (float)wx; out[1] = (float)wy;
  out[2] = ADJ ? (float)(wx * c.isp[0] + c.craw[0] - row[0]) : 0.f;
  out[3] = ADJ ? (float)(wy * c.isp[1] + c.craw[1] - row[1]) : 0.f;
#pragma unroll
  for (int k = 4; k < LENS_STRIDE; ++k) out[k] = (float)row[k];
}

}  // namespace lifcal
