// oracle/model.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement, generic over the scalar type T (double or lo::Jet<N>), of the reference's
// plenoptic camera model and bundle-adjustment residual functors:
//   radial_distortion      <- CameraModel::radialDistortion<T>      reference src/CameraModel.h:205-223
//   tangential_distortion  <- CameraModel::tangentialDistortion<T>  reference src/CameraModel.h:228-241
//   project_point          <- CameraModel::projectPoint<T>          reference src/CameraModel.h:86-199
//   rigid_transform        <- RigidBody::getTransformationMatrix<T> reference src/CameraModel.h:246-264
//   ObsFunctor             <- OurCostFunctionBundle                 reference src/BundleAdjustment/BundleAdjustment.h:25-252
//   distance_constraint    <- OurConstraintFunctionBundle           reference src/BundleAdjustment/BundleAdjustment.h:255-279
// Written from the reference's arithmetic (same operation order where C++ precedence fixes it),
// not copied; Eigen 3 (absent here) is replaced by plain arrays, its AngleAxis*AngleAxis product
// restated as the unit-quaternion product + Quaternion::toRotationMatrix that Eigen evaluates.
//
// PARITY UNPINNED: the reference ships no tests/golden vectors and cannot be built in this image
// (needs Eigen, Ceres, glog, OpenCV, COLMAP, Boost — none installed; no stand-ins are written).
#pragma once
#include "jet.hpp"

namespace lo {

// reference CameraModel.h:205-223 — delta = (x,y) * sum_i k_i r^(2(i+1)); nParam clamped to 5
template <class T>
inline void radial_distortion(const T& x, const T& y, T& dx, T& dy, const T* k, int n) {
  if (n > 5) n = 5;
  T r0 = x * x + y * y;
  T dr = k[0] * r0;
  T ri = r0;
  for (int i = 1; i < n; ++i) {
    ri = ri * r0;
    dr = dr + k[i] * ri;
  }
  dx = x * dr;
  dy = y * dr;
}

// reference CameraModel.h:228-241
template <class T>
inline void tangential_distortion(const T& x, const T& y, T& dx, T& dy, const T* p) {
  if (p == nullptr) { dx = T(0.0); dy = T(0.0); return; }
  T r2 = x * x + y * y;
  dx = p[0] * (r2 + T(2.0) * x * x) + T(2.0) * p[1] * x * y;
  dy = p[1] * (r2 + T(2.0) * y * y) + T(2.0) * p[0] * x * y;
}

// reference CameraModel.h:86-199 — 3D camera point -> raw pixel in the micro image of one lens
template <class T>
inline void project_point(T& out_x, T& out_y, const T pc[3], const T& spx, const T& spy,
                          const T& fL, const T& bL0, const T& B, const T c_raw[2], const T ml[2],
                          const T* radial, int n_radial, const T* tangential, bool ml_center_adj) {
  // undistort the micro-lens centre: 10 fixed-point sweeps (:92-125)
  T c_dist[2] = {(ml[0] - c_raw[0]) * spx, (ml[1] - c_raw[1]) * spy};
  T c_und[2] = {c_dist[0], c_dist[1]};
  const bool any_dist = (n_radial > 0) || (tangential != nullptr);
  if (any_dist) {
    T rx(0.0), ry(0.0), tx(0.0), ty(0.0);
    for (int it = 0; it < 10; ++it) {
      if (n_radial > 0) radial_distortion<T>(c_und[0], c_und[1], rx, ry, radial, n_radial);
      if (tangential) tangential_distortion<T>(c_und[0], c_und[1], tx, ty, tangential);
      c_und[0] = c_dist[0] - rx - tx;
      c_und[1] = c_dist[1] - ry - ty;
    }
  }
  if (ml_center_adj) {  // :127-131
    c_und[0] = c_und[0] / (bL0 + B) * bL0;
    c_und[1] = c_und[1] / (bL0 + B) * bL0;
  }
  T zC0 = fL * bL0 / (fL - bL0);  // :133
  T pML[2] = {-c_und[0] * fL / (fL - bL0), -c_und[1] * fL / (fL - bL0)};  // :135-137
  T q[3] = {pc[0] - pML[0], pc[1] - pML[1], pc[2] + zC0};  // :139-142
  T qz = q[2];
  q[0] = q[0] / qz; q[1] = q[1] / qz; q[2] = q[2] / qz;  // :144 (p3d_p /= p3d_p[2])
  T pMl[2] = {(q[0] - c_und[0] / fL) * fL * B / (fL - bL0),
              (q[1] - c_und[1] / fL) * fL * B / (fL - bL0)};  // :146-148
  T px, py;
  if (ml_center_adj) {  // :152-176
    px = pMl[0] + c_und[0];
    py = pMl[1] + c_und[1];
    if (any_dist) {
      T rx(0.0), ry(0.0), tx(0.0), ty(0.0);
      if (n_radial > 0) radial_distortion<T>(px, py, rx, ry, radial, n_radial);
      if (tangential) tangential_distortion<T>(px, py, tx, ty, tangential);
      px = px + (rx + tx);  // "projected_x += delta_rad_x + delta_tan_x"
      py = py + (ry + ty);
    }
  } else {  // :177-192 (inner "if(mlCenterAdjustment)" is dead code; scale_x = scale_y = 1)
    px = pMl[0] * T(1.0) + c_dist[0];
    py = pMl[1] * T(1.0) + c_dist[1];
  }
  out_x = px / spx + c_raw[0];  // :194-195
  out_y = py / spy + c_raw[1];
}

// reference CameraModel.h:246-264: R = AngleAxis(a0,X) * AngleAxis(a1,Y) * AngleAxis(a2,Z).
// Eigen 3 evaluates AngleAxis*AngleAxis as a quaternion product and converts once at the end.
template <class T>
struct Quat { T w, x, y, z; };

template <class T>
inline Quat<T> quat_mul(const Quat<T>& a, const Quat<T>& b) {
  Quat<T> r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

template <class T>
inline void rigid_transform(const T ang[3], const T tr[3], T RT[3][4]) {
  using std::cos; using std::sin;
  T h0 = T(0.5) * ang[0], h1 = T(0.5) * ang[1], h2 = T(0.5) * ang[2];
  Quat<T> qx{cos(h0), sin(h0) * T(1.0), sin(h0) * T(0.0), sin(h0) * T(0.0)};
  Quat<T> qy{cos(h1), sin(h1) * T(0.0), sin(h1) * T(1.0), sin(h1) * T(0.0)};
  Quat<T> qz{cos(h2), sin(h2) * T(0.0), sin(h2) * T(0.0), sin(h2) * T(1.0)};
  Quat<T> q = quat_mul(quat_mul(qx, qy), qz);
  T tx = T(2.0) * q.x, ty = T(2.0) * q.y, tz = T(2.0) * q.z;
  T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  RT[0][0] = T(1.0) - (tyy + tzz); RT[0][1] = txy - twz;            RT[0][2] = txz + twy;
  RT[1][0] = txy + twz;            RT[1][1] = T(1.0) - (txx + tzz); RT[1][2] = tyz - twx;
  RT[2][0] = txz - twy;            RT[2][1] = tyz + twx;            RT[2][2] = T(1.0) - (txx + tyy);
  RT[0][3] = tr[0]; RT[1][3] = tr[1]; RT[2][3] = tr[2];
}

// decoded config bitmask (reference BundleAdjustment.h:28-79)
struct Config {
  int n_radial; bool tangential, refine_poses, robust, refine_points, ml_center_adj;
  int n_camera;  // 5 + n_radial + 2*tangential
  explicit Config(unsigned c)
      : n_radial(c & 0x3), tangential((c & 0x4) != 0), refine_poses((c & 0x100) != 0),
        robust((c & 0x200) != 0), refine_points((c & 0x400) != 0), ml_center_adj((c & 0x800) != 0) {
    n_camera = 5 + n_radial + (tangential ? 2 : 0);
  }
};

// reference OurCostFunctionBundle: ctor (:26-102) + operator_function<T> (:120-195).
// The three autodiff arities of Create() (:199-222):
//   view != null, point != null : <2,17,6,3>   poses and points are parameters
//   view != null, point == null : <2,17,6>     point held at `fixed_point`
//   view == null                : <2,17>       camera only, camera-frame point precomputed (:94-101)
struct ObsFunctor {
  Config cfg;
  double u, v, spx, spy, scale, mlx, mly;
  double fixed_point[3];
  double h_cam[3];  // RT * P precomputed when the view is constant (:99-100)

  ObsFunctor(unsigned config, double u_, double v_, double spx_, double spy_, double scale_,
             double mlx_, double mly_)
      : cfg(config), u(u_), v(v_), spx(spx_ / scale_), spy(spy_ / scale_), scale(scale_),
        mlx(mlx_), mly(mly_) {
    fixed_point[0] = fixed_point[1] = fixed_point[2] = 0.0;
    h_cam[0] = h_cam[1] = h_cam[2] = 0.0;
  }
  void set_fixed_point(const double* P) { for (int i = 0; i < 3; ++i) fixed_point[i] = P[i]; }
  void set_fixed_view(const double* view) {  // :94-101
    double RT[3][4];
    rigid_transform<double>(view, view + 3, RT);
    for (int i = 0; i < 3; ++i)
      h_cam[i] = RT[i][0] * fixed_point[0] + RT[i][1] * fixed_point[1] + RT[i][2] * fixed_point[2] + RT[i][3] * 1.0;
  }

  template <class T>
  bool operator()(const T* camera, const T* view, const T* point, T* residuals) const {
    T fL = camera[0];  if (fL < T(0.0)) fL = -fL;    // :123-128
    T bL0 = camera[1]; if (bL0 < T(0.0)) bL0 = -bL0;
    T B = camera[2];   if (B < T(0.0)) B = -B;
    T c_raw[2];                                      // :129-133
    c_raw[0] = (camera[3] + T(0.5)) * T(scale) - T(0.5);
    c_raw[1] = (camera[4] + T(0.5)) * T(scale) - T(0.5);
    if (c_raw[0] < T(0.0)) c_raw[0] = -c_raw[0];
    if (c_raw[1] < T(0.0)) c_raw[1] = -c_raw[1];
    const T* radial = cfg.n_radial > 0 ? camera + 5 : nullptr;             // :135-140
    const T* tangential = cfg.tangential ? camera + 5 + cfg.n_radial : nullptr;  // :142-146
    T pc[3];
    if (view != nullptr) {  // refinePoses (:163-173)
      T P[3];
      for (int i = 0; i < 3; ++i) P[i] = (point != nullptr) ? point[i] : T(fixed_point[i]);  // :148-159
      T RT[3][4];
      rigid_transform<T>(view, view + 3, RT);
      for (int i = 0; i < 3; ++i) pc[i] = RT[i][0] * P[0] + RT[i][1] * P[1] + RT[i][2] * P[2] + RT[i][3] * T(1.0);
    } else {                // :174-178
      for (int i = 0; i < 3; ++i) pc[i] = T(h_cam[i]);
    }
    T ml[2] = {T(mlx), T(mly)};
    T px, py;
    project_point<T>(px, py, pc, T(spx), T(spy), fL, bL0, B, c_raw, ml, radial, cfg.n_radial,
                     tangential, cfg.ml_center_adj);
    residuals[0] = px - T(u);  // :191-192
    residuals[1] = py - T(v);
    return true;
  }
};

// reference BundleAdjustment.h:262-267: r = (|P1-P2| - distance) / (sigma + 1e-6)
template <class T>
inline T distance_constraint(const T* p1, const T* p2, double distance, double sigma) {
  T d0 = p1[0] - p2[0], d1 = p1[1] - p2[1], d2 = p1[2] - p2[2];
  T s = d0 * d0 + d1 * d1 + d2 * d2;
  return (pow(s, 0.5) - T(distance)) / (T(sigma) + T(0.000001));
}

}  // namespace lo
