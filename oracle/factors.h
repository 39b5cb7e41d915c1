// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).
// PARITY PINNED to the reference's own text: tests/test_oracle_ref.py compares this restatement BIT-FOR-BIT with oracle/_ref
// (the reference's ceres/{base,visual_error,lidar_error,pose_error}.hpp compiled unmodified, oracle/ref_driver.cpp) and with the
// committed reference outputs tests/golden/ref_v1.npz.
//
// factors.h — the reference cost functors' templated operator() restated on plain arrays.
// Instantiated on double (residual only) and on Jet<N> (residual + exact ambient Jacobian,
// = what ceres::AutoDiffCostFunction hands back to the solver).
#pragma once
#include "se3_ops.h"

namespace lvo {

// Camera = intrinsics + sensor->robot extrinsic (INC/sensor.h:41-44, INC/visual/camera.h:74)
struct Camera {
  double fx, fy, cx, cy;
  double extrinsic[7];  // Sophus order
};

// visual_error.hpp:10-23
template <typename T>
inline void Reproject(const T* pw, const T* Twc, const Camera& cam, T* px) {
  T e[7], e_i[7], Twc_i[7], pc[3], pb[3];
  Se3Inv(Twc, Twc_i);
  Se3Apply(Twc_i, pw, pb);
  CastFrom(cam.extrinsic, 7, e);
  Se3Inv(e, e_i);
  Se3Apply(e_i, pb, pc);
  T xp = pc[0] / pc[2];
  T yp = pc[1] / pc[2];
  px[0] = cam.fx * xp + cam.cx;
  px[1] = cam.fy * yp + cam.cy;
}

// visual_error.hpp:25-33
template <typename T>
inline void PixelToRobot(const T* ob, const T* inv_d, const Camera& cam, T* pb) {
  T d = T(1) / inv_d[0];
  T ps[3] = {T((ob[0] - cam.cx) / cam.fx) * d, T((ob[1] - cam.cy) / cam.fy) * d, d};
  T e[7];
  CastFrom(cam.extrinsic, 7, e);
  Se3Apply(e, ps, pb);
}

// visual_error.hpp:35-46
template <typename T>
inline void RobotToPixel(const T* pb, const Camera& cam, T* px) {
  T e[7], e_i[7], pc[3];
  CastFrom(cam.extrinsic, 7, e);
  Se3Inv(e, e_i);
  Se3Apply(e_i, pb, pc);
  T xp = pc[0] / pc[2];
  T yp = pc[1] / pc[2];
  px[0] = cam.fx * xp + cam.cx;
  px[1] = cam.fy * yp + cam.cy;
}

// PoseOnlyReprojectionError::operator()  visual_error.hpp:54-64     <2,7>
template <typename T>
inline void PoseOnlyResidual(const double ob_[2], const double pw_[3], const Camera& cam0, double weight,
                             const T* Twc, T* r) {
  T px[2];
  T pw[3] = {T(pw_[0]), T(pw_[1]), T(pw_[2])};
  T ob[2] = {T(ob_[0]), T(ob_[1])};
  Reproject(pw, Twc, cam0, px);
  r[0] = T(weight) * (px[0] - ob[0]);
  r[1] = T(weight) * (px[1] - ob[1]);
}

// TwoFrameReprojectionError::operator()  visual_error.hpp:84-96     <2,1,7,7>
// NOTE ctor argument order at the call site (backend.cpp:138): first_ob is the RIGHT-image
// observation in the birth frame and is lifted through `right` (= Camera 1); the current
// LEFT-image observation is compared in `left` (= Camera 0).
template <typename T>
inline void TwoFrameResidual(const double first_ob_[2], const double ob_[2], const Camera& left,
                             const Camera& right, double weight, const T* inv_d, const T* Twc1,
                             const T* Twc2, T* r) {
  T px[2], pw[3], pb[3];
  T first_ob[2] = {T(first_ob_[0]), T(first_ob_[1])};
  T ob2[2] = {T(ob_[0]), T(ob_[1])};
  PixelToRobot(first_ob, inv_d, right, pb);
  Se3Apply(Twc1, pb, pw);
  Reproject(pw, Twc2, left, px);
  r[0] = T(weight) * (px[0] - ob2[0]);
  r[1] = T(weight) * (px[1] - ob2[1]);
}

// TwoCameraReprojectionError::operator()  visual_error.hpp:115-126  <2,1>
template <typename T>
inline void TwoCameraResidual(const double left_ob_[2], const double right_ob_[2], const Camera& left,
                              const Camera& right, double weight, const T* inv_d, T* r) {
  T px[2], pb[3];
  T right_ob[2] = {T(right_ob_[0]), T(right_ob_[1])};
  T left_ob[2] = {T(left_ob_[0]), T(left_ob_[1])};
  PixelToRobot(right_ob, inv_d, right, pb);
  RobotToPixel(pb, left, px);
  r[0] = T(weight) * (px[0] - left_ob[0]);
  r[1] = T(weight) * (px[1] - left_ob[1]);
}

// LidarPlaneError ctor  lidar_error.hpp:13-18 : n = normalised (pa-pb) x (pa-pc)
inline void PlaneNormal(const double pa[3], const double pb[3], const double pc[3], double n[3]) {
  const double u[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
  const double v[3] = {pa[0] - pc[0], pa[1] - pc[1], pa[2] - pc[2]};
  // Eigen cross(): (u1 v2 - u2 v1, u2 v0 - u0 v2, u0 v1 - u1 v0)
  n[0] = u[1] * v[2] - u[2] * v[1];
  n[1] = u[2] * v[0] - u[0] * v[2];
  n[2] = u[0] * v[1] - u[1] * v[0];
  // Eigen normalize(): divide by sqrt(squaredNorm) when > 0
  const double z = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
  if (z > 0.0) { const double s = std::sqrt(z); n[0] /= s; n[1] /= s; n[2] /= s; }
}

// LidarPlaneError::operator()  lidar_error.hpp:20-31     <1,7>
template <typename T>
inline void LidarPlaneResidual(const double p_[3], const double pa_[3], const double n_[3], const T* Twc2, T* r) {
  T cp[3] = {T(p_[0]), T(p_[1]), T(p_[2])};
  T pa[3] = {T(pa_[0]), T(pa_[1]), T(pa_[2])};
  T nn[3] = {T(n_[0]), T(n_[1]), T(n_[2])};
  T lp[3], lp_pa[3];
  Se3Apply(Twc2, cp, lp);
  Sub3(lp, pa, lp_pa);
  r[0] = Dot3(lp_pa, nn);
}

// LidarPlaneErrorRPZ::operator()  lidar_error.hpp:48-63  <1,1,1,1>  params (pitch, roll, z)
// LidarPlaneErrorYXY::operator()  lidar_error.hpp:83-98  <1,1,1,1>  params (yaw, x, y)
// `rpyxyz_live` is the caller's live double[6] (lidar_error.hpp:52,87 read it at evaluate time).
template <typename T>
inline void LidarPlaneRpzResidual(const double p_[3], const double pa_[3], const double n_[3],
                                  const double Twc1_[7], const double* rpyxyz_live, double weight,
                                  const T* pitch, const T* roll, const T* z, T* r) {
  T Twc1[7], Twc2[7], rel[7], rpyxyz[6];
  CastFrom(rpyxyz_live, 6, rpyxyz);
  rpyxyz[1] = *pitch;
  rpyxyz[2] = *roll;
  rpyxyz[5] = *z;
  RpyxyzToSe3(rpyxyz, rel);
  CastFrom(Twc1_, 7, Twc1);
  Se3Mul(Twc1, rel, Twc2);
  LidarPlaneResidual(p_, pa_, n_, Twc2, r);
  r[0] = T(weight) * r[0];
}
template <typename T>
inline void LidarPlaneYxyResidual(const double p_[3], const double pa_[3], const double n_[3],
                                  const double Twc1_[7], const double* rpyxyz_live, double weight,
                                  const T* yaw, const T* x, const T* y, T* r) {
  T Twc1[7], Twc2[7], rel[7], rpyxyz[6];
  CastFrom(rpyxyz_live, 6, rpyxyz);
  rpyxyz[0] = *yaw;
  rpyxyz[3] = *x;
  rpyxyz[4] = *y;
  RpyxyzToSe3(rpyxyz, rel);
  CastFrom(Twc1_, 7, Twc1);
  Se3Mul(Twc1, rel, Twc2);
  LidarPlaneResidual(p_, pa_, n_, Twc2, r);
  r[0] = T(weight) * r[0];
}

// PoseGraphError::operator()  pose_error.hpp:24-38   <6,7,7>; target rpyxyz_ from the ctor :13-17
template <typename T>
inline void PoseGraphResidual(const double target_rpyxyz[6], double weight, double v, const T* Twc1,
                              const T* Twc2, T* r) {
  T inv1[7], rel[7], rpyxyz[6];
  Se3Inv(Twc1, inv1);
  Se3Mul(inv1, Twc2, rel);
  Se3ToRpyxyz(rel, rpyxyz);
  r[0] = T(v * weight) * (T(target_rpyxyz[0]) - rpyxyz[0]);
  r[1] = T(v * weight) * (T(target_rpyxyz[1]) - rpyxyz[1]);
  r[2] = T(v * weight) * (T(target_rpyxyz[2]) - rpyxyz[2]);
  r[3] = T(weight) * (T(target_rpyxyz[3]) - rpyxyz[3]);
  r[4] = T(10 * weight) * (T(target_rpyxyz[4]) - rpyxyz[4]);
  r[5] = T(10 * weight) * (T(target_rpyxyz[5]) - rpyxyz[5]);
}

// PoseError::operator()  pose_error.hpp:60-76   <6,7>
template <typename T>
inline void PosePriorResidual(const double origin_[7], double weight, double v, const T* pose, T* r) {
  T origin[7], origin_inv[7], rel[7], rpyxyz[6];
  CastFrom(origin_, 7, origin);
  Se3Inv(origin, origin_inv);
  Se3Mul(origin_inv, pose, rel);
  Se3ToRpyxyz(rel, rpyxyz);
  r[0] = T(v * weight) * rpyxyz[0];
  r[1] = T(v * weight) * rpyxyz[1];
  r[2] = T(v * weight) * rpyxyz[2];
  r[3] = T(weight) * rpyxyz[3];
  r[4] = T(weight) * rpyxyz[4];
  r[5] = T(weight) * rpyxyz[5];
}

// RError::operator()  pose_error.hpp:93-100   <4,7>
template <typename T>
inline void RErrorResidual(const double origin_[7], double weight, const T* pose, T* r) {
  for (int k = 0; k < 4; ++k) r[k] = T(weight) * (pose[k] - T(origin_[k]));
}

// TError::operator()  pose_error.hpp:117-122   <3,7>
template <typename T>
inline void TErrorResidual(const double p_[3], double weight, const T* pose, T* r) {
  for (int k = 0; k < 3; ++k) r[k] = T(weight) * (pose[4 + k] - T(p_[k]));
}

// RelocateRError::operator()  pose_error.hpp:197-214   <7,4>: the single parameter block is a quaternion r = (x,y,z,w);
// residual = relocated - SE3Product([r, 0], unrelocated), all seven SE3 coefficients (relocator.cpp:261)
template <typename T>
inline void RelocateRResidual(const double relocated_[7], const double unrelocated_[7], const T* r4, T* res) {
  T R[7] = {r4[0], r4[1], r4[2], r4[3], T(0), T(0), T(0)};
  T unrelocated[7];
  CastFrom(unrelocated_, 7, unrelocated);
  T R_unrelocated[7];
  Se3Mul(R, unrelocated, R_unrelocated);
  for (int k = 0; k < 7; ++k) res[k] = T(relocated_[k]) - R_unrelocated[k];
}

// PoseErrorRPZ::operator() pose_error.hpp:147-153 (params p,r,z; note residual order r,p,z)
// PoseErrorYXY::operator() pose_error.hpp:175-181
template <typename T>
inline void PriorRpzResidual(const double rpyxyz0[6], double weight, const T* p, const T* r_, const T* z, T* res) {
  res[0] = T(weight) * (r_[0] - T(rpyxyz0[2]));
  res[1] = T(weight) * (p[0] - T(rpyxyz0[1]));
  res[2] = T(weight) * (z[0] - T(rpyxyz0[5]));
}
template <typename T>
inline void PriorYxyResidual(const double rpyxyz0[6], double weight, const T* Y, const T* x, const T* y, T* res) {
  res[0] = T(weight) * (Y[0] - T(rpyxyz0[0]));
  res[1] = T(weight) * (x[0] - T(rpyxyz0[3]));
  res[2] = T(weight) * (y[0] - T(rpyxyz0[4]));
}

}  // namespace lvo
