// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).
// PARITY PINNED to the reference's own text: tests/test_oracle_ref.py compares this restatement BIT-FOR-BIT with oracle/_ref
// (the reference's ceres/{base,visual_error,lidar_error,pose_error}.hpp compiled unmodified, oracle/ref_driver.cpp) and with the
// committed reference outputs tests/golden/ref_v1.npz.
//
// se3_ops.h — array-based quaternion / SE3 algebra, templated on the scalar
// (double, float, Jet<N>).  Restates
//   src/lvio_fusion/include/lvio_fusion/ceres/base.hpp:10-150   (reference helpers)
// and the three upstream-Ceres primitives those helpers call (Ceres is an un-vendored,
// un-pinned dependency: src/lvio_fusion/CMakeLists.txt:26; README names ROS Kinetic…Noetic,
// i.e. Ceres 1.12–2.0).  DECLARED upstream semantics:
//   * QuaternionRotatePoint(q[w,x,y,z], p): scale = 1/sqrt(q·q); rotate by the unit
//     quaternion with the Ceres-1.x expanded product form (t2..t9,t1 below);
//   * QuaternionProduct: Hamilton product, [w,x,y,z];
//   * DotProduct: x0*y0 + x1*y1 + x2*y2, left to right.
// Pose layout everywhere: Sophus SE3d::data() = [qx,qy,qz,qw,tx,ty,tz].
#pragma once
#include "jet.h"

namespace lvo {

// ---- upstream Ceres rotation.h primitives (published algorithm, restated) ----
template <typename T>
inline void UnitQuatRotate_wxyz(const T q[4], const T pt[3], T out[3]) {
  const T t2 = q[0] * q[1];
  const T t3 = q[0] * q[2];
  const T t4 = q[0] * q[3];
  const T t5 = -q[1] * q[1];
  const T t6 = q[1] * q[2];
  const T t7 = q[1] * q[3];
  const T t8 = -q[2] * q[2];
  const T t9 = q[2] * q[3];
  const T t1 = -q[3] * q[3];
  out[0] = T(2) * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
  out[1] = T(2) * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
  out[2] = T(2) * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}

template <typename T>
inline void QuatRotate_wxyz(const T q[4], const T pt[3], T out[3]) {
  const T scale = T(1) / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
  UnitQuatRotate_wxyz(unit, pt, out);
}

template <typename T>
inline void QuatProduct_wxyz(const T z[4], const T w[4], T zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

template <typename T>
inline T Dot3(const T x[3], const T y[3]) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; }

// ---- reference helpers: base.hpp ----
// base.hpp:10-24
template <typename T> inline void Sub3(const T A[3], const T B[3], T C[3]) { C[0] = A[0] - B[0]; C[1] = A[1] - B[1]; C[2] = A[2] - B[2]; }
template <typename T> inline void Add3(const T A[3], const T B[3], T C[3]) { C[0] = A[0] + B[0]; C[1] = A[1] + B[1]; C[2] = A[2] + B[2]; }

// base.hpp:26-31  (Eigen [x,y,z,w] -> Ceres [w,x,y,z], then normalising rotate)
template <typename T>
inline void EigenQuatRotate(const T eq[4], const T pt[3], T out[3]) {
  const T q[4] = {eq[3], eq[0], eq[1], eq[2]};
  QuatRotate_wxyz(q, pt, out);
}

// base.hpp:33-38
template <typename T>
inline void Se3Apply(const T se3[7], const T pt[3], T out[3]) {
  EigenQuatRotate(se3, pt, out);
  Add3(out, se3 + 4, out);
}

// base.hpp:40-47  (conjugate, NOT a true inverse for non-unit q)
template <typename T>
inline void EigenQuatConj(const T eq[4], T out[4]) { out[0] = -eq[0]; out[1] = -eq[1]; out[2] = -eq[2]; out[3] = eq[3]; }

// base.hpp:49-55
template <typename T>
inline void Se3Inv(const T se3[7], T inv[7]) {
  EigenQuatConj(se3, inv);
  T tneg[3] = {-se3[4], -se3[5], -se3[6]};
  EigenQuatRotate(inv, tneg, inv + 4);
}

// base.hpp:57-69
template <typename T>
inline void EigenQuatMul(const T ez[4], const T ew[4], T ezw[4]) {
  const T z[4] = {ez[3], ez[0], ez[1], ez[2]};
  const T w[4] = {ew[3], ew[0], ew[1], ew[2]};
  T zw[4];
  QuatProduct_wxyz(z, w, zw);
  ezw[0] = zw[1]; ezw[1] = zw[2]; ezw[2] = zw[3]; ezw[3] = zw[0];
}

// base.hpp:71-78
template <typename T>
inline void Se3Mul(const T A[7], const T B[7], T C[7]) {
  EigenQuatMul(A, B, C);
  T t[3];
  EigenQuatRotate(A, B + 4, t);
  Add3(A + 4, t, C + 4);
}

// base.hpp:93-108  RPY is Z-Y-X with rpy[0] = yaw; q = [w,x,y,z]
template <typename T>
inline void QuatToRpy_wxyz(const T* q, T* rpy) {
  rpy[0] = atan2(T(2) * (q[1] * q[2] + q[0] * q[3]), T(1) - T(2) * (q[2] * q[2] + q[3] * q[3]));
  rpy[1] = asin(T(2) * (q[0] * q[2] - q[1] * q[3]));
  rpy[2] = atan2(T(2) * (q[2] * q[3] + q[0] * q[1]), T(1) - T(2) * (q[1] * q[1] + q[2] * q[2]));
}
template <typename T>
inline void EigenQuatToRpy(const T* eq, T* rpy) {
  const T q[4] = {eq[3], eq[0], eq[1], eq[2]};
  QuatToRpy_wxyz(q, rpy);
}

// base.hpp:110-132
template <typename T>
inline void RpyToQuat_wxyz(const T* rpy, T* q) {
  T z = rpy[0] / T(2), y = rpy[1] / T(2), x = rpy[2] / T(2);
  T c_z = cos(z), s_z = sin(z);
  T c_y = cos(y), s_y = sin(y);
  T c_x = cos(x), s_x = sin(x);
  q[0] = c_z * c_y * c_x + s_z * s_y * s_x;
  q[1] = c_z * c_y * s_x - s_z * s_y * c_x;
  q[2] = c_z * s_y * c_x + s_z * c_y * s_x;
  q[3] = s_z * c_y * c_x - c_z * s_y * s_x;
}
template <typename T>
inline void RpyToEigenQuat(const T* rpy, T* eq) {
  T q[4];
  RpyToQuat_wxyz(rpy, q);
  eq[0] = q[1]; eq[1] = q[2]; eq[2] = q[3]; eq[3] = q[0];
}

// base.hpp:134-150
template <typename T>
inline void Se3ToRpyxyz(const T* rel, T* rpyxyz) {
  EigenQuatToRpy(rel, rpyxyz);
  rpyxyz[3] = rel[4]; rpyxyz[4] = rel[5]; rpyxyz[5] = rel[6];
}
template <typename T>
inline void RpyxyzToSe3(const T* rpyxyz, T* rel) {
  RpyToEigenQuat(rpyxyz, rel);
  rel[4] = rpyxyz[3]; rel[5] = rpyxyz[4]; rel[6] = rpyxyz[5];
}

// base.hpp:86-92
template <typename T>
inline void CastFrom(const double* raw, int n, T* out) { for (int i = 0; i < n; ++i) out[i] = T(raw[i]); }

}  // namespace lvo
