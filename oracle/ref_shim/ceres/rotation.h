// ORACLE — TEST INFRASTRUCTURE ONLY.  Shim of the three upstream ceres/rotation.h primitives the reference calls
// (base.hpp:30 QuaternionRotatePoint, :63 QuaternionProduct, :82 DotProduct).  DECLARED upstream semantics:
// quaternions are [w, x, y, z]; QuaternionRotatePoint normalises by 1/sqrt(q.q) and then applies the Ceres-1.x expanded
// product form of UnitQuaternionRotatePoint.
#pragma once
#include "jet.h"

namespace ceres {

template <typename T>
inline void UnitQuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  const T t2 = q[0] * q[1];
  const T t3 = q[0] * q[2];
  const T t4 = q[0] * q[3];
  const T t5 = -q[1] * q[1];
  const T t6 = q[1] * q[2];
  const T t7 = q[1] * q[3];
  const T t8 = -q[2] * q[2];
  const T t9 = q[2] * q[3];
  const T t1 = -q[3] * q[3];
  result[0] = T(2) * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];  // NOLINT
  result[1] = T(2) * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];  // NOLINT
  result[2] = T(2) * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];  // NOLINT
}

template <typename T>
inline void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  // 'scale' is 1 / norm(q).
  const T scale = T(1) / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  // Make unit-norm version of q.
  const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
  UnitQuaternionRotatePoint(unit, pt, result);
}

template <typename T>
inline void QuaternionProduct(const T z[4], const T w[4], T zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

template <typename T>
inline T DotProduct(const T x[3], const T y[3]) { return (x[0] * y[0] + x[1] * y[1] + x[2] * y[2]); }

}  // namespace ceres
