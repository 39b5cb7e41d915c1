// se3_jet.hpp — the reference's array-based SE3 helpers (src/lvio_fusion/include/lvio_fusion/ceres/base.hpp:26-141) templated on the
// device dual number DJet<N> (djet.hpp), for the COLD functors whose ambient Jacobians are taken exactly as the reference's
// autodiff does: the window's weak-constraint priors (prior_kernels.hip) and the loop-correction factors (loop_kernels.hip).
#pragma once
#include "djet.hpp"

namespace lvf {

// ceres::QuaternionProduct, Hamilton, [w,x,y,z]
template <typename T>
__device__ __forceinline__ void quat_product_wxyz(const T z[4], const T w[4], T zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}
// ceres::QuaternionRotatePoint (1.x expanded form): normalise, then the unit-quaternion polynomial; q = [w,x,y,z]
template <typename T>
__device__ __forceinline__ void quat_rotate_wxyz(const T q[4], const T pt[3], T out[3]) {
  const T scale = T(1.0) / jsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T u0 = scale * q[0], u1 = scale * q[1], u2 = scale * q[2], u3 = scale * q[3];
  const T t2 = u0 * u1, t3 = u0 * u2, t4 = u0 * u3, t5 = -(u1 * u1), t6 = u1 * u2, t7 = u1 * u3, t8 = -(u2 * u2), t9 = u2 * u3,
          t1 = -(u3 * u3);
  out[0] = 2.0 * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
  out[1] = 2.0 * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
  out[2] = 2.0 * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}
// base.hpp:26-31 (Eigen order x,y,z,w)
template <typename T>
__device__ __forceinline__ void eigen_quat_rotate(const T eq[4], const T pt[3], T out[3]) {
  const T q[4] = {eq[3], eq[0], eq[1], eq[2]};
  quat_rotate_wxyz(q, pt, out);
}
// base.hpp:49-55
template <typename T>
__device__ __forceinline__ void se3_inverse(const T a[7], T inv[7]) {
  inv[0] = -a[0]; inv[1] = -a[1]; inv[2] = -a[2]; inv[3] = a[3];
  const T ti[3] = {-a[4], -a[5], -a[6]};
  eigen_quat_rotate(inv, ti, inv + 4);
}
// base.hpp:57-78
template <typename T>
__device__ __forceinline__ void se3_product(const T A[7], const T B[7], T C[7]) {
  const T z[4] = {A[3], A[0], A[1], A[2]}, w[4] = {B[3], B[0], B[1], B[2]};
  T zw[4];
  quat_product_wxyz(z, w, zw);
  C[0] = zw[1]; C[1] = zw[2]; C[2] = zw[3]; C[3] = zw[0];
  T t[3];
  eigen_quat_rotate(A, B + 4, t);
  C[4] = A[4] + t[0]; C[5] = A[5] + t[1]; C[6] = A[6] + t[2];
}
// base.hpp:94-108, :134-141  (yaw, pitch, roll, x, y, z)
template <typename T>
__device__ __forceinline__ void se3_to_rpyxyz(const T rel[7], T out[6]) {
  const T q[4] = {rel[3], rel[0], rel[1], rel[2]};
  out[0] = jatan2(2.0 * (q[1] * q[2] + q[0] * q[3]), T(1.0) - 2.0 * (q[2] * q[2] + q[3] * q[3]));
  out[1] = jasin(2.0 * (q[0] * q[2] - q[1] * q[3]));
  out[2] = jatan2(2.0 * (q[2] * q[3] + q[0] * q[1]), T(1.0) - 2.0 * (q[1] * q[1] + q[2] * q[2]));
  out[3] = rel[4]; out[4] = rel[5]; out[5] = rel[6];
}

}  // namespace lvf
