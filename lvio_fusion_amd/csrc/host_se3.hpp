// host_se3.hpp — the handful of SE3 operations Mapping::Optimize / Mapping::Relocate perform on the HOST between device
// solves (src/lvio_fusion/src/mapping.cpp:154,164,264,277,299; src/utility.cpp:27-40): Sophus SE3d product / inverse on
// unit quaternions and the se32rpyxyz / rpyxyz2se3 pair.  Pose = [qx,qy,qz,qw,tx,ty,tz].  Orchestration glue, not a
// compute path.
#pragma once
#include <cmath>

namespace lvf {
namespace hse3 {

inline void normalize4(double q[4]) {
  const double s = 1.0 / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] *= s;
}
inline void rotate(const double q_in[4], const double p[3], double o[3]) {   // unit-normalised inside
  double q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
  normalize4(q);
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double cx = y * p[2] - z * p[1], cy = z * p[0] - x * p[2], cz = x * p[1] - y * p[0];
  const double dx = y * cz - z * cy, dy = z * cx - x * cz, dz = x * cy - y * cx;
  o[0] = p[0] + 2.0 * (w * cx + dx); o[1] = p[1] + 2.0 * (w * cy + dy); o[2] = p[2] + 2.0 * (w * cz + dz);
}
inline void quat_mul(const double a[4], const double b[4], double o[4]) {   // Hamilton, x,y,z,w storage
  const double aw = a[3], ax = a[0], ay = a[1], az = a[2], bw = b[3], bx = b[0], by = b[1], bz = b[2];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by - ax * bz + ay * bw + az * bx;
  o[2] = aw * bz + ax * by - ay * bx + az * bw;
}
inline void mul(const double A[7], const double B[7], double C[7]) {   // SE3d::operator*
  double q[4], t[3];
  quat_mul(A, B, q);
  normalize4(q);
  rotate(A, B + 4, t);
  for (int k = 0; k < 4; ++k) C[k] = q[k];
  for (int k = 0; k < 3; ++k) C[4 + k] = A[4 + k] + t[k];
}
inline void inv(const double A[7], double C[7]) {                      // SE3d::inverse
  double q[4] = {-A[0], -A[1], -A[2], A[3]};
  normalize4(q);
  const double mt[3] = {-A[4], -A[5], -A[6]};
  double t[3];
  rotate(q, mt, t);
  for (int k = 0; k < 4; ++k) C[k] = q[k];
  for (int k = 0; k < 3; ++k) C[4 + k] = t[k];
}
inline void to_rpyxyz(const double T[7], double r[6]) {                // se32rpyxyz, utility.cpp:27-33 (q = w,x,y,z below)
  const double w = T[3], x = T[0], y = T[1], z = T[2];
  r[0] = std::atan2(2.0 * (x * y + w * z), 1.0 - 2.0 * (y * y + z * z));
  r[1] = std::asin(2.0 * (w * y - x * z));
  r[2] = std::atan2(2.0 * (y * z + w * x), 1.0 - 2.0 * (x * x + y * y));
  r[3] = T[4]; r[4] = T[5]; r[5] = T[6];
}
inline void from_rpyxyz(const double r[6], double T[7]) {              // rpyxyz2se3, utility.cpp:35-40
  const double hz = r[0] / 2.0, hy = r[1] / 2.0, hx = r[2] / 2.0;
  const double cz = std::cos(hz), sz = std::sin(hz), cy = std::cos(hy), sy = std::sin(hy), cx = std::cos(hx), sx = std::sin(hx);
  double q[4];
  q[3] = cz * cy * cx + sz * sy * sx;
  q[0] = cz * cy * sx - sz * sy * cx;
  q[1] = cz * sy * cx + sz * cy * sx;
  q[2] = sz * cy * cx - cz * sy * sx;
  normalize4(q);                                                       // the SE3d(Quaterniond, Vector3d) ctor normalises
  for (int k = 0; k < 4; ++k) T[k] = q[k];
  T[4] = r[3]; T[5] = r[4]; T[6] = r[5];
}

}  // namespace hse3
}  // namespace lvf
