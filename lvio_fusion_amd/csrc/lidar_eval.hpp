// lidar_eval.hpp — shared device math of the point-to-plane factors (LidarPlaneErrorRPZ / LidarPlaneErrorYXY,
// src/lvio_fusion/include/lvio_fusion/ceres/lidar_error.hpp:42-110), used by the materialised batch kernels
// (lidar_kernels.hip) and the on-device ICP solve (icp_kernels.hip).
#pragma once
#include "lvf_math.hpp"

namespace lvf {

struct LidarU {
  double u2[4], inv_n2, R2[9], t2[3];
  double dq_a[4], dq_b[4];  // d q2raw / d angle for the (up to) two angular parameters, x,y,z,w
  double R1[9];
  double w;
};

struct LidarArgs {
  double Twc1[7];
  double rpyxyz[6];
  double weight;
  int mode;
};

__device__ __forceinline__ void hamilton_xyzw(const double a[4], const double b[4], double o[4]) {
  // Eigen-order wrappers around the [w,x,y,z] Hamilton product (base.hpp:57-69)
  const double aw = a[3], ax = a[0], ay = a[1], az = a[2];
  const double bw = b[3], bx = b[0], by = b[1], bz = b[2];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by - ax * bz + ay * bw + az * bx;
  o[2] = aw * bz + ax * by - ay * bx + az * bw;
}

__device__ inline void derive_lidar(const LidarArgs& a, LidarU& U) {
  // RPYToQuaternion, base.hpp:110-121 : half angles z=yaw/2, y=pitch/2, x=roll/2
  const double hz = a.rpyxyz[0] / 2.0, hy = a.rpyxyz[1] / 2.0, hx = a.rpyxyz[2] / 2.0;
  const double cz = cos(hz), sz = sin(hz), cy = cos(hy), sy = sin(hy), cx = cos(hx), sx = sin(hx);
  double qr[4];  // x,y,z,w
  qr[3] = cz * cy * cx + sz * sy * sx;
  qr[0] = cz * cy * sx - sz * sy * cx;
  qr[1] = cz * sy * cx + sz * cy * sx;
  qr[2] = sz * cy * cx - cz * sy * sx;
  double dyaw[4], dpitch[4], droll[4];
  dyaw[3] = 0.5 * (-sz * cy * cx + cz * sy * sx);
  dyaw[0] = 0.5 * (-sz * cy * sx - cz * sy * cx);
  dyaw[1] = 0.5 * (-sz * sy * cx + cz * cy * sx);
  dyaw[2] = 0.5 * (cz * cy * cx + sz * sy * sx);
  dpitch[3] = 0.5 * (-cz * sy * cx + sz * cy * sx);
  dpitch[0] = 0.5 * (-cz * sy * sx - sz * cy * cx);
  dpitch[1] = 0.5 * (cz * cy * cx - sz * sy * sx);
  dpitch[2] = 0.5 * (-sz * sy * cx - cz * cy * sx);
  droll[3] = 0.5 * (-cz * cy * sx + sz * sy * cx);
  droll[0] = 0.5 * (cz * cy * cx + sz * sy * sx);
  droll[1] = 0.5 * (-cz * sy * sx + sz * cy * cx);
  droll[2] = 0.5 * (-sz * cy * sx - cz * sy * cx);
  PoseD P1;
  derive_pose(a.Twc1, P1);
  for (int k = 0; k < 9; ++k) U.R1[k] = P1.R[k];
  // SE3Product (base.hpp:71-78): q2 = q1 (x) qr on the RAW q1; t2 = t1 + R(q1/|q1|) tr
  double q2[4];
  hamilton_xyzw(a.Twc1, qr, q2);
  const double s = 1.0 / sqrt(q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3]);
  U.inv_n2 = s;
  for (int k = 0; k < 4; ++k) U.u2[k] = s * q2[k];
  rot_from_unit(U.u2, U.R2);
  double rt[3];
  mat3_mul_vec(P1.R, a.rpyxyz + 3, rt);
  U.t2[0] = a.Twc1[4] + rt[0]; U.t2[1] = a.Twc1[5] + rt[1]; U.t2[2] = a.Twc1[6] + rt[2];
  if (a.mode == 0) { hamilton_xyzw(a.Twc1, dpitch, U.dq_a); hamilton_xyzw(a.Twc1, droll, U.dq_b); }
  else { hamilton_xyzw(a.Twc1, dyaw, U.dq_a); for (int k = 0; k < 4; ++k) U.dq_b[k] = 0.0; }
  U.w = a.weight;
}

// one correspondence: residual and the three scalar derivatives
__device__ __forceinline__ void lidar_point(const LidarU& U, int mode, const double p[3], const double pa[3],
                                            const double nn[3], double& r, double J[3]) {
  double rp[3];
  mat3_mul_vec(U.R2, p, rp);
  const double d[3] = {rp[0] + U.t2[0] - pa[0], rp[1] + U.t2[1] - pa[1], rp[2] + U.t2[2] - pa[2]};
  r = U.w * (d[0] * nn[0] + d[1] * nn[1] + d[2] * nn[2]);
  const double fmp[3] = {rp[0] - p[0], rp[1] - p[1], rp[2] - p[2]};
  double g[4];
  row_times_drot_dq<false>(nn, U.u2, U.inv_n2, p, fmp, g);
  const double ja = U.w * (g[0] * U.dq_a[0] + g[1] * U.dq_a[1] + g[2] * U.dq_a[2] + g[3] * U.dq_a[3]);
  // translation columns: w * n . (R1 e_k)
  const double tx = U.w * (nn[0] * U.R1[0] + nn[1] * U.R1[3] + nn[2] * U.R1[6]);
  const double ty = U.w * (nn[0] * U.R1[1] + nn[1] * U.R1[4] + nn[2] * U.R1[7]);
  const double tz = U.w * (nn[0] * U.R1[2] + nn[1] * U.R1[5] + nn[2] * U.R1[8]);
  if (mode == 0) {
    const double jb = U.w * (g[0] * U.dq_b[0] + g[1] * U.dq_b[1] + g[2] * U.dq_b[2] + g[3] * U.dq_b[3]);
    J[0] = ja; J[1] = jb; J[2] = tz;
  } else {
    J[0] = ja; J[1] = tx; J[2] = ty;
  }
}


// LidarPlaneError ctor (lidar_error.hpp:13-18): unit normal of the plane through pa, pb, pc (Eigen cross + normalize)
__device__ __forceinline__ void plane_normal(const double pa[3], const double pb[3], const double pc[3], double n[3]) {
  const double ux = pa[0] - pb[0], uy = pa[1] - pb[1], uz = pa[2] - pb[2];
  const double vx = pa[0] - pc[0], vy = pa[1] - pc[1], vz = pa[2] - pc[2];
  double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
  const double z = nx * nx + ny * ny + nz * nz;
  if (z > 0.0) { const double s = sqrt(z); nx /= s; ny /= s; nz /= s; }
  n[0] = nx; n[1] = ny; n[2] = nz;
}

}  // namespace lvf
