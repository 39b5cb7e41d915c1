// lvf_math.hpp — device-side SE3 / quaternion algebra with hand-derived analytic Jacobians (gfx950).
//
// The reference differentiates its functors with ceres::AutoDiffCostFunction (Jets) through
// ceres::QuaternionRotatePoint, which NORMALISES the quaternion internally
// (src/lvio_fusion/include/lvio_fusion/ceres/base.hpp:26-38).  The ambient 7-D Jacobians therefore carry the
// projector d(q/|q|)/dq = (I - u u^T)/|q|; everything here reproduces that exactly, in closed form:
//
//   h(u,p) = R(u) p   = p + 2 [  w (v x p) + v x (v x p) ]     (u = (v,w) unit, the Ceres polynomial)
//   g(u,p) = R(u)^T p = p + 2 [ -w (v x p) + v x (v x p) ]
//   dh/dw =  2 (v x p)          dh/dv = 2 [ -w [p]x + (v.p) I + v p^T - 2 p v^T ]
//   dg/dw = -2 (v x p)          dg/dv = 2 [  w [p]x + (v.p) I + v p^T - 2 p v^T ]
//   d/dq f(q/|q|) = (F_u - (F_u u) u^T) / |q|        and  F_u u = 2 (f - p)  (f quadratic in u)
#pragma once
#include <hip/hip_runtime.h>

namespace lvf {

struct PoseD {      // derived per-pose data, staged in LDS per workgroup
  double u[4];      // unit quaternion x,y,z,w
  double inv_n;     // 1/|q|
  double t[3];
  double R[9];      // body -> world, row-major
};

struct CamD {       // derived camera constants (host-prepared at batch creation)
  double fx, fy, cx, cy;
  double E[9];      // R(extrinsic)^T : robot -> sensor rotation, row-major
  double c0[3];     // -E * t_extrinsic
  double Re[9];     // R(extrinsic)   : sensor -> robot
  double te[3];
};

__device__ __forceinline__ void rot_from_unit(const double u[4], double R[9]) {
  const double x = u[0], y = u[1], z = u[2], w = u[3];
  const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  R[0] = 1.0 - 2.0 * (yy + zz); R[1] = 2.0 * (xy - wz);       R[2] = 2.0 * (xz + wy);
  R[3] = 2.0 * (xy + wz);       R[4] = 1.0 - 2.0 * (xx + zz); R[5] = 2.0 * (yz - wx);
  R[6] = 2.0 * (xz - wy);       R[7] = 2.0 * (yz + wx);       R[8] = 1.0 - 2.0 * (xx + yy);
}

__device__ __forceinline__ void derive_pose(const double* __restrict__ pose, PoseD& d) {
  const double qx = pose[0], qy = pose[1], qz = pose[2], qw = pose[3];
  const double s = 1.0 / sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  d.inv_n = s;
  d.u[0] = s * qx; d.u[1] = s * qy; d.u[2] = s * qz; d.u[3] = s * qw;
  d.t[0] = pose[4]; d.t[1] = pose[5]; d.t[2] = pose[6];
  rot_from_unit(d.u, d.R);
}

__device__ __forceinline__ void mat3_mul_vec(const double M[9], const double v[3], double o[3]) {
  o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
__device__ __forceinline__ void mat3t_mul_vec(const double M[9], const double v[3], double o[3]) {
  o[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  o[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  o[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
}

// a (1x3 row) times d f(q/|q|)/dq (3x4) for f = R^T p (inverse=true) or R p (inverse=false).
// f_minus_p = f - p.  Writes the 1x4 row `out` (x,y,z,w order).  This is the only place the projector lives.
template <bool INVERSE>
__device__ __forceinline__ void row_times_drot_dq(const double a[3], const double u[4], double inv_n,
                                                   const double p[3], const double f_minus_p[3], double out[4]) {
  const double vx = u[0], vy = u[1], vz = u[2], w = u[3];
  // c = v x p
  const double cx = vy * p[2] - vz * p[1], cy = vz * p[0] - vx * p[2], cz = vx * p[1] - vy * p[0];
  const double sgn = INVERSE ? -1.0 : 1.0;
  // a . dF/dw = sgn * 2 a.(v x p)
  const double Fw = sgn * 2.0 * (a[0] * cx + a[1] * cy + a[2] * cz);
  // a^T dF/dv = 2 [ -sgn*w (a^T [p]x) + (v.p) a^T + (a.v) p^T - 2 (a.p) v^T ] ;  a^T [p]x = (p x a)^T * (-1) ... :
  // ([p]x)_{ij} = -eps_{ijk} p_k, so (a^T [p]x)_j = sum_i a_i (-eps_{ijk} p_k) = (a x p)_j
  const double axp_x = a[1] * p[2] - a[2] * p[1], axp_y = a[2] * p[0] - a[0] * p[2], axp_z = a[0] * p[1] - a[1] * p[0];
  const double vp = vx * p[0] + vy * p[1] + vz * p[2];
  const double av = a[0] * vx + a[1] * vy + a[2] * vz;
  const double ap = a[0] * p[0] + a[1] * p[1] + a[2] * p[2];
  const double Fx = 2.0 * (-sgn * w * axp_x + vp * a[0] + av * p[0] - 2.0 * ap * vx);
  const double Fy = 2.0 * (-sgn * w * axp_y + vp * a[1] + av * p[1] - 2.0 * ap * vy);
  const double Fz = 2.0 * (-sgn * w * axp_z + vp * a[2] + av * p[2] - 2.0 * ap * vz);
  // projector: (F - (F u) u^T) / |q| with F u = 2 a.(f - p)
  const double Fu = 2.0 * (a[0] * f_minus_p[0] + a[1] * f_minus_p[1] + a[2] * f_minus_p[2]);
  out[0] = (Fx - Fu * vx) * inv_n;
  out[1] = (Fy - Fu * vy) * inv_n;
  out[2] = (Fz - Fu * vz) * inv_n;
  out[3] = (Fw - Fu * w) * inv_n;
}

// Pinhole projection of a sensor-frame point and M = w * dpix/dpc * E  (2x3, d residual / d robot-frame point)
__device__ __forceinline__ void project_and_chain(const CamD& cam, const double pb[3], double w, double px[2], double M[6]) {
  const double pcx = cam.E[0] * pb[0] + cam.E[1] * pb[1] + cam.E[2] * pb[2] + cam.c0[0];
  const double pcy = cam.E[3] * pb[0] + cam.E[4] * pb[1] + cam.E[5] * pb[2] + cam.c0[1];
  const double pcz = cam.E[6] * pb[0] + cam.E[7] * pb[1] + cam.E[8] * pb[2] + cam.c0[2];
  const double iz = 1.0 / pcz;
  const double xp = pcx * iz, yp = pcy * iz;
  px[0] = cam.fx * xp + cam.cx;
  px[1] = cam.fy * yp + cam.cy;
  const double a = w * cam.fx * iz, b = w * cam.fy * iz;
  // row0 = a * (E row0 - xp * E row2), row1 = b * (E row1 - yp * E row2)
  M[0] = a * (cam.E[0] - xp * cam.E[6]); M[1] = a * (cam.E[1] - xp * cam.E[7]); M[2] = a * (cam.E[2] - xp * cam.E[8]);
  M[3] = b * (cam.E[3] - yp * cam.E[6]); M[4] = b * (cam.E[4] - yp * cam.E[7]); M[5] = b * (cam.E[5] - yp * cam.E[8]);
}

// Huber / Trivial corrector scale sqrt(rho'(s)) (rho'' <= 0 branch of Ceres' Corrector) and rho(s)
__device__ __forceinline__ double robust_scale(double a, double s, double& rho) {
  if (a > 0.0 && s > a * a) {
    const double r = sqrt(s);
    rho = 2.0 * a * r - a * a;
    return sqrt(a / r);
  }
  rho = s;
  return 1.0;
}

// EigenQuaternionParameterization plus-Jacobian applied to a 1x4 ambient row (x,y,z,w): out = row * P(q), 1x3
__device__ __forceinline__ void quat_row_to_local(const double row[4], const double q[4], double out[3]) {
  out[0] = row[0] * q[3] - row[1] * q[2] + row[2] * q[1] - row[3] * q[0];
  out[1] = row[0] * q[2] + row[1] * q[3] - row[2] * q[0] - row[3] * q[1];
  out[2] = -row[0] * q[1] + row[1] * q[0] + row[2] * q[3] - row[3] * q[2];
}

}  // namespace lvf
