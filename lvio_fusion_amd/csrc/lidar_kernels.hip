// lidar_kernels.hip — point-to-plane factors of one scan-to-map sub-problem (3 scalar parameter blocks).
//
// Replaces LidarPlaneError ctor (unit normal)            lidar_error.hpp:13-18
//          LidarPlaneErrorRPZ::operator() <1,1,1,1>      lidar_error.hpp:48-63   params (pitch, roll, z)
//          LidarPlaneErrorYXY::operator() <1,1,1,1>      lidar_error.hpp:83-98   params (yaw, x, y)
// of src/lvio_fusion/include/lvio_fusion/ceres/.  Twc2 = Twc1 * RpyxyzToSE3(rpyxyz) is identical for every
// block of the problem, so one lane per workgroup derives it (and d q2 / d angle) into LDS and all lanes reuse it;
// per point only n.(dR2/dq2 p) remains.  Inputs are SoA (x[n],y[n],z[n]) => coalesced 8 B/lane loads.
#include "lvf_internal.hpp"

namespace lvf {

constexpr int kBlockL = 256;

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

__device__ void derive_lidar(const LidarArgs& a, LidarU& U) {
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

__global__ __launch_bounds__(kBlockL) void k_lidar_normals(int n, const double* __restrict__ pa,
                                                           const double* __restrict__ pb, const double* __restrict__ pc,
                                                           double* __restrict__ nrm) {
  const int i = blockIdx.x * kBlockL + threadIdx.x;
  if (i >= n) return;
  const double ux = pa[i] - pb[i], uy = pa[n + i] - pb[n + i], uz = pa[2 * n + i] - pb[2 * n + i];
  const double vx = pa[i] - pc[i], vy = pa[n + i] - pc[n + i], vz = pa[2 * n + i] - pc[2 * n + i];
  double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
  const double z = nx * nx + ny * ny + nz * nz;
  if (z > 0.0) { const double s = sqrt(z); nx /= s; ny /= s; nz /= s; }
  nrm[i] = nx; nrm[n + i] = ny; nrm[2 * n + i] = nz;
}

template <bool WITH_J>
__global__ __launch_bounds__(kBlockL) void k_lidar_plane(int n, const double* __restrict__ p,
                                                         const double* __restrict__ pa, const double* __restrict__ nrm,
                                                         const LidarArgs args, double* __restrict__ res,
                                                         double* __restrict__ j0, double* __restrict__ j1,
                                                         double* __restrict__ j2) {
  __shared__ LidarU U;
  if (threadIdx.x == 0) derive_lidar(args, U);
  __syncthreads();
  const int i = blockIdx.x * kBlockL + threadIdx.x;
  if (i >= n) return;
  const double pp[3] = {p[i], p[n + i], p[2 * n + i]};
  const double qa[3] = {pa[i], pa[n + i], pa[2 * n + i]};
  const double nn[3] = {nrm[i], nrm[n + i], nrm[2 * n + i]};
  double r, J[3];
  lidar_point(U, args.mode, pp, qa, nn, r, J);
  res[i] = r;
  if (WITH_J) { j0[i] = J[0]; j1[i] = J[1]; j2[i] = J[2]; }
}

int launch_lidar_normals(lvf_batch* b, const double* d_pb, const double* d_pc) {
  if (b->n == 0) return LVF_OK;
  hipLaunchKernelGGL(k_lidar_normals, dim3((b->n + kBlockL - 1) / kBlockL), dim3(kBlockL), 0, b->ctx->stream, b->n,
                     b->lpa.p, d_pb, d_pc, b->lnrm.p);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

int launch_lidar_plane(lvf_batch* b, const double* rpyxyz_host, bool want_j) {
  if (b->n == 0) return LVF_OK;
  LidarArgs a;
  std::memcpy(a.Twc1, b->Twc1, sizeof(a.Twc1));
  std::memcpy(a.rpyxyz, rpyxyz_host, sizeof(a.rpyxyz));
  a.weight = b->lidar_weight;
  a.mode = b->lidar_mode;
  const dim3 grid((b->n + kBlockL - 1) / kBlockL), blk(kBlockL);
  if (want_j)
    hipLaunchKernelGGL(k_lidar_plane<true>, grid, blk, 0, b->ctx->stream, b->n, b->lp.p, b->lpa.p, b->lnrm.p, a,
                       b->res.p, b->jac[0].p, b->jac[1].p, b->jac[2].p);
  else
    hipLaunchKernelGGL(k_lidar_plane<false>, grid, blk, 0, b->ctx->stream, b->n, b->lp.p, b->lpa.p, b->lnrm.p, a,
                       b->res.p, (double*)nullptr, (double*)nullptr, (double*)nullptr);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

}  // namespace lvf
