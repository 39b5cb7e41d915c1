// lidar_kernels.hip — point-to-plane factors of one scan-to-map sub-problem (3 scalar parameter blocks).
//
// Replaces LidarPlaneError ctor (unit normal)            lidar_error.hpp:13-18
//          LidarPlaneErrorRPZ::operator() <1,1,1,1>      lidar_error.hpp:48-63   params (pitch, roll, z)
//          LidarPlaneErrorYXY::operator() <1,1,1,1>      lidar_error.hpp:83-98   params (yaw, x, y)
// of src/lvio_fusion/include/lvio_fusion/ceres/.  Twc2 = Twc1 * RpyxyzToSE3(rpyxyz) is identical for every
// block of the problem, so one lane per workgroup derives it (and d q2 / d angle) into LDS and all lanes reuse it;
// per point only n.(dR2/dq2 p) remains.  Inputs are SoA (x[n],y[n],z[n]) => coalesced 8 B/lane loads.
#include "lidar_eval.hpp"
#include "lvf_internal.hpp"

namespace lvf {

constexpr int kBlockL = 256;

__global__ __launch_bounds__(kBlockL) void k_lidar_normals(int n, const double* __restrict__ pa,
                                                           const double* __restrict__ pb, const double* __restrict__ pc,
                                                           double* __restrict__ nrm) {
  const int i = blockIdx.x * kBlockL + threadIdx.x;
  if (i >= n) return;
  const double a3[3] = {pa[i], pa[n + i], pa[2 * n + i]}, b3[3] = {pb[i], pb[n + i], pb[2 * n + i]}, c3[3] = {pc[i], pc[n + i], pc[2 * n + i]};
  double nn[3];
  plane_normal(a3, b3, c3, nn);
  nrm[i] = nn[0]; nrm[n + i] = nn[1]; nrm[2 * n + i] = nn[2];
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
