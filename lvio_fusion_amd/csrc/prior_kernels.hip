// prior_kernels.hip — the weak-constraint pose priors of a BA window, evaluated on device.
//
// Replaces per-block ceres::AutoDiffCostFunction::Evaluate of
//   PoseGraphError <6,7,7>   src/lvio_fusion/include/lvio_fusion/ceres/pose_error.hpp:10-53
//   PoseError      <6,7>     pose_error.hpp:55-86
//   RError         <4,7>     pose_error.hpp:88-110 (quaternion prior of the pose-graph problem, src/pose_graph.cpp:190-191)
// as Backend::BuildProblem adds them to frames with no IMU factor and < 20 near visual blocks
// (src/lvio_fusion/src/backend.cpp:164-178; weight 100, v 0).  Both chain SE3Inverse -> SE3Product -> SE3ToRpyxyz
// (include/lvio_fusion/ceres/base.hpp:41-55, :71-78, :94-141): atan2/asin of the UN-normalised quaternion product,
// so the ambient 7-D Jacobian is taken exactly as the reference's autodiff does — with dual numbers (djet.hpp).
// At most one block per keyframe: a latency-bound corner, one thread per block.
#include "lvf_internal.hpp"
#include "prior_eval.hpp"

namespace lvf {

// One thread per prior block (prior_eval.hpp).  Outputs: res [n][6]; ja, jb [n][6][7] row-major (ja is all-zero for PoseError blocks).
template <bool WITH_J>
__global__ __launch_bounds__(64) void k_pose_prior(int n, const int* __restrict__ kf_a, const int* __restrict__ kf_b,
                                                   const double* __restrict__ target, const double* __restrict__ weight,
                                                   const double* __restrict__ vv, const double* __restrict__ poses,
                                                   double* __restrict__ res, double* __restrict__ ja, double* __restrict__ jb) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  pose_prior_eval<WITH_J>(kf_a[i], kf_b[i], target + 7 * i, weight[i], vv[i], poses, res + 6 * i, WITH_J ? ja + (size_t)42 * i : nullptr, WITH_J ? jb + (size_t)42 * i : nullptr);
}

int launch_pose_prior(lvf_batch* b, const lvf_state* st, bool want_j) {
  if (b->n == 0) return LVF_OK;
  const dim3 grid((b->n + 63) / 64), blk(64);
  hipStream_t s = b->ctx->stream;
  if (want_j)
    hipLaunchKernelGGL(k_pose_prior<true>, grid, blk, 0, s, b->n, b->idx_a.p, b->idx_b.p, b->table.p, b->ob_a.p, b->ob_b.p,
                       st->poses.p, b->res.p, b->jac[0].p, b->jac[1].p);
  else
    hipLaunchKernelGGL(k_pose_prior<false>, grid, blk, 0, s, b->n, b->idx_a.p, b->idx_b.p, b->table.p, b->ob_a.p, b->ob_b.p,
                       st->poses.p, b->res.p, (double*)nullptr, (double*)nullptr);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

}  // namespace lvf

using namespace lvf;

extern "C" {

int lvf_pose_prior_create(lvf_ctx* ctx, int n, const int32_t* kf_a, const int32_t* kf_b, const double* target,
                          const double* weight, const double* v, lvf_batch** out) {
  LVF_REQUIRE(ctx && out, "lvf_pose_prior_create: null argument");
  LVF_REQUIRE(n >= 0, "lvf_pose_prior_create: negative size");
  LVF_REQUIRE(n == 0 || (kf_a && kf_b && target && weight && v), "lvf_pose_prior_create: null input array");
  int32_t mx = -1;
  for (int i = 0; i < n; ++i) {
    LVF_REQUIRE(kf_b[i] >= 0, "lvf_pose_prior_create: kf_b[%d] = %d is negative", i, kf_b[i]);
    LVF_REQUIRE(kf_a[i] != kf_b[i], "lvf_pose_prior_create: block %d links keyframe %d to itself", i, kf_b[i]);
    mx = std::max(mx, std::max(kf_a[i], kf_b[i]));
  }
  LVF_TRY(lvf::enter(ctx));
  auto* b = new lvf_batch();
  b->ctx = ctx; b->kind = LVF_K_POSE_PRIOR; b->n = n; b->n_res = 6; b->n_blocks = 2; b->block_size[0] = 7; b->block_size[1] = 7;
  b->min_n_kf = mx + 1;
  b->host_kf1.assign(kf_a, kf_a + n); b->host_kf2.assign(kf_b, kf_b + n);
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = b->idx_a.upload(kf_a, n, s)) || (rc = b->idx_b.upload(kf_b, n, s)) || (rc = b->table.upload(target, (size_t)7 * n, s)) ||
      (rc = b->ob_a.upload(weight, n, s)) || (rc = b->ob_b.upload(v, n, s)) || (rc = b->res.alloc((size_t)6 * n)) ||
      (rc = b->jac[0].alloc((size_t)42 * n)) || (rc = b->jac[1].alloc((size_t)42 * n))) { delete b; return rc; }
  LVF_HIP(hipStreamSynchronize(s));
  *out = b;
  return LVF_OK;
}

// PoseGraphError ctor (pose_error.hpp:13-17): rpyxyz_ = SE3ToRpyxyz(last_pose^-1 * pose).  A constructor-time
// constant the reference computes on the host with Sophus; done here in plain C so a caller without Sophus gets
// the identical target.  (Host arithmetic of a constant, not a fallback of any device path.)
int lvf_relative_rpyxyz(const double* last_pose, const double* pose, double* rpyxyz6) {
  LVF_REQUIRE(last_pose && pose && rpyxyz6, "lvf_relative_rpyxyz: null argument");
  auto rot = [](const double q[4], const double p[3], double o[3]) {   // q = x,y,z,w (normalised inside)
    const double s = 1.0 / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = s * q[0], y = s * q[1], z = s * q[2], w = s * q[3];
    const double cx = y * p[2] - z * p[1], cy = z * p[0] - x * p[2], cz = x * p[1] - y * p[0];
    const double dx = y * cz - z * cy, dy = z * cx - x * cz, dz = x * cy - y * cx;
    o[0] = p[0] + 2.0 * (w * cx + dx); o[1] = p[1] + 2.0 * (w * cy + dy); o[2] = p[2] + 2.0 * (w * cz + dz);
  };
  // Sophus SE3d::inverse(): (q^-1, -(q^-1 * t)) with a unit quaternion; operator*: (q1 q2, t1 + q1 * t2)
  double n1 = std::sqrt(last_pose[0] * last_pose[0] + last_pose[1] * last_pose[1] + last_pose[2] * last_pose[2] + last_pose[3] * last_pose[3]);
  double n2 = std::sqrt(pose[0] * pose[0] + pose[1] * pose[1] + pose[2] * pose[2] + pose[3] * pose[3]);
  LVF_REQUIRE(n1 > 0.0 && n2 > 0.0, "lvf_relative_rpyxyz: zero quaternion");
  const double qi[4] = {-last_pose[0] / n1, -last_pose[1] / n1, -last_pose[2] / n1, last_pose[3] / n1};
  const double q2[4] = {pose[0] / n2, pose[1] / n2, pose[2] / n2, pose[3] / n2};
  const double mt[3] = {-last_pose[4], -last_pose[5], -last_pose[6]};
  double ti[3], t2r[3];
  rot(qi, mt, ti);
  rot(qi, pose + 4, t2r);
  // Hamilton product qi (x) q2, components w,x,y,z
  const double aw = qi[3], ax = qi[0], ay = qi[1], az = qi[2], bw = q2[3], bx = q2[0], by = q2[1], bz = q2[2];
  const double w = aw * bw - ax * bx - ay * by - az * bz, x = aw * bx + ax * bw + ay * bz - az * by,
               y = aw * by - ax * bz + ay * bw + az * bx, z = aw * bz + ax * by - ay * bx + az * bw;
  rpyxyz6[0] = std::atan2(2.0 * (x * y + w * z), 1.0 - 2.0 * (y * y + z * z));
  rpyxyz6[1] = std::asin(2.0 * (w * y - x * z));
  rpyxyz6[2] = std::atan2(2.0 * (y * z + w * x), 1.0 - 2.0 * (x * x + y * y));
  rpyxyz6[3] = ti[0] + t2r[0]; rpyxyz6[4] = ti[1] + t2r[1]; rpyxyz6[5] = ti[2] + t2r[2];
  return LVF_OK;
}

}  // extern "C"
