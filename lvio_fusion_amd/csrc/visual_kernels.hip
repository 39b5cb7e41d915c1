// visual_kernels.hip — batched residual + analytic Jacobian evaluation of the three reprojection factors
// (materialised, Ceres-layout outputs).  gfx950 / wave64.
//
// Replaces per-block ceres::AutoDiffCostFunction::Evaluate of
//   PoseOnlyReprojectionError  <2,7>      src/lvio_fusion/include/lvio_fusion/ceres/visual_error.hpp:48-76
//   TwoFrameReprojectionError  <2,1,7,7>  visual_error.hpp:78-107
//   TwoCameraReprojectionError <2,1>      visual_error.hpp:109-137
//
// Design (HBM-bound; PoseOnly = 152 algorithmic B/block of which 128 B are stores):
//   * one thread per residual block; observations are read as coalesced double2 (16 B/lane), indices as int32;
//   * every workgroup derives the window's pose blocks (unit quaternion, 1/|q|, R) ONCE into LDS — the
//     "LDS-staged SE3 pose blocks" — so per-thread pose access is an LDS broadcast, not a global gather + sqrt/div;
//   * Jacobian rows are produced in registers, transposed through a padded LDS tile (row stride 15 doubles:
//     conflict-free ds_write_b64) and written back as fully coalesced 16 B/lane stores in the exact
//     row-major num_residuals x block_size Ceres layout.
#include "lvf_internal.hpp"

namespace lvf {

constexpr int kBlock = 256;
constexpr int kMaxStagedKf = 64;
constexpr int kTileStride = 15;  // 14 Jacobian doubles + 1 pad

__device__ __forceinline__ void stage_poses(PoseD* s_pose, const double* __restrict__ poses, int n_kf) {
  if (n_kf <= kMaxStagedKf) {
    for (int k = threadIdx.x; k < n_kf; k += kBlock) derive_pose(poses + 7 * k, s_pose[k]);
    __syncthreads();
  }
}
__device__ __forceinline__ PoseD fetch_pose(const PoseD* s_pose, const double* __restrict__ poses, int n_kf, int kf) {
  if (n_kf <= kMaxStagedKf) return s_pose[kf];
  PoseD d;
  derive_pose(poses + 7 * kf, d);
  return d;
}

// cooperative, coalesced write-back of a [cnt][14] tile staged in LDS with row stride kTileStride
__device__ __forceinline__ void flush_tile14(const double* s_tile, double* __restrict__ out, int first, int cnt) {
  double2* dst = reinterpret_cast<double2*>(out + (size_t)first * 14);
  const int pairs = cnt * 7;
  for (int e2 = threadIdx.x; e2 < pairs; e2 += kBlock) {
    const int th = e2 / 7, k = 2 * (e2 - th * 7);
    const double* s = s_tile + th * kTileStride + k;
    dst[e2] = make_double2(s[0], s[1]);
  }
}

// ------------------------------------------------------------------------------------------ PoseOnly
template <bool WITH_J>
__global__ __launch_bounds__(kBlock) void k_pose_only(int n, int n_kf, const double2* __restrict__ ob,
                                                      const int* __restrict__ kf_idx, const int* __restrict__ pw_idx,
                                                      const double* __restrict__ pw, const double* __restrict__ poses,
                                                      const double* __restrict__ w_kf, const CamD cam,
                                                      double2* __restrict__ res, double* __restrict__ jac) {
  __shared__ PoseD s_pose[kMaxStagedKf];
  __shared__ double s_tile[WITH_J ? kBlock * kTileStride : 1];
  stage_poses(s_pose, poses, n_kf);
  const int first = blockIdx.x * kBlock;
  const int i = first + threadIdx.x;
  if (i < n) {
    const int kf = kf_idx[i];
    const int l = pw_idx[i];
    const double2 o = ob[i];
    const double w = w_kf[kf];
    const PoseD P = fetch_pose(s_pose, poses, n_kf, kf);
    const double d[3] = {pw[3 * l] - P.t[0], pw[3 * l + 1] - P.t[1], pw[3 * l + 2] - P.t[2]};
    double pb[3];
    mat3t_mul_vec(P.R, d, pb);
    double px[2], M[6];
    project_and_chain(cam, pb, w, px, M);
    res[i] = make_double2(w * (px[0] - o.x), w * (px[1] - o.y));
    if (WITH_J) {
      const double fmp[3] = {pb[0] - d[0], pb[1] - d[1], pb[2] - d[2]};
      double* row = s_tile + threadIdx.x * kTileStride;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        double q4[4], t3[3];
        row_times_drot_dq<true>(M + 3 * a, P.u, P.inv_n, d, fmp, q4);
        mat3_mul_vec(P.R, M + 3 * a, t3);  // (M_row R^T)^T = R M_row^T
        row[7 * a + 0] = q4[0]; row[7 * a + 1] = q4[1]; row[7 * a + 2] = q4[2]; row[7 * a + 3] = q4[3];
        row[7 * a + 4] = -t3[0]; row[7 * a + 5] = -t3[1]; row[7 * a + 6] = -t3[2];
      }
    }
  }
  if (WITH_J) {
    __syncthreads();
    const int cnt = min(kBlock, n - first);
    flush_tile14(s_tile, jac, first, cnt);
  }
}

// ------------------------------------------------------------------------------------------ TwoFrame
template <bool WITH_J>
__global__ __launch_bounds__(kBlock) void k_two_frame(int n, int n_kf, const double2* __restrict__ first_ob,
                                                      const double2* __restrict__ ob, const int* __restrict__ lm_idx,
                                                      const int* __restrict__ kf1_idx, const int* __restrict__ kf2_idx,
                                                      const double* __restrict__ inv_depth,
                                                      const double* __restrict__ poses, const double* __restrict__ w_kf,
                                                      const CamD left, const CamD right, double2* __restrict__ res,
                                                      double2* __restrict__ jd, double* __restrict__ j1,
                                                      double* __restrict__ j2) {
  __shared__ PoseD s_pose[kMaxStagedKf];
  __shared__ double s_tile[WITH_J ? kBlock * kTileStride : 1];
  stage_poses(s_pose, poses, n_kf);
  const int first = blockIdx.x * kBlock;
  const int i = first + threadIdx.x;
  const int cnt = min(kBlock, n - first);
  double J2row[14];
  if (i < n) {
    const int k1 = kf1_idx[i], k2 = kf2_idx[i];
    const double2 fo = first_ob[i], o = ob[i];
    const double rho = inv_depth[lm_idx[i]];
    const double w = w_kf[k2];
    const PoseD P1 = fetch_pose(s_pose, poses, n_kf, k1);
    const PoseD P2 = fetch_pose(s_pose, poses, n_kf, k2);
    // Pixel2Robot through the RIGHT camera (visual_error.hpp:25-33)
    const double dpt = 1.0 / rho;
    const double dir[3] = {(fo.x - right.cx) / right.fx, (fo.y - right.cy) / right.fy, 1.0};
    const double ps[3] = {dir[0] * dpt, dir[1] * dpt, dpt};
    double pb1[3];
    mat3_mul_vec(right.Re, ps, pb1);
    pb1[0] += right.te[0]; pb1[1] += right.te[1]; pb1[2] += right.te[2];
    double rp[3];
    mat3_mul_vec(P1.R, pb1, rp);                                   // R1 pb1
    const double pwd[3] = {rp[0] + P1.t[0], rp[1] + P1.t[1], rp[2] + P1.t[2]};
    const double dd[3] = {pwd[0] - P2.t[0], pwd[1] - P2.t[1], pwd[2] - P2.t[2]};
    double pb2[3];
    mat3t_mul_vec(P2.R, dd, pb2);
    double px[2], M[6];
    project_and_chain(left, pb2, w, px, M);
    res[i] = make_double2(w * (px[0] - o.x), w * (px[1] - o.y));
    if (WITH_J) {
      const double fmp2[3] = {pb2[0] - dd[0], pb2[1] - dd[1], pb2[2] - dd[2]};
      const double fmp1[3] = {rp[0] - pb1[0], rp[1] - pb1[1], rp[2] - pb1[2]};
      double rd[3];                                                  // R1 Re dir
      { double t[3]; mat3_mul_vec(right.Re, dir, t); mat3_mul_vec(P1.R, t, rd); }
      const double md2 = -(dpt * dpt);
      double* row = s_tile + threadIdx.x * kTileStride;
      double jdv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        double A[3], q4[4];
        mat3_mul_vec(P2.R, M + 3 * a, A);                            // d r_a / d pw
        row_times_drot_dq<true>(M + 3 * a, P2.u, P2.inv_n, dd, fmp2, q4);
        J2row[7 * a + 0] = q4[0]; J2row[7 * a + 1] = q4[1]; J2row[7 * a + 2] = q4[2]; J2row[7 * a + 3] = q4[3];
        J2row[7 * a + 4] = -A[0]; J2row[7 * a + 5] = -A[1]; J2row[7 * a + 6] = -A[2];
        row_times_drot_dq<false>(A, P1.u, P1.inv_n, pb1, fmp1, q4);
        row[7 * a + 0] = q4[0]; row[7 * a + 1] = q4[1]; row[7 * a + 2] = q4[2]; row[7 * a + 3] = q4[3];
        row[7 * a + 4] = A[0]; row[7 * a + 5] = A[1]; row[7 * a + 6] = A[2];
        jdv[a] = (A[0] * rd[0] + A[1] * rd[1] + A[2] * rd[2]) * md2;
      }
      jd[i] = make_double2(jdv[0], jdv[1]);
    }
  }
  if (WITH_J) {
    __syncthreads();
    flush_tile14(s_tile, j1, first, cnt);
    __syncthreads();
    if (i < n) {
      double* row = s_tile + threadIdx.x * kTileStride;
#pragma unroll
      for (int k = 0; k < 14; ++k) row[k] = J2row[k];
    }
    __syncthreads();
    flush_tile14(s_tile, j2, first, cnt);
  }
}

// ------------------------------------------------------------------------------------------ TwoCamera
template <bool WITH_J>
__global__ __launch_bounds__(kBlock) void k_two_camera(int n, const double2* __restrict__ left_ob,
                                                       const double2* __restrict__ right_ob,
                                                       const int* __restrict__ lm_idx, const int* __restrict__ kf_idx,
                                                       const double* __restrict__ inv_depth,
                                                       const double* __restrict__ w_kf, const CamD left,
                                                       const CamD right, double2* __restrict__ res,
                                                       double2* __restrict__ jac) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double2 lo = left_ob[i], ro = right_ob[i];
  const double rho = inv_depth[lm_idx[i]];
  const double w = 5.0 * w_kf[kf_idx[i]];                            // backend.cpp:123
  const double dpt = 1.0 / rho;
  const double dir[3] = {(ro.x - right.cx) / right.fx, (ro.y - right.cy) / right.fy, 1.0};
  const double ps[3] = {dir[0] * dpt, dir[1] * dpt, dpt};
  double pb[3];
  mat3_mul_vec(right.Re, ps, pb);
  pb[0] += right.te[0]; pb[1] += right.te[1]; pb[2] += right.te[2];
  double px[2], M[6];
  project_and_chain(left, pb, w, px, M);
  res[i] = make_double2(w * (px[0] - lo.x), w * (px[1] - lo.y));
  if (WITH_J) {
    double rd[3];
    mat3_mul_vec(right.Re, dir, rd);
    const double md2 = -(dpt * dpt);
    jac[i] = make_double2((M[0] * rd[0] + M[1] * rd[1] + M[2] * rd[2]) * md2,
                          (M[3] * rd[0] + M[4] * rd[1] + M[5] * rd[2]) * md2);
  }
}

// ------------------------------------------------------------------------------------------ launchers
static inline int grid_for(int n) { return (n + kBlock - 1) / kBlock; }

int launch_pose_only(lvf_batch* b, const lvf_state* st, bool want_j) {
  if (b->n == 0) return LVF_OK;
  hipStream_t s = b->ctx->stream;
  auto ob = reinterpret_cast<const double2*>(b->ob_a.p);
  auto res = reinterpret_cast<double2*>(b->res.p);
  if (want_j)
    hipLaunchKernelGGL(k_pose_only<true>, dim3(grid_for(b->n)), dim3(kBlock), 0, s, b->n, st->n_kf, ob, b->idx_a.p,
                       b->idx_b.p, b->table.p, st->poses.p, st->w_visual.p, b->cam_a, res, b->jac[0].p);
  else
    hipLaunchKernelGGL(k_pose_only<false>, dim3(grid_for(b->n)), dim3(kBlock), 0, s, b->n, st->n_kf, ob, b->idx_a.p,
                       b->idx_b.p, b->table.p, st->poses.p, st->w_visual.p, b->cam_a, res, (double*)nullptr);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

int launch_two_frame(lvf_batch* b, const lvf_state* st, bool want_j) {
  if (b->n == 0) return LVF_OK;
  hipStream_t s = b->ctx->stream;
  auto fo = reinterpret_cast<const double2*>(b->ob_a.p);
  auto ob = reinterpret_cast<const double2*>(b->ob_b.p);
  auto res = reinterpret_cast<double2*>(b->res.p);
  if (want_j)
    hipLaunchKernelGGL(k_two_frame<true>, dim3(grid_for(b->n)), dim3(kBlock), 0, s, b->n, st->n_kf, fo, ob, b->idx_a.p,
                       b->idx_b.p, b->idx_c.p, st->inv_depth.p, st->poses.p, st->w_visual.p, b->cam_a, b->cam_b, res,
                       reinterpret_cast<double2*>(b->jac[0].p), b->jac[1].p, b->jac[2].p);
  else
    hipLaunchKernelGGL(k_two_frame<false>, dim3(grid_for(b->n)), dim3(kBlock), 0, s, b->n, st->n_kf, fo, ob,
                       b->idx_a.p, b->idx_b.p, b->idx_c.p, st->inv_depth.p, st->poses.p, st->w_visual.p, b->cam_a,
                       b->cam_b, res, (double2*)nullptr, (double*)nullptr, (double*)nullptr);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

int launch_two_camera(lvf_batch* b, const lvf_state* st, bool want_j) {
  if (b->n == 0) return LVF_OK;
  hipStream_t s = b->ctx->stream;
  auto lo = reinterpret_cast<const double2*>(b->ob_a.p);
  auto ro = reinterpret_cast<const double2*>(b->ob_b.p);
  auto res = reinterpret_cast<double2*>(b->res.p);
  if (want_j)
    hipLaunchKernelGGL(k_two_camera<true>, dim3(grid_for(b->n)), dim3(kBlock), 0, s, b->n, lo, ro, b->idx_a.p,
                       b->idx_b.p, st->inv_depth.p, st->w_visual.p, b->cam_a, b->cam_b, res,
                       reinterpret_cast<double2*>(b->jac[0].p));
  else
    hipLaunchKernelGGL(k_two_camera<false>, dim3(grid_for(b->n)), dim3(kBlock), 0, s, b->n, lo, ro, b->idx_a.p,
                       b->idx_b.p, st->inv_depth.p, st->w_visual.p, b->cam_a, b->cam_b, res, (double2*)nullptr);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

}  // namespace lvf
