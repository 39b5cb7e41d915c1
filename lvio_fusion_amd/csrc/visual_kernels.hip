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
#include <algorithm>
#include <cstdlib>
#include "factor_eval.hpp"
#include "lvf_internal.hpp"

namespace lvf {

constexpr int kBlock = 256;
constexpr int kTileStride = 15;  // 14 Jacobian doubles + 1 pad

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
  stage_poses<kBlock>(s_pose, poses, n_kf);
  const int first = blockIdx.x * kBlock;
  const int i = first + threadIdx.x;
  if (i < n) {
    const int kf = kf_idx[i];
    const int l = pw_idx[i];
    const double2 o = ob[i];
    const PoseD P = fetch_pose(s_pose, poses, n_kf, kf);
    const double pwl[3] = {pw[3 * l], pw[3 * l + 1], pw[3 * l + 2]};
    double r[2], J[14];
    eval_pose_only<WITH_J>(P, cam, o.x, o.y, pwl, w_kf[kf], r, J);
    res[i] = make_double2(r[0], r[1]);
    if (WITH_J) {
      double* row = s_tile + threadIdx.x * kTileStride;
#pragma unroll
      for (int k = 0; k < 14; ++k) row[k] = J[k];
    }
  }
  if (WITH_J) {
    __syncthreads();
    flush_tile14(s_tile, jac, first, min(kBlock, n - first));
  }
}

// Materialised r + J, the bench headline: write-once outputs dominate (128 of 152 B/block), so
//   * outputs leave with NON-TEMPORAL 16-B stores (measured 15.5 -> 13.1 us at 500 k blocks: the stream no longer
//     allocates in L2/MALL; nt LOADS of the inputs measured slower and are not used),
//   * the grid is persistent (<= 4 workgroups per CU, each walking the same number of 256-block tiles round-robin) so the
//     pose staging is paid once per workgroup; concurrently active workgroups write ADJACENT 128-B aligned tiles (an even
//     contiguous split per workgroup measured 20 % slower: unaligned range starts turn wave stores into partial lines).
typedef double d2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store2_nt(double2* p, double2 v) {
  d2v_t t = {v.x, v.y};
  __builtin_nontemporal_store(t, reinterpret_cast<d2v_t*>(p));
}
__global__ __launch_bounds__(kBlock) void k_pose_only_rj(int n, int n_kf, const double2* __restrict__ ob,
                                                         const int* __restrict__ kf_idx, const int* __restrict__ pw_idx,
                                                         const double* __restrict__ pw, const double* __restrict__ poses,
                                                         const double* __restrict__ w_kf, const CamD cam,
                                                         double2* __restrict__ res, double* __restrict__ jac) {
  __shared__ PoseD s_pose[kMaxStagedKf];
  __shared__ double s_tile[kBlock * kTileStride];
  stage_poses<kBlock>(s_pose, poses, n_kf);
  const int ntiles = (n + kBlock - 1) / kBlock;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int first = tile * kBlock;
    const int i = first + threadIdx.x;
    if (i < n) {
      const int kf = kf_idx[i];
      const int l = pw_idx[i];
      const double2 o = ob[i];
      const PoseD P = fetch_pose(s_pose, poses, n_kf, kf);
      const double pwl[3] = {pw[3 * l], pw[3 * l + 1], pw[3 * l + 2]};
      double r[2], J[14];
      eval_pose_only<true>(P, cam, o.x, o.y, pwl, w_kf[kf], r, J);
      store2_nt(res + i, make_double2(r[0], r[1]));
      double* row = s_tile + threadIdx.x * kTileStride;
#pragma unroll
      for (int k = 0; k < 14; ++k) row[k] = J[k];
    }
    __syncthreads();
    {
      const int cnt = min(kBlock, n - first);
      double2* dst = reinterpret_cast<double2*>(jac + (size_t)first * 14);
      const int pairs = cnt * 7;
      for (int e2 = threadIdx.x; e2 < pairs; e2 += kBlock) {
        const int th = e2 / 7, k = 2 * (e2 - th * 7);
        const double* sp = s_tile + th * kTileStride + k;
        store2_nt(dst + e2, make_double2(sp[0], sp[1]));
      }
    }
    __syncthreads();
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
  stage_poses<kBlock>(s_pose, poses, n_kf);
  const int first = blockIdx.x * kBlock;
  const int i = first + threadIdx.x;
  const int cnt = min(kBlock, n - first);
  double J2row[14];
  if (i < n) {
    const int k1 = kf1_idx[i], k2 = kf2_idx[i];
    const double2 fo = first_ob[i], o = ob[i];
    const PoseD P1 = fetch_pose(s_pose, poses, n_kf, k1);
    const PoseD P2 = fetch_pose(s_pose, poses, n_kf, k2);
    double r[2], Jd[2], J1[14];
    eval_two_frame<WITH_J>(P1, P2, left, right, fo.x, fo.y, o.x, o.y, inv_depth[lm_idx[i]], w_kf[k2], r, Jd, J1, J2row);
    res[i] = make_double2(r[0], r[1]);
    if (WITH_J) {
      jd[i] = make_double2(Jd[0], Jd[1]);
      double* row = s_tile + threadIdx.x * kTileStride;
#pragma unroll
      for (int k = 0; k < 14; ++k) row[k] = J1[k];
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
                                                       const double* __restrict__ w_kf, const double* __restrict__ wblk, const CamD left,
                                                       const CamD right, double2* __restrict__ res,
                                                       double2* __restrict__ jac) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double2 lo = left_ob[i], ro = right_ob[i];
  double r[2], J[2];
  eval_two_camera<WITH_J>(left, right, lo.x, lo.y, ro.x, ro.y, inv_depth[lm_idx[i]], wblk ? wblk[i] : 5.0 * w_kf[kf_idx[i]], r, J);
  res[i] = make_double2(r[0], r[1]);
  if (WITH_J) jac[i] = make_double2(J[0], J[1]);
}

// ------------------------------------------------------------------------------------------ launchers
static inline int grid_for(int n) { return (n + kBlock - 1) / kBlock; }

int launch_pose_only(lvf_batch* b, const lvf_state* st, bool want_j) {
  if (b->n == 0) return LVF_OK;
  hipStream_t s = b->ctx->stream;
  auto ob = reinterpret_cast<const double2*>(b->ob_a.p);
  auto res = reinterpret_cast<double2*>(b->res.p);
  if (want_j && st->n_kf <= kMaxStagedKf) {
    // balanced persistent grid: every workgroup gets the same number of tiles (the last one possibly fewer)
    const int ntiles = grid_for(b->n), cap = b->ctx->num_cu * 4;
    const int per_wg = (ntiles + cap - 1) / cap;
    const int grid = (ntiles + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(k_pose_only_rj, dim3(grid), dim3(kBlock), 0, s, b->n, st->n_kf, ob, b->idx_a.p, b->idx_b.p, b->table.p, st->poses.p,
                       st->w_visual.p, b->cam_a, res, b->jac[0].p);
    LVF_HIP(hipGetLastError());
    return LVF_OK;
  }
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
                       b->idx_b.p, st->inv_depth.p, st->w_visual.p, b->wblk.n ? b->wblk.p : (const double*)nullptr, b->cam_a, b->cam_b, res,
                       reinterpret_cast<double2*>(b->jac[0].p));
  else
    hipLaunchKernelGGL(k_two_camera<false>, dim3(grid_for(b->n)), dim3(kBlock), 0, s, b->n, lo, ro, b->idx_a.p,
                       b->idx_b.p, st->inv_depth.p, st->w_visual.p, b->wblk.n ? b->wblk.p : (const double*)nullptr, b->cam_a, b->cam_b, res, (double2*)nullptr);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

}  // namespace lvf
