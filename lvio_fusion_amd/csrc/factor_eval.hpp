// factor_eval.hpp — per-residual-block device math of the three reprojection factors (residual + ambient Jacobian
// rows), shared by the materialised Ceres-layout kernels (visual_kernels.hip) and the fused linearisation
// (solver_kernels.hip).  References: src/lvio_fusion/include/lvio_fusion/ceres/visual_error.hpp:10-137.
#pragma once
#include "lvf_math.hpp"

namespace lvf {

// PoseOnlyReprojectionError <2,7>: r[2], J[14] row-major 2x7 = [d/dq(4) | d/dt(3)]
template <bool WITH_J>
__device__ __forceinline__ void eval_pose_only(const PoseD& P, const CamD& cam, double obx, double oby, const double pw[3],
                                               double w, double r[2], double J[14]) {
  const double d[3] = {pw[0] - P.t[0], pw[1] - P.t[1], pw[2] - P.t[2]};
  double pb[3];
  mat3t_mul_vec(P.R, d, pb);
  double px[2], M[6];
  project_and_chain(cam, pb, w, px, M);
  r[0] = w * (px[0] - obx); r[1] = w * (px[1] - oby);
  if (WITH_J) {
    const double fmp[3] = {pb[0] - d[0], pb[1] - d[1], pb[2] - d[2]};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double q4[4], t3[3];
      row_times_drot_dq<true>(M + 3 * a, P.u, P.inv_n, d, fmp, q4);
      mat3_mul_vec(P.R, M + 3 * a, t3);  // (M_row R^T)^T = R M_row^T
      J[7 * a + 0] = q4[0]; J[7 * a + 1] = q4[1]; J[7 * a + 2] = q4[2]; J[7 * a + 3] = q4[3];
      J[7 * a + 4] = -t3[0]; J[7 * a + 5] = -t3[1]; J[7 * a + 6] = -t3[2];
    }
  }
}

// TwoFrameReprojectionError <2,1,7,7>: r[2], Jd[2], J1[14], J2[14]
template <bool WITH_J>
__device__ __forceinline__ void eval_two_frame(const PoseD& P1, const PoseD& P2, const CamD& left, const CamD& right,
                                               double fox, double foy, double obx, double oby, double rho, double w,
                                               double r[2], double Jd[2], double J1[14], double J2[14]) {
  // Pixel2Robot through the RIGHT camera (visual_error.hpp:25-33)
  const double dpt = 1.0 / rho;
  const double dir[3] = {(fox - right.cx) / right.fx, (foy - right.cy) / right.fy, 1.0};
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
  r[0] = w * (px[0] - obx); r[1] = w * (px[1] - oby);
  if (WITH_J) {
    const double fmp2[3] = {pb2[0] - dd[0], pb2[1] - dd[1], pb2[2] - dd[2]};
    const double fmp1[3] = {rp[0] - pb1[0], rp[1] - pb1[1], rp[2] - pb1[2]};
    double rd[3];                                                  // R1 Re dir
    { double t[3]; mat3_mul_vec(right.Re, dir, t); mat3_mul_vec(P1.R, t, rd); }
    const double md2 = -(dpt * dpt);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double A[3], q4[4];
      mat3_mul_vec(P2.R, M + 3 * a, A);                            // d r_a / d pw
      row_times_drot_dq<true>(M + 3 * a, P2.u, P2.inv_n, dd, fmp2, q4);
      J2[7 * a + 0] = q4[0]; J2[7 * a + 1] = q4[1]; J2[7 * a + 2] = q4[2]; J2[7 * a + 3] = q4[3];
      J2[7 * a + 4] = -A[0]; J2[7 * a + 5] = -A[1]; J2[7 * a + 6] = -A[2];
      row_times_drot_dq<false>(A, P1.u, P1.inv_n, pb1, fmp1, q4);
      J1[7 * a + 0] = q4[0]; J1[7 * a + 1] = q4[1]; J1[7 * a + 2] = q4[2]; J1[7 * a + 3] = q4[3];
      J1[7 * a + 4] = A[0]; J1[7 * a + 5] = A[1]; J1[7 * a + 6] = A[2];
      Jd[a] = (A[0] * rd[0] + A[1] * rd[1] + A[2] * rd[2]) * md2;
    }
  }
}

// TwoCameraReprojectionError <2,1>: r[2], J[2]; w already includes the 5x (backend.cpp:123)
template <bool WITH_J>
__device__ __forceinline__ void eval_two_camera(const CamD& left, const CamD& right, double lox, double loy, double rox,
                                                double roy, double rho, double w, double r[2], double J[2]) {
  const double dpt = 1.0 / rho;
  const double dir[3] = {(rox - right.cx) / right.fx, (roy - right.cy) / right.fy, 1.0};
  const double ps[3] = {dir[0] * dpt, dir[1] * dpt, dpt};
  double pb[3];
  mat3_mul_vec(right.Re, ps, pb);
  pb[0] += right.te[0]; pb[1] += right.te[1]; pb[2] += right.te[2];
  double px[2], M[6];
  project_and_chain(left, pb, w, px, M);
  r[0] = w * (px[0] - lox); r[1] = w * (px[1] - loy);
  if (WITH_J) {
    double rd[3];
    mat3_mul_vec(right.Re, dir, rd);
    const double md2 = -(dpt * dpt);
    J[0] = (M[0] * rd[0] + M[1] * rd[1] + M[2] * rd[2]) * md2;
    J[1] = (M[3] * rd[0] + M[4] * rd[1] + M[5] * rd[2]) * md2;
  }
}

// ambient 2x7 pose Jacobian -> local 2x6 (EigenQuaternionParameterization x Identity3), optionally scaled
__device__ __forceinline__ void pose_rows_to_local(const double J7[14], const double* __restrict__ q, double scale, double J6[12]) {
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    double l3[3];
    quat_row_to_local(J7 + 7 * a, q, l3);
    J6[6 * a + 0] = scale * l3[0]; J6[6 * a + 1] = scale * l3[1]; J6[6 * a + 2] = scale * l3[2];
    J6[6 * a + 3] = scale * J7[7 * a + 4]; J6[6 * a + 4] = scale * J7[7 * a + 5]; J6[6 * a + 5] = scale * J7[7 * a + 6];
  }
}

// ---- the fused linearisation's forms: pose rows directly in LOCAL (tangent) coordinates, closed form.
// EigenQuaternionParameterization::Plus(x, delta) = q_delta (x) x perturbs the rotation on the LEFT: R+ = (I + 2 [delta]x) R to first order
// (|x| drops out: J_ambient carries 1/|q|, the plus-Jacobian P(q) carries |q|).  For a row a of d residual / d point:
//     forward  f = R p    :  d(a . f)/d delta = 2 (R p) x a              (a in the world frame)
//     inverse  g = R^T d  :  d(a . g)/d delta = 2 (R a) x d              (a in the body frame; R a = d residual / d world point)
// — exactly J_ambient P(q) (row_times_drot_dq followed by quat_row_to_local: ~75 flops per row) as one cross product; checked against
// the ambient chain in tests/test_gpu_factors.py.  The ambient forms above stay what lvf_batch_evaluate / the materialised kernels use.
__device__ __forceinline__ void cross2(const double a[3], const double b[3], double& o0, double& o1, double& o2) {      // o = 2 (a x b)
  o0 = 2.0 * (a[1] * b[2] - a[2] * b[1]); o1 = 2.0 * (a[2] * b[0] - a[0] * b[2]); o2 = 2.0 * (a[0] * b[1] - a[1] * b[0]);
}

// PoseOnlyReprojectionError: r[2] and the 2 x 6 local pose Jacobian scaled by `scale` is applied by the caller through sc (rows are UNSCALED here)
__device__ __forceinline__ void eval_pose_only_local(const PoseD& P, const CamD& cam, double obx, double oby, const double pw[3], double w,
                                                     double r[2], double L[12]) {
  const double d[3] = {pw[0] - P.t[0], pw[1] - P.t[1], pw[2] - P.t[2]};
  double pb[3];
  mat3t_mul_vec(P.R, d, pb);
  double px[2], M[6];
  project_and_chain(cam, pb, w, px, M);
  r[0] = w * (px[0] - obx); r[1] = w * (px[1] - oby);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    double A[3];
    mat3_mul_vec(P.R, M + 3 * a, A);                               // d r_a / d pw
    cross2(A, d, L[6 * a + 0], L[6 * a + 1], L[6 * a + 2]);
    L[6 * a + 3] = -A[0]; L[6 * a + 4] = -A[1]; L[6 * a + 5] = -A[2];
  }
}

// TwoFrameReprojectionError: r[2], Jd[2] and the two 2 x 6 local pose Jacobians (first keyframe L1, current keyframe L2), unscaled
__device__ __forceinline__ void eval_two_frame_local(const PoseD& P1, const PoseD& P2, const CamD& left, const CamD& right,
                                                     double fox, double foy, double obx, double oby, double rho, double w,
                                                     double r[2], double Jd[2], double L1[12], double L2[12]) {
  const double dpt = 1.0 / rho;
  const double dir[3] = {(fox - right.cx) / right.fx, (foy - right.cy) / right.fy, 1.0};
  double rd0[3];                                                   // Re dir: pb1 = dpt * rd0 + te
  mat3_mul_vec(right.Re, dir, rd0);
  const double pb1[3] = {rd0[0] * dpt + right.te[0], rd0[1] * dpt + right.te[1], rd0[2] * dpt + right.te[2]};
  double rp[3], rd[3];
  mat3_mul_vec(P1.R, pb1, rp);                                     // R1 pb1
  mat3_mul_vec(P1.R, rd0, rd);                                     // R1 Re dir
  const double dd[3] = {rp[0] + P1.t[0] - P2.t[0], rp[1] + P1.t[1] - P2.t[1], rp[2] + P1.t[2] - P2.t[2]};
  double pb2[3];
  mat3t_mul_vec(P2.R, dd, pb2);
  double px[2], M[6];
  project_and_chain(left, pb2, w, px, M);
  r[0] = w * (px[0] - obx); r[1] = w * (px[1] - oby);
  const double md2 = -(dpt * dpt);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    double A[3];
    mat3_mul_vec(P2.R, M + 3 * a, A);                              // d r_a / d pw
    cross2(A, dd, L2[6 * a + 0], L2[6 * a + 1], L2[6 * a + 2]);
    L2[6 * a + 3] = -A[0]; L2[6 * a + 4] = -A[1]; L2[6 * a + 5] = -A[2];
    cross2(rp, A, L1[6 * a + 0], L1[6 * a + 1], L1[6 * a + 2]);
    L1[6 * a + 3] = A[0]; L1[6 * a + 4] = A[1]; L1[6 * a + 5] = A[2];
    Jd[a] = (A[0] * rd[0] + A[1] * rd[1] + A[2] * rd[2]) * md2;
  }
}

constexpr int kMaxStagedKf = 64;
// every workgroup derives the window's pose blocks once into LDS ("LDS-staged SE3 pose blocks")
template <int BLOCK>
__device__ __forceinline__ void stage_poses(PoseD* s_pose, const double* __restrict__ poses, int n_kf) {
  if (n_kf <= kMaxStagedKf) {
    for (int k = threadIdx.x; k < n_kf; k += BLOCK) derive_pose(poses + 7 * k, s_pose[k]);
    __syncthreads();
  }
}
__device__ __forceinline__ PoseD fetch_pose(const PoseD* s_pose, const double* __restrict__ poses, int n_kf, int kf) {
  if (n_kf <= kMaxStagedKf) return s_pose[kf];
  PoseD d;
  derive_pose(poses + 7 * kf, d);
  return d;
}

}  // namespace lvf
