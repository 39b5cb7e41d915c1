// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  PARITY UNPINNED.
//
// oracle_capi.cpp — extern "C" batched entry points over the restated reference math so
// that tests/ and bench.py's cpu_baseline leg can drive it through ctypes.  `threads`
// mirrors the reference's Ceres num_threads (src/lvio_fusion/src/estimator.cpp:10):
// OpenMP over residual blocks / query points.
#include <omp.h>
#include <cstdint>
#include <cstring>
#include <vector>
#include "factors.h"
#include "imu.h"
#include "icp.h"
#include "cloud.h"
#include "extract.h"
#include "knn.h"
#include "lm.h"
#include "loop.h"
#include "robust.h"

using namespace lvo;

extern "C" {

struct lvo_camera { double fx, fy, cx, cy; double extrinsic[7]; };
static inline Camera cam_of(const lvo_camera* c) {
  Camera k; k.fx = c->fx; k.fy = c->fy; k.cx = c->cx; k.cy = c->cy; std::memcpy(k.extrinsic, c->extrinsic, sizeof(k.extrinsic)); return k;
}

// flattened imu::Preint (same field order as include/lvf.h's lvf_preint)
struct lvo_preint {
  double sum_dt; double lin_ba[3]; double lin_bg[3]; double dp[3]; double dq[4]; double dv[3];
  double jac[225]; double cov[225];
};

int lvo_max_threads(void) { return omp_get_max_threads(); }

// ---------------- visual ----------------
// PoseOnly: r[n][2], J[n][2][7] (row-major 2x7), J may be null.
void lvo_pose_only_eval(int n, const double* ob, const int* kf_idx, const int* pw_idx, const double* pw,
                        const double* poses, const double* w_kf, const lvo_camera* cam0_, double* r,
                        double* J, int threads) {
  const Camera cam0 = cam_of(cam0_);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < n; ++i) {
    const double* pose = poses + 7 * kf_idx[i];
    const double w = w_kf[kf_idx[i]];
    if (J) {
      Jet<7> T[7], rr[2];
      for (int k = 0; k < 7; ++k) T[k] = Jet<7>(pose[k], k);
      PoseOnlyResidual(ob + 2 * i, pw + 3 * pw_idx[i], cam0, w, T, rr);
      for (int a = 0; a < 2; ++a) { r[2 * i + a] = rr[a].a; for (int k = 0; k < 7; ++k) J[14 * i + 7 * a + k] = rr[a].v[k]; }
    } else {
      PoseOnlyResidual<double>(ob + 2 * i, pw + 3 * pw_idx[i], cam0, w, pose, r + 2 * i);
    }
  }
}

// TwoFrame: r[n][2], J_d[n][2], J_p1[n][14], J_p2[n][14]; weight = w_kf[kf2]
void lvo_two_frame_eval(int n, const double* first_ob, const double* ob, const int* lm_idx, const int* kf1_idx,
                        const int* kf2_idx, const double* inv_depth, const double* poses, const double* w_kf,
                        const lvo_camera* left_, const lvo_camera* right_, double* r, double* Jd, double* J1,
                        double* J2, int threads) {
  const Camera left = cam_of(left_), right = cam_of(right_);
  const bool want_j = Jd || J1 || J2;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < n; ++i) {
    const double* p1 = poses + 7 * kf1_idx[i];
    const double* p2 = poses + 7 * kf2_idx[i];
    const double w = w_kf[kf2_idx[i]];
    const double rho = inv_depth[lm_idx[i]];
    if (want_j) {
      Jet<15> d(rho, 0), A[7], B[7], rr[2];
      for (int k = 0; k < 7; ++k) { A[k] = Jet<15>(p1[k], 1 + k); B[k] = Jet<15>(p2[k], 8 + k); }
      TwoFrameResidual(first_ob + 2 * i, ob + 2 * i, left, right, w, &d, A, B, rr);
      for (int a = 0; a < 2; ++a) {
        r[2 * i + a] = rr[a].a;
        if (Jd) Jd[2 * i + a] = rr[a].v[0];
        if (J1) for (int k = 0; k < 7; ++k) J1[14 * i + 7 * a + k] = rr[a].v[1 + k];
        if (J2) for (int k = 0; k < 7; ++k) J2[14 * i + 7 * a + k] = rr[a].v[8 + k];
      }
    } else {
      TwoFrameResidual<double>(first_ob + 2 * i, ob + 2 * i, left, right, w, &rho, p1, p2, r + 2 * i);
    }
  }
}

// TwoCamera: r[n][2], J[n][2]; weight = 5 * w_kf[kf]  (backend.cpp:123)
void lvo_two_camera_eval(int n, const double* left_ob, const double* right_ob, const int* lm_idx, const int* kf_idx,
                         const double* inv_depth, const double* w_kf, const lvo_camera* left_,
                         const lvo_camera* right_, double* r, double* J, int threads) {
  const Camera left = cam_of(left_), right = cam_of(right_);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < n; ++i) {
    const double w = 5 * w_kf[kf_idx[i]];
    const double rho = inv_depth[lm_idx[i]];
    if (J) {
      Jet<1> d(rho, 0), rr[2];
      TwoCameraResidual(left_ob + 2 * i, right_ob + 2 * i, left, right, w, &d, rr);
      for (int a = 0; a < 2; ++a) { r[2 * i + a] = rr[a].a; J[2 * i + a] = rr[a].v[0]; }
    } else {
      TwoCameraResidual<double>(left_ob + 2 * i, right_ob + 2 * i, left, right, w, &rho, r + 2 * i);
    }
  }
}

// ---------------- lidar ----------------
void lvo_plane_normals(int n, const double* pa, const double* pb, const double* pc, double* nrm) {
  for (int i = 0; i < n; ++i) PlaneNormal(pa + 3 * i, pb + 3 * i, pc + 3 * i, nrm + 3 * i);
}
// mode 0 = RPZ (params pitch=rpyxyz[1], roll=[2], z=[5]); mode 1 = YXY (yaw=[0], x=[3], y=[4]).
// The three scalar parameter blocks are read from rpyxyz itself (the reference passes
// para+1, para+2, para+5 / para+0, para+3, para+4 — association.cpp:274-276,332-334).
// r[n], J[n][3] (d r / d param0..2).
void lvo_lidar_plane_eval(int mode, int n, const double* p, const double* pa, const double* nrm,
                          const double* Twc1, const double* rpyxyz, double weight, double* r, double* J,
                          int threads) {
  const int i0 = mode == 0 ? 1 : 0, i1 = mode == 0 ? 2 : 3, i2 = mode == 0 ? 5 : 4;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < n; ++i) {
    if (J) {
      Jet<3> a(rpyxyz[i0], 0), b(rpyxyz[i1], 1), c(rpyxyz[i2], 2), rr;
      if (mode == 0) LidarPlaneRpzResidual(p + 3 * i, pa + 3 * i, nrm + 3 * i, Twc1, rpyxyz, weight, &a, &b, &c, &rr);
      else LidarPlaneYxyResidual(p + 3 * i, pa + 3 * i, nrm + 3 * i, Twc1, rpyxyz, weight, &a, &b, &c, &rr);
      r[i] = rr.a; J[3 * i] = rr.v[0]; J[3 * i + 1] = rr.v[1]; J[3 * i + 2] = rr.v[2];
    } else {
      const double a = rpyxyz[i0], b = rpyxyz[i1], c = rpyxyz[i2];
      if (mode == 0) LidarPlaneRpzResidual<double>(p + 3 * i, pa + 3 * i, nrm + 3 * i, Twc1, rpyxyz, weight, &a, &b, &c, r + i);
      else LidarPlaneYxyResidual<double>(p + 3 * i, pa + 3 * i, nrm + 3 * i, Twc1, rpyxyz, weight, &a, &b, &c, r + i);
    }
  }
}
// inner factor LidarPlaneError<1,7>: r[n], J[n][7]
void lvo_lidar_plane_se3_eval(int n, const double* p, const double* pa, const double* nrm, const double* Twc2,
                              double* r, double* J) {
  for (int i = 0; i < n; ++i) {
    Jet<7> T[7], rr;
    for (int k = 0; k < 7; ++k) T[k] = Jet<7>(Twc2[k], k);
    LidarPlaneResidual(p + 3 * i, pa + 3 * i, nrm + 3 * i, T, &rr);
    r[i] = rr.a;
    if (J) for (int k = 0; k < 7; ++k) J[7 * i + k] = rr.v[k];
  }
}

// ---------------- pose priors (a11) ----------------
void lvo_pose_graph_eval(const double* target_rpyxyz, double weight, double v, const double* Twc1,
                         const double* Twc2, double* r, double* J1, double* J2) {
  Jet<14> A[7], B[7], rr[6];
  for (int k = 0; k < 7; ++k) { A[k] = Jet<14>(Twc1[k], k); B[k] = Jet<14>(Twc2[k], 7 + k); }
  PoseGraphResidual(target_rpyxyz, weight, v, A, B, rr);
  for (int a = 0; a < 6; ++a) {
    r[a] = rr[a].a;
    for (int k = 0; k < 7; ++k) { if (J1) J1[7 * a + k] = rr[a].v[k]; if (J2) J2[7 * a + k] = rr[a].v[7 + k]; }
  }
}
void lvo_pose_graph_target(const double* last_pose, const double* pose, double* target_rpyxyz) {
  double inv[7], rel[7];   // pose_error.hpp:13-17  (Sophus inverse*product on unit quaternions)
  Se3Inv<double>(last_pose, inv); Se3Mul<double>(inv, pose, rel); Se3ToRpyxyz<double>(rel, target_rpyxyz);
}
void lvo_pose_prior_eval(const double* origin, double weight, double v, const double* pose, double* r, double* J) {
  Jet<7> T[7], rr[6];
  for (int k = 0; k < 7; ++k) T[k] = Jet<7>(pose[k], k);
  PosePriorResidual(origin, weight, v, T, rr);
  for (int a = 0; a < 6; ++a) { r[a] = rr[a].a; if (J) for (int k = 0; k < 7; ++k) J[7 * a + k] = rr[a].v[k]; }
}
void lvo_r_error_eval(const double* origin, double weight, const double* pose, double* r, double* J) {
  Jet<7> T[7], rr[4];
  for (int k = 0; k < 7; ++k) T[k] = Jet<7>(pose[k], k);
  RErrorResidual(origin, weight, T, rr);
  for (int a = 0; a < 4; ++a) { r[a] = rr[a].a; if (J) for (int k = 0; k < 7; ++k) J[7 * a + k] = rr[a].v[k]; }
}
void lvo_t_error_eval(const double* p3, double weight, const double* pose, double* r, double* J) {
  Jet<7> T[7], rr[3];
  for (int k = 0; k < 7; ++k) T[k] = Jet<7>(pose[k], k);
  TErrorResidual(p3, weight, T, rr);
  for (int a = 0; a < 3; ++a) { r[a] = rr[a].a; if (J) for (int k = 0; k < 7; ++k) J[7 * a + k] = rr[a].v[k]; }
}
// RelocateRError <7,4>: r[7], J[7][4]
void lvo_relocate_r_eval(const double* relocated, const double* unrelocated, const double* q4, double* r, double* J) {
  Jet<4> Q[4], rr[7];
  for (int k = 0; k < 4; ++k) Q[k] = Jet<4>(q4[k], k);
  RelocateRResidual(relocated, unrelocated, Q, rr);
  for (int a = 0; a < 7; ++a) { r[a] = rr[a].a; if (J) for (int k = 0; k < 4; ++k) J[4 * a + k] = rr[a].v[k]; }
}
// Relocator::UpdateNewSubmap's rotation solve (relocator.cpp:247-268); opts6 = (max_iters, function_tol, gradient_tol, parameter_tol,
// min_relative_decrease, initial radius); out5 = (initial_cost, final_cost, iterations, successful steps, termination)
void lvo_relocate_rotation_solve(int n, const double* relocated, const double* unrelocated, double* q4, const double* opts6, double* out5) {
  RelocOut o;
  relocate_rotation_solve(n, relocated, unrelocated, q4, (int)opts6[0], opts6[1], opts6[2], opts6[3], opts6[4], opts6[5], &o);
  out5[0] = o.initial_cost; out5[1] = o.final_cost; out5[2] = o.iters; out5[3] = o.successes; out5[4] = o.termination;
}
void lvo_cr_atan2f(int n, const float* y, const float* x, float* out) { for (int i = 0; i < n; ++i) out[i] = cr_atan2f(y[i], x[i]); }
void lvo_forward_update(const double* T, int n, double* poses, double* vw) { forward_update(T, n, poses, vw); }
void lvo_prior3_eval(int mode, const double* rpyxyz0, double weight, const double* rpyxyz, double* r, double* J) {
  if (mode == 0) PriorRpzResidual<double>(rpyxyz0, weight, rpyxyz + 1, rpyxyz + 2, rpyxyz + 5, r);
  else PriorYxyResidual<double>(rpyxyz0, weight, rpyxyz + 0, rpyxyz + 3, rpyxyz + 4, r);
  if (J) {  // 3x3, d r / d (param0,param1,param2); RPZ residual order is (roll, pitch, z) vs params (pitch, roll, z)
    std::memset(J, 0, 9 * sizeof(double));
    if (mode == 0) { J[0 * 3 + 1] = weight; J[1 * 3 + 0] = weight; J[2 * 3 + 2] = weight; }
    else { J[0] = weight; J[4] = weight; J[8] = weight; }
  }
}

// ---------------- se3 helpers for tests ----------------
void lvo_se3_to_rpyxyz(const double* se3, double* rpyxyz) { Se3ToRpyxyz<double>(se3, rpyxyz); }
void lvo_rpyxyz_to_se3(const double* rpyxyz, double* se3) { RpyxyzToSe3<double>(rpyxyz, se3); }
void lvo_se3_mul(const double* A, const double* B, double* C) { Se3Mul<double>(A, B, C); }
void lvo_se3_inv(const double* A, double* C) { Se3Inv<double>(A, C); }
void lvo_se3_apply(const double* A, const double* p, double* o) { Se3Apply<double>(A, p, o); }
void lvo_se3_apply_f32(const float* A, const float* p, float* o) { Se3Apply<float>(A, p, o); }

// ---------------- robust / parameterisation ----------------
void lvo_loss(double a, double s, double* rho) { loss_eval(a, s, rho); }
void lvo_quat_plus(const double* x, const double* d, double* out) { eigen_quat_plus(x, d, out); }
void lvo_quat_plus_jacobian(const double* x, double* j12) { eigen_quat_plus_jacobian(x, j12); }
void lvo_pose_jac_to_local(const double* pose, int rows, const double* J7, double* J6) { pose_jac_to_local(pose, rows, J7, J6); }

// ---------------- IMU ----------------
static void preint_to_flat(const imu::Preint& P, lvo_preint* o) {
  o->sum_dt = P.sum_dt;
  for (int i = 0; i < 3; ++i) { o->lin_ba[i] = P.lin_ba[i]; o->lin_bg[i] = P.lin_bg[i]; o->dp[i] = P.dp[i]; o->dv[i] = P.dv[i]; }
  o->dq[0] = P.dq.x; o->dq[1] = P.dq.y; o->dq[2] = P.dq.z; o->dq[3] = P.dq.w;
  std::memcpy(o->jac, P.jac, sizeof(o->jac)); std::memcpy(o->cov, P.cov, sizeof(o->cov));
}
static void flat_to_preint(const lvo_preint* f, imu::Preint& P) {
  std::memset(&P, 0, sizeof(P));
  P.sum_dt = f->sum_dt;
  for (int i = 0; i < 3; ++i) { P.lin_ba[i] = f->lin_ba[i]; P.lin_bg[i] = f->lin_bg[i]; P.dp[i] = f->dp[i]; P.dv[i] = f->dv[i]; }
  P.dq = imu::Quat{f->dq[0], f->dq[1], f->dq[2], f->dq[3]};
  std::memcpy(P.jac, f->jac, sizeof(P.jac)); std::memcpy(P.cov, f->cov, sizeof(P.cov));
}
// samples[ns][7] = (dt, acc xyz, gyr xyz); acc0/gyr0 = the measurement latched at the first Append
void lvo_imu_preintegrate(int ns, const double* samples, const double* acc0, const double* gyr0, const double* ba,
                          const double* bg, const double* noise4, lvo_preint* out) {
  imu::Preint P;
  P.init(ba, bg, acc0, gyr0, imu::Noise{noise4[0], noise4[1], noise4[2], noise4[3]});
  for (int s = 0; s < ns; ++s) P.propagate(samples[7 * s], samples + 7 * s + 1, samples + 7 * s + 4);
  preint_to_flat(P, out);
}
void lvo_imu_sqrt_info(const lvo_preint* f, double* S225) {
  imu::Preint P; flat_to_preint(f, P);
  double S[15][15]; imu::sqrt_info_from_cov(P.cov, S); std::memcpy(S225, S, sizeof(S));
}
// n factors; state arrays poses[nkf][7], vel/ba/bg[nkf][3]; r[n][15]; J packed per factor as
// 15x7,15x3,15x3,15x3,15x7,15x3,15x3,15x3 row-major blocks = 15*32 doubles (J may be null).
void lvo_imu_eval(int n, const lvo_preint* pre, const int* kf_i, const int* kf_j, const double* poses,
                  const double* vel, const double* ba, const double* bg, double* r, double* J, int threads) {
  static const int off[8] = {0, 105, 150, 195, 240, 345, 390, 435};
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int f = 0; f < n; ++f) {
    imu::Preint P; flat_to_preint(pre + f, P);
    const int i = kf_i[f], j = kf_j[f];
    const double* prm[8] = {poses + 7 * i, vel + 3 * i, ba + 3 * i, bg + 3 * i, poses + 7 * j, vel + 3 * j, ba + 3 * j, bg + 3 * j};
    double* Jp[8];
    if (J) for (int k = 0; k < 8; ++k) Jp[k] = J + (size_t)480 * f + off[k];
    imu::imu_error_evaluate(P, prm, r + 15 * f, J ? Jp : nullptr);
  }
}

// ---------------- kNN association ----------------
// query/map clouds: float xyz with `stride` floats per point.  tf_d = frame pose (double, Sophus
// order), cast to float exactly as association.cpp:287.  Outputs idx[Q][3], d2[Q][3], valid[Q].
// method 0 = brute force (checker), 1 = leaf-15 kd-tree (timing baseline).
void lvo_knn3(const float* map, int M, int mstride, const float* query, int Q, int qstride, const double* tf_d,
              float thr, int* idx, float* d2, uint8_t* valid, int method, int threads) {
  float tf[7];
  for (int k = 0; k < 7; ++k) tf[k] = (float)tf_d[k];
  KdTree tree;
  if (method == 1) tree.build(map, M, mstride);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256)
  for (int i = 0; i < Q; ++i) {
    float w[3];
    transform_query_f32(tf, query + (size_t)i * qstride, w);
    Best3 b;
    if (method == 1) b = tree.query(w); else knn3_brute(map, M, mstride, w, &b);
    for (int k = 0; k < 3; ++k) { idx[3 * i + k] = b.i[k]; d2[3 * i + k] = b.d[k]; }
    valid[i] = (b.i[0] >= 0 && b.d[0] < thr && b.i[1] >= 0 && b.d[1] < thr && b.i[2] >= 0 && b.d[2] < thr) ? 1 : 0;
  }
}
double lvo_kdtree_build_seconds(const float* map, int M, int mstride) {
  const double t0 = omp_get_wtime();
  KdTree tree; tree.build(map, M, mstride);
  return omp_get_wtime() - t0 + (tree.nodes.empty() ? 1e-12 : 0.0);
}

// ---------------- sliding-window problem / LM iteration ----------------
struct lvo_window_c {
  int n_kf, n_lm;
  double *poses, *vel, *ba, *bg, *inv_depth;
  const double* w_kf;
  lvo_camera cam0, cam1;
  int n_tc; const double *tc_left_ob, *tc_right_ob; const int *tc_lm, *tc_kf;
  int n_tf; const double *tf_first_ob, *tf_ob; const int *tf_lm, *tf_kf1, *tf_kf2;
  int n_po; const double* po_ob; const int *po_kf, *po_pw; const double* po_pwtab;
  int n_imu; const lvo_preint* pre; const int *imu_i, *imu_j;
  const unsigned char* pose_const;
  int n_prior; const int *prior_a, *prior_b; const double *prior_target, *prior_w, *prior_v;
  const unsigned char* vbb_const;      // may be null: per keyframe bit 0 / 1 / 2 = constant velocity / ba / bg block
};
static void to_window(const lvo_window_c* c, Window& w, std::vector<imu::Preint>& pre) {
  w.n_kf = c->n_kf; w.n_lm = c->n_lm; w.poses = c->poses; w.vel = c->vel; w.ba = c->ba; w.bg = c->bg; w.inv_depth = c->inv_depth;
  w.w_kf = c->w_kf; w.cam0 = cam_of(&c->cam0); w.cam1 = cam_of(&c->cam1);
  w.n_tc = c->n_tc; w.tc_left_ob = c->tc_left_ob; w.tc_right_ob = c->tc_right_ob; w.tc_lm = c->tc_lm; w.tc_kf = c->tc_kf;
  w.n_tf = c->n_tf; w.tf_first_ob = c->tf_first_ob; w.tf_ob = c->tf_ob; w.tf_lm = c->tf_lm; w.tf_kf1 = c->tf_kf1; w.tf_kf2 = c->tf_kf2;
  w.n_po = c->n_po; w.po_ob = c->po_ob; w.po_kf = c->po_kf; w.po_pw = c->po_pw; w.po_pwtab = c->po_pwtab;
  pre.resize(c->n_imu);
  for (int f = 0; f < c->n_imu; ++f) flat_to_preint(c->pre + f, pre[f]);
  w.n_imu = c->n_imu; w.pre = pre.data(); w.imu_i = c->imu_i; w.imu_j = c->imu_j;
  w.pose_const = c->pose_const;
  w.n_prior = c->n_prior; w.prior_a = c->prior_a; w.prior_b = c->prior_b; w.prior_target = c->prior_target; w.prior_w = c->prior_w; w.prior_v = c->prior_v;
  w.vbb_const = c->vbb_const;
}
double lvo_window_cost(const lvo_window_c* c, double huber_a) {
  Window w; std::vector<imu::Preint> pre; to_window(c, w, pre);
  return window_cost(w, huber_a, w.poses, w.vel, w.ba, w.bg, w.inv_depth);
}
// B[d*d], gc[d], E[n_lm*6n_kf], C[n_lm], gr[n_lm]; any output may be null
double lvo_window_linearize(const lvo_window_c* c, double huber_a, double* B, double* gc, double* E, double* C, double* gr) {
  Window w; std::vector<imu::Preint> pre; to_window(c, w, pre);
  Linearization L; window_linearize(w, huber_a, L);
  if (B) std::memcpy(B, L.B.data(), L.B.size() * 8);
  if (gc) std::memcpy(gc, L.gc.data(), L.gc.size() * 8);
  if (E) std::memcpy(E, L.E.data(), L.E.size() * 8);
  if (C) std::memcpy(C, L.C.data(), L.C.size() * 8);
  if (gr) std::memcpy(gr, L.gr.data(), L.gr.size() * 8);
  return L.cost;
}
// out6 = {cost_before, cost_after, model_cost_change, rho, accepted, solved}; S[d*d], rhs[d] optional taps.
// The state arrays inside c are updated in place when the step is accepted; radius/decrease_factor are updated.
// h0 (may be null: a one-iteration solve, H0 = this linearisation's diagonal): the solve's Jacobi scaling state, 15 n_kf + n_lm doubles of
// diag(J^T J) at iteration 0; *frozen == 0 -> taken from this linearisation and written back with *frozen = 1 (lm.h JacobiScale).
void lvo_window_lm_iteration_js(lvo_window_c* c, double huber_a, double min_relative_decrease, double* radius,
                                double* decrease_factor, double* out6, double* S, double* rhs, double* h0, int* frozen) {
  Window w; std::vector<imu::Preint> pre; to_window(c, w, pre);
  LmStep st;
  JacobiScale js;
  const int d = 15 * w.n_kf;
  if (h0 && frozen && *frozen) { js.c.assign(h0, h0 + d); js.l.assign(h0 + d, h0 + d + w.n_lm); js.frozen = true; }
  lm_iteration(w, huber_a, min_relative_decrease, radius, decrease_factor, st, h0 ? &js : nullptr);
  if (h0 && frozen && !*frozen && js.frozen) { std::memcpy(h0, js.c.data(), 8 * (size_t)d); std::memcpy(h0 + d, js.l.data(), 8 * (size_t)w.n_lm); *frozen = 1; }
  out6[0] = st.cost_before; out6[1] = st.cost_after; out6[2] = st.model_cost_change; out6[3] = st.rho;
  out6[4] = st.accepted ? 1.0 : 0.0; out6[5] = st.solved ? 1.0 : 0.0;
  if (S) std::memcpy(S, st.S.data(), st.S.size() * 8);
  if (rhs) std::memcpy(rhs, st.rhs.data(), st.rhs.size() * 8);
}

void lvo_window_lm_iteration(lvo_window_c* c, double huber_a, double min_relative_decrease, double* radius,
                             double* decrease_factor, double* out6, double* S, double* rhs) {
  lvo_window_lm_iteration_js(c, huber_a, min_relative_decrease, radius, decrease_factor, out6, S, rhs, nullptr, nullptr);
}

// The whole solve (lm.h lm_solve: Ceres' TrustRegionMinimizer loop restated).  opts7 = {max_num_iterations, huber_a, initial radius,
// function_tolerance, gradient_tolerance, parameter_tolerance, min_relative_decrease}; out9 = {initial_cost, final_cost, num_iterations,
// num_successful_steps, num_unsuccessful_steps, termination, why, final_radius, final_decrease_factor, num_trials}; trace (may be null): 6 doubles
// per trial step, max_num_iterations + 1 rows.  The state arrays inside c are updated in place.
// opts7[7] (an EIGHTH value) != 0: the pre-round-5 damping (clamp on the unscaled diagonal) — test-only, see lm.h JacobiScale
void lvo_window_solve(lvo_window_c* c, const double* opts7, double* out9 /* 10 values */, double* trace) {
  Window w; std::vector<imu::Preint> pre; to_window(c, w, pre);
  SolveOptions o{(int)opts7[0], opts7[1], opts7[2], opts7[3], opts7[4], opts7[5], opts7[6]};
  o.unscaled_clamp = opts7[7] != 0.0;
  SolveSummary s;
  lm_solve(w, o, s, trace);
  out9[0] = s.initial_cost; out9[1] = s.final_cost; out9[2] = s.num_iterations; out9[3] = s.num_successful_steps; out9[4] = s.num_unsuccessful_steps;
  out9[5] = s.termination; out9[6] = s.why; out9[7] = s.final_radius; out9[8] = s.final_decrease_factor; out9[9] = s.num_trials;
}

// ---------------- scan-to-map sub-problem ----------------
// out5 = {initial_cost, final_cost, num_residual_blocks, iterations, successful steps}; rpyxyz updated in place
void lvo_icp_solve(const float* map, int M, int mstride, const float* query, int Q, int qstride, const double* map_pose,
                   const double* frame_pose, double* rpyxyz, int mode, float thr, double weight, double huber_a, double prior_w,
                   int max_iters, int use_kdtree, double* out5) {
  IcpOut o;
  icp_solve(map, M, mstride, query, Q, qstride, map_pose, frame_pose, rpyxyz, mode, thr, weight, huber_a, prior_w, max_iters, use_kdtree != 0, &o);
  out5[0] = o.initial_cost; out5[1] = o.final_cost; out5[2] = o.nres; out5[3] = o.iters; out5[4] = o.successes;
}


// ---------------- map-cloud maintenance (cloud.h) ----------------
int lvo_align_scan_range(int n1, double stamp1, int n2, double stamp2, double cycle_time, double time, long long* first_last2) {
  return align_scan_range(n1, stamp1, n2, stamp2, cycle_time, time, first_last2, first_last2 + 1) ? 1 : 0;
}
void lvo_cloud_transform(const float* in, int n, const double* pose, float* out) { cloud_transform(in, n, pose, out); }
// returns the number of output points; out (capacity n*4 floats) receives them
int lvo_voxel_filter(const float* in, int n, float leaf, float* out) {
  const std::vector<float> v = voxel_filter(in, n, leaf);
  std::memcpy(out, v.data(), v.size() * sizeof(float));
  return (int)(v.size() / 4);
}
void lvo_radius_outlier_keep(const float* in, int n, float radius, int min_neighbors, unsigned char* keep) {
  const auto k = radius_outlier_keep(in, n, radius, min_neighbors);
  std::memcpy(keep, k.data(), k.size());
}
int lvo_segment_plane(const float* in, int n, float thr, int max_iterations, unsigned long long seed, unsigned char* mask, double* coeff4) {
  int iters = 0;
  const auto m = segment_plane(in, n, thr, max_iterations, seed, coeff4, &iters);
  std::memcpy(mask, m.data(), m.size());
  return iters;
}


// ---------------- LiDAR feature extraction (extract.h) ----------------
struct lvo_lidar_params_c { int num_scans, horizon_scan, ground_rows; float ang_res_y, ang_bottom, min_range, max_range, resolution; double cycle_time; };
// outputs (caller-allocated, capacity n points each): ground / surf final clouds; ground_raw / surf_raw picks; label/ground/range mats
// counts8 = {n_filtered, n_segmented, n_ground_raw, n_surf_raw, n_ground, n_surf, 0, 0}
void lvo_lidar_extract(const float* pts, int n, int stride, const lvo_lidar_params_c* p, const double* extrinsic, unsigned long long seed, float* ground,
                       float* surf, float* ground_raw, float* surf_raw, int* label_mat, signed char* ground_mat, float* range_mat, int* counts8) {
  LidarParams P{p->num_scans, p->horizon_scan, p->ground_rows, p->ang_res_y, p->ang_bottom, p->min_range, p->max_range, p->resolution, p->cycle_time};
  ExtractDebug D;
  lidar_extract<false>(pts, n, stride, P, D);
  std::vector<float> g, s;
  lidar_extract_tail(D, P, extrinsic, seed, g, s);
  std::memcpy(ground, g.data(), g.size() * 4); std::memcpy(surf, s.data(), s.size() * 4);
  std::memcpy(ground_raw, D.ground_raw.data(), D.ground_raw.size() * 4); std::memcpy(surf_raw, D.surf_raw.data(), D.surf_raw.size() * 4);
  std::memcpy(label_mat, D.label_mat.data(), D.label_mat.size() * 4); std::memcpy(ground_mat, D.ground_mat.data(), D.ground_mat.size());
  std::memcpy(range_mat, D.range_mat.data(), D.range_mat.size() * 4);
  counts8[0] = (int)D.filtered.size() / 4; counts8[1] = (int)D.segmented.size() / 4; counts8[2] = (int)D.ground_raw.size() / 4; counts8[3] = (int)D.surf_raw.size() / 4;
  counts8[4] = (int)g.size() / 4; counts8[5] = (int)s.size() / 4; counts8[6] = counts8[7] = 0;
}

// the restatement's taps, with libm's atan2f (libm != 0: the form that is pinned bit for bit to the reference's text) or cr_atan2f (what the GPU is
// compared with).  Arrays sized like oracle/_ref's lvr_lidar_extract: clouds [cap][4], mats [rows * cols].  counts6 = {filtered, segmented, ground picks, surf picks, 0, 0}
void lvo_lidar_extract_taps(const float* pts, int n, int stride, const lvo_lidar_params_c* p, int libm, float* filtered, float* range_mat, signed char* ground_mat,
                            int* label_mat, float* segmented, unsigned char* seg_ground, int* seg_col, float* seg_range, int* start_ring, int* end_ring, float* curvature,
                            float* ground_raw, float* surf_raw, int* counts6) {
  LidarParams P{p->num_scans, p->horizon_scan, p->ground_rows, p->ang_res_y, p->ang_bottom, p->min_range, p->max_range, p->resolution, p->cycle_time};
  ExtractDebug D;
  if (libm) lidar_extract<true>(pts, n, stride, P, D); else lidar_extract<false>(pts, n, stride, P, D);
  std::memcpy(filtered, D.filtered.data(), D.filtered.size() * 4);
  std::memcpy(range_mat, D.range_mat.data(), D.range_mat.size() * 4); std::memcpy(ground_mat, D.ground_mat.data(), D.ground_mat.size()); std::memcpy(label_mat, D.label_mat.data(), D.label_mat.size() * 4);
  std::memcpy(segmented, D.segmented.data(), D.segmented.size() * 4);
  const size_t m = D.segmented.size() / 4;
  for (size_t k = 0; k < m; ++k) { seg_ground[k] = D.seg_ground[k]; seg_col[k] = D.seg_col[k]; seg_range[k] = D.seg_range[k]; curvature[k] = D.curvature[k]; }
  for (int i = 0; i < p->num_scans; ++i) { start_ring[i] = D.start_ring[i]; end_ring[i] = D.end_ring[i]; }
  std::memcpy(ground_raw, D.ground_raw.data(), D.ground_raw.size() * 4); std::memcpy(surf_raw, D.surf_raw.data(), D.surf_raw.size() * 4);
  counts6[0] = (int)D.filtered.size() / 4; counts6[1] = (int)m; counts6[2] = (int)D.ground_raw.size() / 4; counts6[3] = (int)D.surf_raw.size() / 4; counts6[4] = counts6[5] = 0;
}

}  // extern "C"
