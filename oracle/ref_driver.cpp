// ORACLE — TEST INFRASTRUCTURE ONLY.  oracle/_ref/liblvf_ref.so: the REFERENCE's own cost functors, compiled unmodified from
//   /root/reference/src/lvio_fusion/include/lvio_fusion/ceres/{base,visual_error,lidar_error,pose_error}.hpp
//   (+ visual/camera.h, sensor.h, common.h they include)
// against the stand-in third-party headers under oracle/ref_shim/ (Ceres Jet/AutoDiffCostFunction/rotation, Eigen vectors, Sophus
// SE3d, OpenCV Mat, PCL point names).  No reference source is copied into this repository: the headers are included from where
// they lie (-I/root/reference/src/lvio_fusion/include, oracle/Makefile target `ref`).  Every entry point builds the functor through
// the reference's own `X::Create(...)` factory and calls ceres::CostFunction::Evaluate — the same surface backend.cpp drives.
// Used by tests/test_oracle_ref.py to pin oracle/factors.h (the restatement) against the reference's text, and to generate
// tests/golden/ref_v1.npz (the reference does not exist on the GPU box).
// Round 3: ImuError / Preintegration are covered too — ceres/imu_error.hpp, imu/preintegration.h, imu/imu.h, utility.h and frame.h
// are included, and src/preintegration.cpp is compiled as a second translation unit (oracle/Makefile), all UNMODIFIED, against the
// fixed-size Matrix / Quaternion / LLT / inverse stand-in in ref_shim/Eigen/Core (whose header declares the evaluation order it
// fixes: real Eigen's rounding depends on its version and vector ISA).
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "lvio_fusion/ceres/imu_error.hpp"
#include "lvio_fusion/ceres/lidar_error.hpp"
#include "lvio_fusion/ceres/pose_error.hpp"
#include "lvio_fusion/ceres/visual_error.hpp"

// statics the reference defines in its .cpp files (src/visual/camera.cpp, src/estimator.cpp); not on the functor path
namespace lvio_fusion {
std::vector<Camera::Ptr> Camera::devices_;
double Camera::baseline = 1;
std::vector<Imu::Ptr> Imu::devices_;      // src/imu/imu.cpp
}  // namespace lvio_fusion
const double epsilon = 1e-3;
const int num_threads = 1;

using namespace lvio_fusion;

extern "C" {

struct lvr_camera { double fx, fy, cx, cy; double extrinsic[7]; };

// Camera::Create appends to a process-wide registry (camera.h:18-22); the driver resets it per call so ids are 0 / 1
static Camera::Ptr make_camera(const lvr_camera* c) {
  const int id = Camera::Create(c->fx, c->fy, c->cx, c->cy, SE3d(c->extrinsic));
  return Camera::Get(id);
}

const char* lvr_sources(void) {
  return "lvio_fusion/ceres/base.hpp visual_error.hpp lidar_error.hpp pose_error.hpp imu_error.hpp imu/preintegration.h utility.h + src/preintegration.cpp (unmodified, from /root/reference)";
}

// PoseOnlyReprojectionError::Create(ob, pw, camera, weight)   visual_error.hpp:66-70 ; call site backend.cpp:129-130
void lvr_pose_only_eval(int n, const double* ob, const int* kf_idx, const int* pw_idx, const double* pw, const double* poses,
                        const double* w_kf, const lvr_camera* cam0, double* r, double* J) {
  Camera::Ptr cam = make_camera(cam0);
  for (int i = 0; i < n; ++i) {
    const double* p = pw + 3 * pw_idx[i];
    std::unique_ptr<ceres::CostFunction> f(PoseOnlyReprojectionError::Create(Vector2d(ob[2 * i], ob[2 * i + 1]), Vector3d(p[0], p[1], p[2]), cam, w_kf[kf_idx[i]]));
    const double* params[1] = {poses + 7 * kf_idx[i]};
    double* jac[1] = {J ? J + 14 * i : nullptr};
    f->Evaluate(params, r + 2 * i, J ? jac : nullptr);
  }
}

// TwoFrameReprojectionError::Create(first_ob, ob, left, right, weight)   visual_error.hpp:98-102 ; backend.cpp:138
void lvr_two_frame_eval(int n, const double* first_ob, const double* ob, const int* lm_idx, const int* kf1_idx, const int* kf2_idx,
                        const double* inv_depth, const double* poses, const double* w_kf, const lvr_camera* left_, const lvr_camera* right_,
                        double* r, double* Jd, double* J1, double* J2) {
  Camera::Ptr left = make_camera(left_), right = make_camera(right_);
  for (int i = 0; i < n; ++i) {
    std::unique_ptr<ceres::CostFunction> f(TwoFrameReprojectionError::Create(Vector2d(first_ob[2 * i], first_ob[2 * i + 1]), Vector2d(ob[2 * i], ob[2 * i + 1]),
                                                                             left, right, w_kf[kf2_idx[i]]));
    const double* params[3] = {inv_depth + lm_idx[i], poses + 7 * kf1_idx[i], poses + 7 * kf2_idx[i]};
    double* jac[3] = {Jd ? Jd + 2 * i : nullptr, J1 ? J1 + 14 * i : nullptr, J2 ? J2 + 14 * i : nullptr};
    f->Evaluate(params, r + 2 * i, (Jd || J1 || J2) ? jac : nullptr);
  }
}

// TwoCameraReprojectionError::Create(left_ob, right_ob, left, right, 5 * weights.visual)   visual_error.hpp:128-132 ; backend.cpp:123
void lvr_two_camera_eval(int n, const double* left_ob, const double* right_ob, const int* lm_idx, const int* kf_idx, const double* inv_depth,
                         const double* w_kf, const lvr_camera* left_, const lvr_camera* right_, double* r, double* J) {
  Camera::Ptr left = make_camera(left_), right = make_camera(right_);
  for (int i = 0; i < n; ++i) {
    std::unique_ptr<ceres::CostFunction> f(TwoCameraReprojectionError::Create(Vector2d(left_ob[2 * i], left_ob[2 * i + 1]), Vector2d(right_ob[2 * i], right_ob[2 * i + 1]),
                                                                              left, right, 5 * w_kf[kf_idx[i]]));
    const double* params[1] = {inv_depth + lm_idx[i]};
    double* jac[1] = {J ? J + 2 * i : nullptr};
    f->Evaluate(params, r + 2 * i, J ? jac : nullptr);
  }
}

// LidarPlaneErrorRPZ / YXY ::Create(p, pa, pb, pc, Twc1, rpyxyz (LIVE pointer), weight)   lidar_error.hpp:65-69, 100-104
// association.cpp:313-316, :371-374.  r[n], J[n][3]; nrm_out (may be null) is not available from the functor (private) — the
// normal is checked through the residual.
void lvr_lidar_plane_eval(int mode, int n, const double* p, const double* pa, const double* pb, const double* pc, const double* Twc1,
                          double* rpyxyz_live, double weight, double* r, double* J) {
  const SE3d T1(Twc1);
  const int i0 = mode == 0 ? 1 : 0, i1 = mode == 0 ? 2 : 3, i2 = mode == 0 ? 5 : 4;
  for (int i = 0; i < n; ++i) {
    const Vector3d vp(p[3 * i], p[3 * i + 1], p[3 * i + 2]), va(pa[3 * i], pa[3 * i + 1], pa[3 * i + 2]), vb(pb[3 * i], pb[3 * i + 1], pb[3 * i + 2]),
        vc(pc[3 * i], pc[3 * i + 1], pc[3 * i + 2]);
    std::unique_ptr<ceres::CostFunction> f(mode == 0 ? LidarPlaneErrorRPZ::Create(vp, va, vb, vc, T1, rpyxyz_live, weight)
                                                     : LidarPlaneErrorYXY::Create(vp, va, vb, vc, T1, rpyxyz_live, weight));
    const double* params[3] = {rpyxyz_live + i0, rpyxyz_live + i1, rpyxyz_live + i2};
    double* jac[3] = {J ? J + 3 * i : nullptr, J ? J + 3 * i + 1 : nullptr, J ? J + 3 * i + 2 : nullptr};
    f->Evaluate(params, r + i, J ? jac : nullptr);
  }
}

// LidarPlaneError::Create(p, pa, pb, pc) <1,7>   lidar_error.hpp:33-36
void lvr_lidar_plane_se3_eval(int n, const double* p, const double* pa, const double* pb, const double* pc, const double* Twc2, double* r, double* J) {
  for (int i = 0; i < n; ++i) {
    std::unique_ptr<ceres::CostFunction> f(LidarPlaneError::Create(Vector3d(p[3 * i], p[3 * i + 1], p[3 * i + 2]), Vector3d(pa[3 * i], pa[3 * i + 1], pa[3 * i + 2]),
                                                                   Vector3d(pb[3 * i], pb[3 * i + 1], pb[3 * i + 2]), Vector3d(pc[3 * i], pc[3 * i + 1], pc[3 * i + 2])));
    const double* params[1] = {Twc2};
    double* jac[1] = {J ? J + 7 * i : nullptr};
    f->Evaluate(params, r + i, J ? jac : nullptr);
  }
}

// PoseGraphError::Create(last_pose, pose, weight, v) <6,7,7>   pose_error.hpp:40-43 ; backend.cpp:170
void lvr_pose_graph_eval(const double* last_pose, const double* pose, double weight, double v, const double* Twc1, const double* Twc2, double* r,
                         double* J1, double* J2) {
  std::unique_ptr<ceres::CostFunction> f(PoseGraphError::Create(SE3d(last_pose), SE3d(pose), weight, v));
  const double* params[2] = {Twc1, Twc2};
  double* jac[2] = {J1, J2};
  f->Evaluate(params, r, (J1 || J2) ? jac : nullptr);
}
// PoseGraphError::Create(relative_i_j, weight, v)   pose_error.hpp:45-48 ; pose_graph.cpp
void lvr_pose_graph_rel_eval(const double* relative_i_j, double weight, double v, const double* Twc1, const double* Twc2, double* r, double* J1, double* J2) {
  std::unique_ptr<ceres::CostFunction> f(PoseGraphError::Create(SE3d(relative_i_j), weight, v));
  const double* params[2] = {Twc1, Twc2};
  double* jac[2] = {J1, J2};
  f->Evaluate(params, r, (J1 || J2) ? jac : nullptr);
}
// PoseError::Create(pose, weight, v) <6,7>   pose_error.hpp:78-81 ; backend.cpp:175
void lvr_pose_prior_eval(const double* origin, double weight, double v, const double* pose, double* r, double* J) {
  std::unique_ptr<ceres::CostFunction> f(PoseError::Create(SE3d(origin), weight, v));
  const double* params[1] = {pose};
  double* jac[1] = {J};
  f->Evaluate(params, r, J ? jac : nullptr);
}
// RError::Create(pose, weight) <4,7>   pose_error.hpp:102-105 ; pose_graph.cpp:192
void lvr_r_error_eval(const double* origin, double weight, const double* pose, double* r, double* J) {
  std::unique_ptr<ceres::CostFunction> f(RError::Create(SE3d(origin), weight));
  const double* params[1] = {pose};
  double* jac[1] = {J};
  f->Evaluate(params, r, J ? jac : nullptr);
}
// TError::Create(p, weight) <3,7>   pose_error.hpp:124-127
void lvr_t_error_eval(const double* p3, double weight, const double* pose, double* r, double* J) {
  std::unique_ptr<ceres::CostFunction> f(TError::Create(Vector3d(p3[0], p3[1], p3[2]), weight));
  const double* params[1] = {pose};
  double* jac[1] = {J};
  f->Evaluate(params, r, J ? jac : nullptr);
}
// PoseErrorRPZ / PoseErrorYXY ::Create(rpyxyz, weight) <3,1,1,1>   pose_error.hpp:155-158, :183-186 ; association.cpp:323,381
// x3 in PARAMETER order — (pitch, roll, z) / (yaw, x, y); J = three 3x1 blocks concatenated [block][row]
void lvr_prior3_eval(int mode, double* rpyxyz0, double weight, const double* x3, double* r, double* J9) {
  std::unique_ptr<ceres::CostFunction> f(mode == 0 ? PoseErrorRPZ::Create(rpyxyz0, weight) : PoseErrorYXY::Create(rpyxyz0, weight));
  const double* params[3] = {x3, x3 + 1, x3 + 2};
  double* jac[3] = {J9, J9 ? J9 + 3 : nullptr, J9 ? J9 + 6 : nullptr};
  f->Evaluate(params, r, J9 ? jac : nullptr);
}
// RelocateRError::Create(relocated, unrelocated) <7,4>   pose_error.hpp:216-219 ; relocator.cpp:261
void lvr_relocate_r_eval(const double* relocated, const double* unrelocated, const double* q4, double* r, double* J) {
  std::unique_ptr<ceres::CostFunction> f(RelocateRError::Create(SE3d(relocated), SE3d(unrelocated)));
  const double* params[1] = {q4};
  double* jac[1] = {J};
  f->Evaluate(params, r, J ? jac : nullptr);
}

// ---------------- IMU: imu::Preintegration (preintegration.h:16-91, preintegration.cpp:15-165) and ImuError (imu_error.hpp:12-122) ----------------
// flattened pre-integration, same field order as oracle_capi.cpp's lvo_preint / include/lvf.h's lvf_preint
struct lvr_preint {
  double sum_dt; double lin_ba[3]; double lin_bg[3]; double dp[3]; double dq[4]; double dv[3];
  double jac[225]; double cov[225];
};

// the noise densities live in the process-wide Imu device (preintegration.cpp:21-27 reads Imu::Get()); device 0 is created once
static void set_imu_noise(const double* noise4) {
  if (Imu::Num() == 0) Imu::Create(SE3d(), 0, 0, 0, 0, 9.81007);
  Imu::Ptr d = Imu::Get();
  d->ACC_N = noise4[0]; d->GYR_N = noise4[1]; d->ACC_W = noise4[2]; d->GYR_W = noise4[3];
}
static void preint_to_flat(const imu::Preintegration& P, lvr_preint* o) {
  o->sum_dt = P.sum_dt;
  for (int i = 0; i < 3; ++i) { o->lin_ba[i] = P.linearized_ba(i); o->lin_bg[i] = P.linearized_bg(i); o->dp[i] = P.delta_p(i); o->dv[i] = P.delta_v(i); }
  o->dq[0] = P.delta_q.x(); o->dq[1] = P.delta_q.y(); o->dq[2] = P.delta_q.z(); o->dq[3] = P.delta_q.w();
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { o->jac[15 * i + j] = P.jacobian(i, j); o->cov[15 * i + j] = P.covariance(i, j); }
}
// the public fields ImuError / Preintegration::Evaluate read, set from a flattened record
static imu::Preintegration::Ptr flat_to_preint(const lvr_preint* f) {
  imu::Preintegration::Ptr P = imu::Preintegration::Create(Bias(Vector3d(f->lin_ba[0], f->lin_ba[1], f->lin_ba[2]), Vector3d(f->lin_bg[0], f->lin_bg[1], f->lin_bg[2])));
  P->sum_dt = f->sum_dt;
  P->delta_p = Vector3d(f->dp[0], f->dp[1], f->dp[2]);
  P->delta_q = Quaterniond(f->dq[3], f->dq[0], f->dq[1], f->dq[2]);
  P->delta_v = Vector3d(f->dv[0], f->dv[1], f->dv[2]);
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { P->jacobian(i, j) = f->jac[15 * i + j]; P->covariance(i, j) = f->cov[15 * i + j]; }
  return P;
}
// samples[ns][7] = (dt, acc xyz, gyr xyz); acc0 / gyr0 = the measurement latched by the first Append (preintegration.h:29-36).
// Preintegration::Create(bias) then Append per sample — the sequence frontend / initializer code drives.
void lvr_imu_preintegrate(int ns, const double* samples, const double* acc0, const double* gyr0, const double* ba, const double* bg,
                          const double* noise4, lvr_preint* out) {
  set_imu_noise(noise4);
  imu::Preintegration::Ptr P = imu::Preintegration::Create(Bias(Vector3d(ba[0], ba[1], ba[2]), Vector3d(bg[0], bg[1], bg[2])));
  const Vector3d a0(acc0[0], acc0[1], acc0[2]), g0(gyr0[0], gyr0[1], gyr0[2]);
  for (int s = 0; s < ns; ++s) {
    const double* q = samples + 7 * s;
    P->Append(q[0], Vector3d(q[1], q[2], q[3]), Vector3d(q[4], q[5], q[6]), a0, g0);
  }
  if (ns == 0) { P->acc0 = a0; P->gyr0 = g0; }
  preint_to_flat(*P, out);
}
// Repropagate(ba, bg) over the buffered samples (preintegration.cpp:128-142) — same inputs as above plus the new biases
void lvr_imu_repropagate(int ns, const double* samples, const double* acc0, const double* gyr0, const double* ba, const double* bg,
                         const double* new_ba, const double* new_bg, const double* noise4, lvr_preint* out) {
  set_imu_noise(noise4);
  imu::Preintegration::Ptr P = imu::Preintegration::Create(Bias(Vector3d(ba[0], ba[1], ba[2]), Vector3d(bg[0], bg[1], bg[2])));
  const Vector3d a0(acc0[0], acc0[1], acc0[2]), g0(gyr0[0], gyr0[1], gyr0[2]);
  for (int s = 0; s < ns; ++s) {
    const double* q = samples + 7 * s;
    P->Append(q[0], Vector3d(q[1], q[2], q[3]), Vector3d(q[4], q[5], q[6]), a0, g0);
  }
  P->Repropagate(Vector3d(new_ba[0], new_ba[1], new_ba[2]), Vector3d(new_bg[0], new_bg[1], new_bg[2]));
  preint_to_flat(*P, out);
}
// Preintegration::Evaluate (the UNWEIGHTED 15-residual, preintegration.cpp:144-165)
void lvr_imu_raw_residual(int n, const lvr_preint* pre, const int* kf_i, const int* kf_j, const double* poses, const double* vel,
                          const double* ba, const double* bg, const double* noise4, double* r) {
  set_imu_noise(noise4);
  for (int f = 0; f < n; ++f) {
    imu::Preintegration::Ptr P = flat_to_preint(pre + f);
    const double *pi = poses + 7 * kf_i[f], *pj = poses + 7 * kf_j[f];
    const int i = kf_i[f], j = kf_j[f];
    Matrix<double, 15, 1> e = P->Evaluate(Vector3d(pi[4], pi[5], pi[6]), Quaterniond(pi[3], pi[0], pi[1], pi[2]), Vector3d(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]),
                                          Vector3d(ba[3 * i], ba[3 * i + 1], ba[3 * i + 2]), Vector3d(bg[3 * i], bg[3 * i + 1], bg[3 * i + 2]),
                                          Vector3d(pj[4], pj[5], pj[6]), Quaterniond(pj[3], pj[0], pj[1], pj[2]), Vector3d(vel[3 * j], vel[3 * j + 1], vel[3 * j + 2]),
                                          Vector3d(ba[3 * j], ba[3 * j + 1], ba[3 * j + 2]), Vector3d(bg[3 * j], bg[3 * j + 1], bg[3 * j + 2]));
    for (int k = 0; k < 15; ++k) r[15 * f + k] = e(k);
  }
}
// ImuError::Create(preintegration)->Evaluate   imu_error.hpp:17-118 ; call site backend.cpp:150-152.
// n factors over state arrays poses[nkf][7], vel/ba/bg[nkf][3]; r[n][15]; J packed per factor as the eight row-major blocks
// 15x7,15x3,15x3,15x3,15x7,15x3,15x3,15x3 = 480 doubles (J may be null) — oracle_capi.cpp's lvo_imu_eval layout.
void lvr_imu_eval(int n, const lvr_preint* pre, const int* kf_i, const int* kf_j, const double* poses, const double* vel, const double* ba,
                  const double* bg, const double* noise4, double* r, double* J) {
  static const int off[8] = {0, 105, 150, 195, 240, 345, 390, 435};
  set_imu_noise(noise4);
  for (int f = 0; f < n; ++f) {
    std::unique_ptr<ceres::CostFunction> fn(ImuError::Create(flat_to_preint(pre + f)));
    const int i = kf_i[f], j = kf_j[f];
    const double* prm[8] = {poses + 7 * i, vel + 3 * i, ba + 3 * i, bg + 3 * i, poses + 7 * j, vel + 3 * j, ba + 3 * j, bg + 3 * j};
    double* Jp[8];
    if (J) for (int k = 0; k < 8; ++k) Jp[k] = J + (size_t)480 * f + off[k];
    fn->Evaluate(prm, r + 15 * f, J ? Jp : nullptr);
  }
}

// The evaluation half of the reference's CPU path on a whole window, as the reference runs it: Backend::BuildProblem creates ONE HEAP FUNCTOR PER
// BLOCK through X::Create (backend.cpp:119-160; twice per tick), and every Levenberg-Marquardt iteration then calls CostFunction::Evaluate —
// residuals and all Jacobians — on each of them from up to num_threads workers (estimator.cpp:10).  This entry point does exactly that
// with the reference's own functor text (AutoDiffCostFunction over the stand-in Jet, ImuError's analytic Jacobians) and times the two parts:
// times2 = {seconds to create every functor, seconds per full evaluation pass (average of `reps`)}; checksum = sum of squared residuals (keeps
// the work alive and lets the caller compare with its own cost).  Used by bench.py's cpu_baseline (kind "reference").
void lvr_window_eval_timed(int n_tc, const double* tc_left_ob, const double* tc_right_ob, const int* tc_lm, const int* tc_kf,
                           int n_tf, const double* tf_first_ob, const double* tf_ob, const int* tf_lm, const int* tf_kf1, const int* tf_kf2,
                           int n_po, const double* po_ob, const int* po_kf, const int* po_pwi, const double* po_pw,
                           int n_imu, const lvr_preint* pre, const int* imu_i, const int* imu_j, const double* noise4,
                           const double* inv_depth, const double* poses, const double* vel, const double* ba, const double* bg, const double* w_kf,
                           const lvr_camera* left_, const lvr_camera* right_, int threads, int reps, double* times2, double* checksum) {
  struct Block { std::unique_ptr<ceres::CostFunction> f; const double* prm[8]; int np, nres, sizes[8]; };
  Camera::Ptr left = make_camera(left_), right = make_camera(right_);
  if (n_imu > 0) set_imu_noise(noise4);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<Block> blocks((size_t)n_tc + n_tf + n_po + n_imu);
  size_t k = 0;
  for (int i = 0; i < n_tc; ++i, ++k) {      // backend.cpp:123-124
    Block& b = blocks[k];
    b.f.reset(TwoCameraReprojectionError::Create(Vector2d(tc_left_ob[2 * i], tc_left_ob[2 * i + 1]), Vector2d(tc_right_ob[2 * i], tc_right_ob[2 * i + 1]), left, right, 5 * w_kf[tc_kf[i]]));
    b.np = 1; b.nres = 2; b.prm[0] = inv_depth + tc_lm[i]; b.sizes[0] = 1;
  }
  for (int i = 0; i < n_tf; ++i, ++k) {      // backend.cpp:138-139
    Block& b = blocks[k];
    b.f.reset(TwoFrameReprojectionError::Create(Vector2d(tf_first_ob[2 * i], tf_first_ob[2 * i + 1]), Vector2d(tf_ob[2 * i], tf_ob[2 * i + 1]), left, right, w_kf[tf_kf2[i]]));
    b.np = 3; b.nres = 2; b.prm[0] = inv_depth + tf_lm[i]; b.prm[1] = poses + 7 * tf_kf1[i]; b.prm[2] = poses + 7 * tf_kf2[i]; b.sizes[0] = 1; b.sizes[1] = 7; b.sizes[2] = 7;
  }
  for (int i = 0; i < n_po; ++i, ++k) {      // backend.cpp:129-130
    Block& b = blocks[k];
    const double* p = po_pw + 3 * po_pwi[i];
    b.f.reset(PoseOnlyReprojectionError::Create(Vector2d(po_ob[2 * i], po_ob[2 * i + 1]), Vector3d(p[0], p[1], p[2]), left, w_kf[po_kf[i]]));
    b.np = 1; b.nres = 2; b.prm[0] = poses + 7 * po_kf[i]; b.sizes[0] = 7;
  }
  for (int f = 0; f < n_imu; ++f, ++k) {     // backend.cpp:150-152
    Block& b = blocks[k];
    b.f.reset(ImuError::Create(flat_to_preint(pre + f)));
    const int i = imu_i[f], j = imu_j[f];
    const double* prm[8] = {poses + 7 * i, vel + 3 * i, ba + 3 * i, bg + 3 * i, poses + 7 * j, vel + 3 * j, ba + 3 * j, bg + 3 * j};
    const int sz[8] = {7, 3, 3, 3, 7, 3, 3, 3};
    b.np = 8; b.nres = 15;
    for (int q = 0; q < 8; ++q) { b.prm[q] = prm[q]; b.sizes[q] = sz[q]; }
  }
  const auto t1 = std::chrono::steady_clock::now();
  double sum = 0.0;
  const long long nb = (long long)blocks.size();
  for (int rep = 0; rep < reps; ++rep) {
    double s = 0.0;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads) reduction(+ : s)
    for (long long q = 0; q < nb; ++q) {
      const Block& b = blocks[(size_t)q];
      double r[15], J[15 * 32];
      double* jac[8];
      int off = 0;
      for (int c = 0; c < b.np; ++c) { jac[c] = J + off; off += b.nres * b.sizes[c]; }
      b.f->Evaluate(b.prm, r, jac);
      for (int c = 0; c < b.nres; ++c) s += r[c] * r[c];
    }
    sum = s;
  }
  const auto t2 = std::chrono::steady_clock::now();
  times2[0] = std::chrono::duration<double>(t1 - t0).count();
  times2[1] = std::chrono::duration<double>(t2 - t1).count() / (reps > 0 ? reps : 1);
  *checksum = sum;
}

// base.hpp helpers instantiated on double / float (the float SE3TransformPoint is the association's transform, association.cpp:289)
void lvr_se3_to_rpyxyz(const double* se3, double* rpyxyz) { ceres::SE3ToRpyxyz<double>(se3, rpyxyz); }
void lvr_rpyxyz_to_se3(const double* rpyxyz, double* se3) { ceres::RpyxyzToSE3<double>(rpyxyz, se3); }
void lvr_se3_mul(const double* A, const double* B, double* C) { ceres::SE3Product<double>(A, B, C); }
void lvr_se3_inv(const double* A, double* C) { ceres::SE3Inverse<double>(A, C); }
void lvr_se3_apply(const double* A, const double* p, double* o) { ceres::SE3TransformPoint<double>(A, p, o); }
void lvr_se3_apply_f32(const float* A, const float* p, float* o) { ceres::SE3TransformPoint<float>(A, p, o); }

}  // extern "C"
