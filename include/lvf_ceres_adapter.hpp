// lvf_ceres_adapter.hpp — lvio_fusion's own host interface for the hot path, re-seated on the MI355X library.
//
// What the reference calls                                   what this header provides (namespace lvio_fusion::gpu)
//   XError::Create(...)  -> ceres::CostFunction*               same class names, same factory argument meaning, same
//     include/lvio_fusion/ceres/visual_error.hpp:66-70,98-102,128-132     template sizes (SizedCostFunction<...>); the object
//     lidar_error.hpp:65-69,100-104  imu_error.hpp:110-113                carries the functor's constructor constants and a
//     pose_error.hpp:40-49,78-81,155-158,183-186                          type tag instead of a Jet-differentiated operator()
//   CostFunction::Evaluate(parameters, residuals, jacobians)   batch-of-one evaluation ON THE GPU through the C-ABI
//     (Ceres calls it per block, per thread)                     (parity / debugging surface; not the fast path)
//   adapt::Solve(options, &problem, &summary)                  gpu::Solve — the single choke point (adapt/problem.h:83-88):
//     src/backend.cpp:211,267  src/mapping.cpp:162,177,276,290   walks the ceres::Problem, recognises the tagged blocks,
//                                                                uploads them as SoA batches, runs the device LM loop and
//                                                                writes the result back IN PLACE into the caller's arrays
//   ceres::Problem::Evaluate(opts, &cost, &residuals, ...)     gpu::Evaluate — batched cost + residual vector
//
// Eigen / Sophus / OpenCV types do not appear: factories take plain double arrays (Vector2d::data(), SE3d::data(),
// lvf_camera); INTEGRATION.md shows the one-line wrappers that restore the reference's exact signatures.
//
// Failure is SOFT, as the reference expects (it never reads Summary::termination_type, backend.cpp:210-211): on any
// error parameters are left untouched, Summary::termination_type = FAILURE and Summary::message says why.  There is no
// CPU solver behind this header: an unsupported problem is reported, not silently solved elsewhere.
#ifndef LVF_CERES_ADAPTER_HPP_
#define LVF_CERES_ADAPTER_HPP_

#if defined(__has_include)
#if __has_include(<ceres/ceres.h>) && !defined(LVF_FORCE_CERES_COMPAT)
#include <ceres/ceres.h>
#define LVF_HAVE_REAL_CERES 1
#endif
#endif
#ifndef LVF_HAVE_REAL_CERES
#include "lvf_ceres_compat.h"
#endif

#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <typeinfo>
#include <cstdint>
#include <thread>
#include <unordered_map>
#include <new>
#include <vector>

#include "lvf.h"

namespace lvio_fusion {
namespace gpu {

// --------------------------------------------------------------------------------------------- context per host thread
// Backend::Optimize and Relocator::CorrectLoop -> Mapping::Relocate can be in flight at once (relocator.cpp:188), and
// Ceres calls Evaluate from its worker threads: every host thread gets its own context (= its own HIP stream).
// Device: LVF_DEVICE environment variable, default 0.
struct ThreadContext {
  lvf_ctx* ctx = nullptr;
  std::string error;
  ThreadContext() {
    const char* d = std::getenv("LVF_DEVICE");
    if (lvf_ctx_create(d ? std::atoi(d) : 0, nullptr, &ctx) != LVF_OK) { error = lvf_last_error(); ctx = nullptr; }
  }
  ~ThreadContext() { if (ctx) lvf_ctx_destroy(ctx); }
};
inline ThreadContext& thread_context() {
  static thread_local ThreadContext tc;
  return tc;
}

enum class Kind { PoseOnly, TwoFrame, TwoCamera, Imu, LidarPlaneRPZ, LidarPlaneYXY, PoseErrorRPZ, PoseErrorYXY, PoseGraph, Pose, R, RelocateR };

// RAII for the C handles used inside one call
struct Handles {
  std::vector<lvf_batch*> batches;
  lvf_state* st = nullptr;
  lvf_problem* prob = nullptr;
  ~Handles() {
    if (prob) lvf_problem_destroy(prob);
    for (auto* b : batches) if (b) lvf_batch_destroy(b);
    if (st) lvf_state_destroy(st);
  }
  lvf_batch* keep(lvf_batch* b) { batches.push_back(b); return b; }
};

// --------------------------------------------------------------------------------------------- tagged cost functions
class GpuCostFunction {
 public:
  virtual ~GpuCostFunction() {}
  virtual Kind kind() const = 0;
};

namespace detail {
inline bool fetch(lvf_batch* b, int n_blocks, const int* sizes, int n_res, double* residuals, double** jacobians) {
  if (lvf_batch_download_residuals(b, residuals) != LVF_OK) return false;
  if (jacobians)
    for (int k = 0; k < n_blocks; ++k)
      if (jacobians[k] && lvf_batch_download_jacobian(b, k, jacobians[k]) != LVF_OK) return false;
  (void)sizes; (void)n_res;
  return true;
}
inline bool set_state(lvf_state* st, int field, const double* v) { return lvf_state_set(st, field, v) == LVF_OK; }
}  // namespace detail

// PoseOnlyReprojectionError <2,7>  visual_error.hpp:48-76 ; Create(ob, pw, camera, weight) :66
class PoseOnlyReprojectionError : public ceres::SizedCostFunction<2, 7>, public GpuCostFunction {
 public:
  PoseOnlyReprojectionError(const double ob[2], const double pw[3], const lvf_camera& camera, double weight) : cam(camera), weight(weight) {
    std::memcpy(this->ob, ob, sizeof(this->ob)); std::memcpy(this->pw, pw, sizeof(this->pw));
  }
  static ceres::CostFunction* Create(const double ob[2], const double pw[3], const lvf_camera& camera, double weight) {
    return new PoseOnlyReprojectionError(ob, pw, camera, weight);
  }
  Kind kind() const override { return Kind::PoseOnly; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    const int32_t zero = 0;
    if (lvf_state_create(tc.ctx, 1, 0, &h.st) != LVF_OK) return false;
    if (!detail::set_state(h.st, LVF_POSES, parameters[0]) || !detail::set_state(h.st, LVF_W_VISUAL, &weight)) return false;
    lvf_batch* b = nullptr;
    if (lvf_pose_only_create(tc.ctx, &cam, 1, ob, &zero, &zero, 1, pw, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, h.st, nullptr, jacobians != nullptr) != LVF_OK) return false;
    const int sizes[1] = {7};
    return detail::fetch(b, 1, sizes, 2, residuals, jacobians);
  }
  double ob[2], pw[3];
  lvf_camera cam;
  double weight;
};

// TwoFrameReprojectionError <2,1,7,7>  visual_error.hpp:78-107 ; Create(first_ob, ob, left, right, weight) :98
class TwoFrameReprojectionError : public ceres::SizedCostFunction<2, 1, 7, 7>, public GpuCostFunction {
 public:
  TwoFrameReprojectionError(const double first_ob[2], const double ob[2], const lvf_camera& left, const lvf_camera& right, double weight)
      : left(left), right(right), weight(weight) {
    std::memcpy(this->first_ob, first_ob, sizeof(this->first_ob)); std::memcpy(this->ob, ob, sizeof(this->ob));
  }
  static ceres::CostFunction* Create(const double first_ob[2], const double ob[2], const lvf_camera& left, const lvf_camera& right, double weight) {
    return new TwoFrameReprojectionError(first_ob, ob, left, right, weight);
  }
  Kind kind() const override { return Kind::TwoFrame; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    const int32_t zero = 0, one = 1;
    if (lvf_state_create(tc.ctx, 2, 1, &h.st) != LVF_OK) return false;
    double poses[14], w2[2] = {weight, weight};
    std::memcpy(poses, parameters[1], 56); std::memcpy(poses + 7, parameters[2], 56);
    if (!detail::set_state(h.st, LVF_POSES, poses) || !detail::set_state(h.st, LVF_W_VISUAL, w2) ||
        !detail::set_state(h.st, LVF_INV_DEPTH, parameters[0])) return false;
    lvf_batch* b = nullptr;
    if (lvf_two_frame_create(tc.ctx, &left, &right, 1, first_ob, ob, &zero, &zero, &one, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, h.st, nullptr, jacobians != nullptr) != LVF_OK) return false;
    const int sizes[3] = {1, 7, 7};
    return detail::fetch(b, 3, sizes, 2, residuals, jacobians);
  }
  double first_ob[2], ob[2];
  lvf_camera left, right;
  double weight;
};

// TwoCameraReprojectionError <2,1>  visual_error.hpp:109-137 ; Create(left_ob, right_ob, left, right, weight) :128
// (weight is what the caller passes — backend.cpp:123 passes 5 * frame->weights.visual)
class TwoCameraReprojectionError : public ceres::SizedCostFunction<2, 1>, public GpuCostFunction {
 public:
  TwoCameraReprojectionError(const double left_ob[2], const double right_ob[2], const lvf_camera& left, const lvf_camera& right, double weight)
      : left(left), right(right), weight(weight) {
    std::memcpy(this->left_ob, left_ob, sizeof(this->left_ob)); std::memcpy(this->right_ob, right_ob, sizeof(this->right_ob));
  }
  static ceres::CostFunction* Create(const double left_ob[2], const double right_ob[2], const lvf_camera& left, const lvf_camera& right, double weight) {
    return new TwoCameraReprojectionError(left_ob, right_ob, left, right, weight);
  }
  Kind kind() const override { return Kind::TwoCamera; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    const int32_t zero = 0;
    if (lvf_state_create(tc.ctx, 1, 1, &h.st) != LVF_OK) return false;
    const double wk = weight / 5.0;   // the library applies the reference's 5x to the per-keyframe visual weight
    if (!detail::set_state(h.st, LVF_W_VISUAL, &wk) || !detail::set_state(h.st, LVF_INV_DEPTH, parameters[0])) return false;
    lvf_batch* b = nullptr;
    if (lvf_two_camera_create(tc.ctx, &left, &right, 1, left_ob, right_ob, &zero, &zero, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, h.st, nullptr, jacobians != nullptr) != LVF_OK) return false;
    const int sizes[1] = {1};
    return detail::fetch(b, 1, sizes, 2, residuals, jacobians);
  }
  double left_ob[2], right_ob[2];
  lvf_camera left, right;
  double weight;
};

// ImuError : SizedCostFunction<15,7,3,3,3,7,3,3,3>  imu_error.hpp:12-122 ; Create(preintegration) :110
// The reference holds a shared_ptr to the live Preintegration (read-only during a solve); here the snapshot is copied.
class ImuError : public ceres::SizedCostFunction<15, 7, 3, 3, 3, 7, 3, 3, 3>, public GpuCostFunction {
 public:
  explicit ImuError(const lvf_preint& preintegration) : pre(preintegration) {}
  static ceres::CostFunction* Create(const lvf_preint& preintegration) { return new ImuError(preintegration); }
  Kind kind() const override { return Kind::Imu; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    const int32_t zero = 0, one = 1;
    if (lvf_state_create(tc.ctx, 2, 0, &h.st) != LVF_OK) return false;
    double poses[14], v[6], ba[6], bg[6];
    std::memcpy(poses, parameters[0], 56); std::memcpy(poses + 7, parameters[4], 56);
    std::memcpy(v, parameters[1], 24); std::memcpy(v + 3, parameters[5], 24);
    std::memcpy(ba, parameters[2], 24); std::memcpy(ba + 3, parameters[6], 24);
    std::memcpy(bg, parameters[3], 24); std::memcpy(bg + 3, parameters[7], 24);
    if (!detail::set_state(h.st, LVF_POSES, poses) || !detail::set_state(h.st, LVF_VEL, v) || !detail::set_state(h.st, LVF_BA, ba) ||
        !detail::set_state(h.st, LVF_BG, bg)) return false;
    lvf_batch* b = nullptr;
    if (lvf_imu_create(tc.ctx, 1, &pre, &zero, &one, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, h.st, nullptr, jacobians != nullptr) != LVF_OK) return false;
    const int sizes[8] = {7, 3, 3, 3, 7, 3, 3, 3};
    return detail::fetch(b, 8, sizes, 15, residuals, jacobians);
  }
  lvf_preint pre;
};

// LidarPlaneErrorRPZ / LidarPlaneErrorYXY <1,1,1,1>  lidar_error.hpp:42-110
// Create(p, pa, pb, pc, Twc1, rpyxyz, weight) :65,:100 — rpyxyz is the caller's LIVE array (lidar_error.hpp:52,87).
class LidarPlaneErrorBase : public ceres::SizedCostFunction<1, 1, 1, 1>, public GpuCostFunction {
 public:
  LidarPlaneErrorBase(int mode, const double p[3], const double pa[3], const double pb[3], const double pc[3], const double Twc1[7],
                      double* rpyxyz, double weight) : mode(mode), rpyxyz(rpyxyz), weight(weight) {
    std::memcpy(this->p, p, 24); std::memcpy(this->pa, pa, 24); std::memcpy(this->pb, pb, 24); std::memcpy(this->pc, pc, 24);
    std::memcpy(this->Twc1, Twc1, 56);
  }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    double x[6];
    std::memcpy(x, rpyxyz, sizeof(x));
    const int s0 = mode == 0 ? 1 : 0, s1 = mode == 0 ? 2 : 3, s2 = mode == 0 ? 5 : 4;
    x[s0] = parameters[0][0]; x[s1] = parameters[1][0]; x[s2] = parameters[2][0];
    lvf_batch* b = nullptr;
    if (lvf_lidar_plane_create(tc.ctx, mode, 1, p, pa, pb, pc, Twc1, weight, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, nullptr, x, jacobians != nullptr) != LVF_OK) return false;
    const int sizes[3] = {1, 1, 1};
    return detail::fetch(b, 3, sizes, 1, residuals, jacobians);
  }
  int mode;
  double p[3], pa[3], pb[3], pc[3], Twc1[7];
  double* rpyxyz;
  double weight;
};
class LidarPlaneErrorRPZ : public LidarPlaneErrorBase {
 public:
  using LidarPlaneErrorBase::LidarPlaneErrorBase;
  static ceres::CostFunction* Create(const double p[3], const double pa[3], const double pb[3], const double pc[3], const double Twc1[7],
                                     double* rpyxyz, double weight) { return new LidarPlaneErrorRPZ(0, p, pa, pb, pc, Twc1, rpyxyz, weight); }
  Kind kind() const override { return Kind::LidarPlaneRPZ; }
};
class LidarPlaneErrorYXY : public LidarPlaneErrorBase {
 public:
  using LidarPlaneErrorBase::LidarPlaneErrorBase;
  static ceres::CostFunction* Create(const double p[3], const double pa[3], const double pb[3], const double pc[3], const double Twc1[7],
                                     double* rpyxyz, double weight) { return new LidarPlaneErrorYXY(1, p, pa, pb, pc, Twc1, rpyxyz, weight); }
  Kind kind() const override { return Kind::LidarPlaneYXY; }
};

// PoseErrorRPZ / PoseErrorYXY <3,1,1,1>  pose_error.hpp:135-190 ; Create(rpyxyz, weight): the target is COPIED at
// construction (:139-141, :167-169).  target[] is kept in parameter order (p,r,z) / (Y,x,y).
class PoseError3Base : public ceres::SizedCostFunction<3, 1, 1, 1>, public GpuCostFunction {
 public:
  PoseError3Base(int mode, const double* rpyxyz, double weight) : mode(mode), weight(weight) {
    if (mode == 0) { target[0] = rpyxyz[1]; target[1] = rpyxyz[2]; target[2] = rpyxyz[5]; }
    else { target[0] = rpyxyz[0]; target[1] = rpyxyz[3]; target[2] = rpyxyz[4]; }
  }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    const double x[3] = {parameters[0][0], parameters[1][0], parameters[2][0]};
    double J[9];
    if (lvf_prior3_evaluate(tc.ctx, mode, target, weight, x, residuals, jacobians ? J : nullptr) != LVF_OK) return false;
    if (jacobians) for (int k = 0; k < 3; ++k) if (jacobians[k]) std::memcpy(jacobians[k], J + 3 * k, 24);
    return true;
  }
  int mode;
  double target[3];
  double weight;
};
class PoseErrorRPZ : public PoseError3Base {
 public:
  using PoseError3Base::PoseError3Base;
  static ceres::CostFunction* Create(double* rpyxyz, double weight = 1) { return new PoseErrorRPZ(0, rpyxyz, weight); }
  Kind kind() const override { return Kind::PoseErrorRPZ; }
};
class PoseErrorYXY : public PoseError3Base {
 public:
  using PoseError3Base::PoseError3Base;
  static ceres::CostFunction* Create(double* rpyxyz, double weight = 1) { return new PoseErrorYXY(1, rpyxyz, weight); }
  Kind kind() const override { return Kind::PoseErrorYXY; }
};

// PoseGraphError <6,7,7>  pose_error.hpp:10-53 ; Create(last_pose, pose, weight, v) :40 / Create(relative_i_j, weight, v) :45
class PoseGraphError : public ceres::SizedCostFunction<6, 7, 7>, public GpuCostFunction {
 public:
  PoseGraphError(const double rpyxyz_target[6], double weight, double v) : weight(weight), v(v) { std::memcpy(target, rpyxyz_target, 48); target[6] = 0.0; }
  static ceres::CostFunction* Create(const double last_pose[7], const double pose[7], double weight = 1, double v = 1) {
    double t[6];
    if (lvf_relative_rpyxyz(last_pose, pose, t) != LVF_OK) return nullptr;
    return new PoseGraphError(t, weight, v);
  }
  static ceres::CostFunction* Create(const double relative_i_j[7], double weight = 1, double v = 1) {
    const double id[7] = {0, 0, 0, 1, 0, 0, 0};
    return Create(id, relative_i_j, weight, v);
  }
  Kind kind() const override { return Kind::PoseGraph; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    const int32_t zero = 0, one = 1;
    if (lvf_state_create(tc.ctx, 2, 0, &h.st) != LVF_OK) return false;
    double poses[14];
    std::memcpy(poses, parameters[0], 56); std::memcpy(poses + 7, parameters[1], 56);
    if (!detail::set_state(h.st, LVF_POSES, poses)) return false;
    lvf_batch* b = nullptr;
    if (lvf_pose_prior_create(tc.ctx, 1, &zero, &one, target, &weight, &v, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, h.st, nullptr, jacobians != nullptr) != LVF_OK) return false;
    const int sizes[2] = {7, 7};
    return detail::fetch(b, 2, sizes, 6, residuals, jacobians);
  }
  double target[7];
  double weight, v;
};

// PoseError <6,7>  pose_error.hpp:55-86 ; Create(pose, weight, v) :78
class PoseError : public ceres::SizedCostFunction<6, 7>, public GpuCostFunction {
 public:
  PoseError(const double pose[7], double weight, double v) : weight(weight), v(v) { std::memcpy(origin, pose, 56); }
  static ceres::CostFunction* Create(const double pose[7], double weight = 1, double v = 1) { return new PoseError(pose, weight, v); }
  Kind kind() const override { return Kind::Pose; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    const int32_t minus = -1, zero = 0;
    if (lvf_state_create(tc.ctx, 1, 0, &h.st) != LVF_OK) return false;
    if (!detail::set_state(h.st, LVF_POSES, parameters[0])) return false;
    lvf_batch* b = nullptr;
    if (lvf_pose_prior_create(tc.ctx, 1, &minus, &zero, origin, &weight, &v, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, h.st, nullptr, jacobians != nullptr) != LVF_OK) return false;
    if (lvf_batch_download_residuals(b, residuals) != LVF_OK) return false;
    if (jacobians && jacobians[0] && lvf_batch_download_jacobian(b, 1, jacobians[0]) != LVF_OK) return false;   // block 1 = the pose
    return true;
  }
  double origin[7];
  double weight, v;
};

// RError <4,7>  pose_error.hpp:88-110 ; Create(pose, weight) :102 — the quaternion prior of PoseGraph::BuildProblem
// (src/pose_graph.cpp:190-191)
class RError : public ceres::SizedCostFunction<4, 7>, public GpuCostFunction {
 public:
  RError(const double pose[7], double weight) : weight(weight) { std::memcpy(origin, pose, 56); }
  static ceres::CostFunction* Create(const double pose[7], double weight = 1) { return new RError(pose, weight); }
  Kind kind() const override { return Kind::R; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    Handles h;
    const int32_t minus2 = -2, zero = 0;
    const double v = 0.0;
    if (lvf_state_create(tc.ctx, 1, 0, &h.st) != LVF_OK) return false;
    if (!detail::set_state(h.st, LVF_POSES, parameters[0])) return false;
    lvf_batch* b = nullptr;
    if (lvf_pose_prior_create(tc.ctx, 1, &minus2, &zero, origin, &weight, &v, &b) != LVF_OK) return false;
    h.keep(b);
    if (lvf_batch_evaluate(b, h.st, nullptr, jacobians != nullptr) != LVF_OK) return false;
    double r6[6], J[42];
    if (lvf_batch_download_residuals(b, r6) != LVF_OK) return false;
    std::memcpy(residuals, r6, 32);                                   // rows 4,5 of the batch are padding
    if (jacobians && jacobians[0]) {
      if (lvf_batch_download_jacobian(b, 1, J) != LVF_OK) return false;
      std::memcpy(jacobians[0], J, 28 * sizeof(double));
    }
    return true;
  }
  double origin[7];
  double weight;
};

// RelocateRError <7,4>  pose_error.hpp:192-222 ; Create(relocated, unrelocated) :215 — the blocks of Relocator::UpdateNewSubmap's rotation solve
// (src/relocator.cpp:258-264: one quaternion parameter block under EigenQuaternionParameterization, one block per keyframe of the new sub-map)
class RelocateRError : public ceres::SizedCostFunction<7, 4>, public GpuCostFunction {
 public:
  RelocateRError(const double relocated_[7], const double unrelocated_[7]) { std::memcpy(relocated, relocated_, 56); std::memcpy(unrelocated, unrelocated_, 56); }
  static ceres::CostFunction* Create(const double relocated[7], const double unrelocated[7]) { return new RelocateRError(relocated, unrelocated); }
  Kind kind() const override { return Kind::RelocateR; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    ThreadContext& tc = thread_context();
    if (!tc.ctx) return false;
    return lvf_relocate_r_evaluate(tc.ctx, 1, relocated, unrelocated, parameters[0], residuals, (jacobians && jacobians[0]) ? jacobians[0] : nullptr) == LVF_OK;
  }
  double relocated[7], unrelocated[7];
};

// --------------------------------------------------------------------------------------------- problem -> device
namespace detail {

// HuberLoss(a) / TrivialLoss / NULL are told apart through the LossFunction's own public Evaluate: for s > a^2 Huber has
// rho'(s) = a / sqrt(s) (so rho'(4s) = rho'(s)/2), Trivial has rho' = 1.  Returns a (0 = no robustification), or -1 if
// the loss is neither.
inline double probe_huber(const ceres::LossFunction* loss) {
  if (!loss) return 0.0;
  double r1[3], r2[3];
  const double s = 1e12;
  loss->Evaluate(s, r1); loss->Evaluate(4.0 * s, r2);
  if (r1[1] == 1.0 && r2[1] == 1.0) return 0.0;
  const double a = r1[1] * std::sqrt(s);
  if (a > 0.0 && std::fabs(r2[1] * 2.0 - r1[1]) <= 1e-12 * r1[1]) {
    double r0[3];
    loss->Evaluate(0.25 * a * a, r0);   // inside the quadratic zone rho(s) = s
    if (std::fabs(r0[0] - 0.25 * a * a) <= 1e-12 * a * a && r0[1] == 1.0) return a;
  }
  return -1.0;
}

inline bool same_cam(const lvf_camera& a, const lvf_camera& b) { return std::memcmp(&a, &b, sizeof(lvf_camera)) == 0; }

struct Fail {
  ceres::Solver::Summary* s;
  bool operator()(const std::string& why) const {
    s->termination_type = ceres::FAILURE;
    s->message = "lvf: " + why;
    return false;
  }
};

struct BlockView {
  const ceres::CostFunction* cf;
  const GpuCostFunction* g;
  const ceres::LossFunction* loss;
  double* const* params;      // into the caller's flat pointer store (one allocation for the whole problem, not one per block)
};

// dynamic_cast<const GpuCostFunction*> is a cross-cast (the tag is a second base): libstdc++ walks the class graph for every call,
// several hundred ns.  The offset of the tag sub-object is a constant of the most-derived type, so it is looked up once per
// dynamic type (typeid compares a pointer read from the vtable) and applied by hand afterwards.
struct CastCache {
  const std::type_info* type[16];
  std::ptrdiff_t offset[16];
  int n = 0;
};
inline const GpuCostFunction* as_gpu(const ceres::CostFunction* cf, CastCache& c) {
  const std::type_info& ti = typeid(*cf);
  for (int k = 0; k < c.n; ++k)
    if (c.type[k] == &ti) return c.offset[k] == PTRDIFF_MIN ? nullptr : reinterpret_cast<const GpuCostFunction*>(reinterpret_cast<const char*>(cf) + c.offset[k]);
  const GpuCostFunction* g = dynamic_cast<const GpuCostFunction*>(cf);
  if (c.n < 16) {
    c.type[c.n] = &ti;
    c.offset[c.n] = g ? reinterpret_cast<const char*>(g) - reinterpret_cast<const char*>(cf) : PTRDIFF_MIN;
    ++c.n;
  }
  return g;
}

// Walking a large problem through the Ceres accessors is pointer chasing over heap objects (one cost function, one residual-block
// record and one pointer vector per block): ~130 ns per block on one core, 12-15 ms for the 91 k blocks of a 50-keyframe window.
// The walk only READS the problem, so it is split over a few threads (const accessors of ceres::Problem are safe for concurrent
// readers); each also pulls the cost-function object into cache for the classification pass that follows.
inline bool collect(ceres::Problem* problem, std::vector<BlockView>* out, std::vector<double*>* flat, const Fail& fail) {
  std::vector<ceres::ResidualBlockId> ids;
  problem->GetResidualBlocks(&ids);
  const size_t n = ids.size();
  out->assign(n, BlockView{nullptr, nullptr, nullptr, nullptr});
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t T = n < 4096 ? 1 : std::max<size_t>(1, std::min<size_t>(16, (hw ? hw : 2) / 2));   // latency-bound: threads buy memory-level parallelism
  std::vector<std::vector<double*>> part(T);
  std::vector<std::vector<size_t>> poff(T);
  std::vector<size_t> bad(T, n);
  auto work = [&](size_t t) {
    const size_t b = n * t / T, e = n * (t + 1) / T;
    part[t].reserve((e - b) * 3);
    poff[t].reserve(e - b);
    std::vector<double*> scratch;
    CastCache cache;
    for (size_t i = b; i < e; ++i) {
      BlockView& v = (*out)[i];
      if (i + 8 < e) __builtin_prefetch(ids[i + 8]);      // the residual-block record a few iterations ahead
      v.cf = problem->GetCostFunctionForResidualBlock(ids[i]);
      v.g = as_gpu(v.cf, cache);
      if (!v.g) { bad[t] = i; return; }
      __builtin_prefetch(reinterpret_cast<const char*>(v.g) + 64);
      __builtin_prefetch(reinterpret_cast<const char*>(v.g) + 128);
      v.loss = problem->GetLossFunctionForResidualBlock(ids[i]);
      problem->GetParameterBlocksForResidualBlock(ids[i], &scratch);
      poff[t].push_back(part[t].size());
      part[t].insert(part[t].end(), scratch.begin(), scratch.end());
    }
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  for (size_t t = 0; t < T; ++t)
    if (bad[t] < n) return fail("residual block " + std::to_string(bad[t]) + " is not an lvio_fusion::gpu cost function (no CPU solver is linked)");
  size_t total = 0;
  for (size_t t = 0; t < T; ++t) total += part[t].size();
  flat->resize(total);
  size_t base = 0;
  for (size_t t = 0; t < T; ++t) {
    std::copy(part[t].begin(), part[t].end(), flat->begin() + base);
    const size_t b = n * t / T;
    for (size_t k = 0; k < poff[t].size(); ++k) (*out)[b + k].params = flat->data() + base + poff[t][k];
    base += part[t].size();
  }
  return true;
}

// ---- scan-to-map sub-problem: N x LidarPlaneError{RPZ|YXY} + at most one PoseError{RPZ|YXY}
inline bool solve_lidar(const ceres::Solver::Options& options, ceres::Problem* problem, const std::vector<BlockView>& blocks,
                        ceres::Solver::Summary* summary, const Fail& fail) {
  ThreadContext& tc = thread_context();
  if (!tc.ctx) return fail("no usable GPU context: " + tc.error);
  int mode = -1;
  const LidarPlaneErrorBase* first = nullptr;
  const PoseError3Base* prior = nullptr;
  double* prm[3] = {nullptr, nullptr, nullptr};
  std::vector<double> p, pa, pb, pc;
  const ceres::LossFunction* loss = nullptr;
  for (const BlockView& b : blocks) {
    const Kind k = b.g->kind();
    const int m = (k == Kind::LidarPlaneRPZ || k == Kind::PoseErrorRPZ) ? 0 : 1;
    if (mode < 0) mode = m;
    if (m != mode) return fail("RPZ and YXY blocks mixed in one scan-to-map problem");
    if (!prm[0]) { prm[0] = b.params[0]; prm[1] = b.params[1]; prm[2] = b.params[2]; }
    if (b.params[0] != prm[0] || b.params[1] != prm[1] || b.params[2] != prm[2]) return fail("scan-to-map blocks do not share their three parameter blocks");
    if (k == Kind::LidarPlaneRPZ || k == Kind::LidarPlaneYXY) {
      const auto* f = static_cast<const LidarPlaneErrorBase*>(b.g);
      if (!first) { first = f; loss = b.loss; }
      if (std::memcmp(f->Twc1, first->Twc1, 56) != 0 || f->weight != first->weight || f->rpyxyz != first->rpyxyz)
        return fail("lidar blocks of one problem must share Twc1, weight and the live rpyxyz array");
      if (b.loss != loss) return fail("lidar blocks of one problem must share one loss function");
      p.insert(p.end(), f->p, f->p + 3); pa.insert(pa.end(), f->pa, f->pa + 3);
      pb.insert(pb.end(), f->pb, f->pb + 3); pc.insert(pc.end(), f->pc, f->pc + 3);
    } else {
      if (prior) return fail("more than one PoseErrorRPZ/YXY prior");
      if (b.loss && probe_huber(b.loss) != 0.0) return fail("a robust loss on the PoseErrorRPZ/YXY prior is not supported");
      prior = static_cast<const PoseError3Base*>(b.g);
    }
  }
  for (int k = 0; k < 3; ++k)
    if (problem->IsParameterBlockConstant(prm[k])) return fail("constant parameter blocks are not supported in the scan-to-map problem");
  const double huber = probe_huber(loss);
  if (huber < 0.0) return fail("unsupported loss function (only HuberLoss / TrivialLoss / NULL)");
  const int s0 = mode == 0 ? 1 : 0, s1 = mode == 0 ? 2 : 3, s2 = mode == 0 ? 5 : 4;
  double x[6] = {0, 0, 0, 0, 0, 0};
  const double id7[7] = {0, 0, 0, 1, 0, 0, 0};
  if (first) std::memcpy(x, first->rpyxyz, sizeof(x));
  x[s0] = *prm[0]; x[s1] = *prm[1]; x[s2] = *prm[2];
  if (prior && (prior->target[0] != x[s0] || prior->target[1] != x[s1] || prior->target[2] != x[s2]))
    return fail("the PoseErrorRPZ/YXY target must equal the initial parameter values (as mapping.cpp builds it)");
  Handles h;
  lvf_batch* b = nullptr;
  const int n = (int)(p.size() / 3);
  if (lvf_lidar_plane_create(tc.ctx, mode, n, p.data(), pa.data(), pb.data(), pc.data(), first ? first->Twc1 : id7, first ? first->weight : 1.0, &b) != LVF_OK)
    return fail(lvf_last_error());
  h.keep(b);
  lvf_icp_options o;
  std::memset(&o, 0, sizeof(o));
  o.mode = mode; o.huber_a = huber; o.prior_weight = prior ? prior->weight : 0.0; o.max_num_iterations = options.max_num_iterations;
  lvf_icp_summary s;
  if (lvf_lidar_solve(b, x, &o, &s) != LVF_OK) return fail(lvf_last_error());
  *prm[0] = x[s0]; *prm[1] = x[s1]; *prm[2] = x[s2];          // in place, like ceres::Solve
  summary->initial_cost = s.initial_cost; summary->final_cost = s.final_cost;
  summary->num_successful_steps = s.num_successful_steps; summary->num_unsuccessful_steps = s.num_iterations - s.num_successful_steps;
  summary->num_residual_blocks = summary->num_residual_blocks_reduced = (int)blocks.size();
  summary->num_parameter_blocks = summary->num_parameter_blocks_reduced = 3;
  summary->termination_type = s.num_iterations >= options.max_num_iterations ? ceres::NO_CONVERGENCE : ceres::CONVERGENCE;
  summary->message = "scan-to-map sub-problem solved on device";
  return true;
}

// pointer -> dense id, open addressing (the classification pass does two or three look-ups per residual block; std::unordered_map's
// node chasing was a third of its time)
struct PtrMap {
  std::vector<double*> key;
  std::vector<int> val;
  size_t mask = 0, used = 0;
  void reserve(size_t n) {
    size_t cap = 64;
    while (cap < 2 * n + 2) cap <<= 1;
    if (cap <= key.size()) return;
    std::vector<double*> ok; std::vector<int> ov;
    ok.swap(key); ov.swap(val);
    key.assign(cap, nullptr); val.assign(cap, -1); mask = cap - 1; used = 0;
    for (size_t i = 0; i < ok.size(); ++i) if (ok[i]) insert(ok[i], ov[i]);
  }
  static size_t hash(const double* p) { return (size_t)((reinterpret_cast<uintptr_t>(p) >> 3) * 0x9E3779B97F4A7C15ull >> 17); }
  int find(double* p) const {
    if (key.empty()) return -1;
    for (size_t h = hash(p) & mask;; h = (h + 1) & mask) {
      if (key[h] == p) return val[h];
      if (!key[h]) return -1;
    }
  }
  void insert(double* p, int v) {
    if (2 * (used + 1) > key.size()) reserve(used + 1 + key.size() / 2);
    for (size_t h = hash(p) & mask;; h = (h + 1) & mask) {
      if (!key[h]) { key[h] = p; val[h] = v; ++used; return; }
      if (key[h] == p) { val[h] = v; return; }
    }
  }
};

// Payload arrays live in page-locked memory (lvf_host_alloc): they are filled block by block while BuildProblem runs and then handed to
// lvf_*_create, whose uploads are then plain DMA instead of pageable copies (the runtime pins and unpins a pageable source around
// every copy: 0.5 ms of a 91 k-block adapt::Solve).
template <class T>
struct PinnedAllocator {
  using value_type = T;
  PinnedAllocator() = default;
  template <class U> PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(std::size_t n) { void* p = lvf_host_alloc(n * sizeof(T)); if (!p) throw std::bad_alloc(); return static_cast<T*>(p); }
  void deallocate(T* p, std::size_t n) { lvf_host_free(p, n * sizeof(T)); }
  template <class U> bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <class U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
template <class T> using PinVec = std::vector<T, PinnedAllocator<T>>;

// ---- sliding-window BA: the device image of what Backend::BuildProblem registered
struct Window {
  std::vector<double*> pose_ptr, lm_ptr;
  PtrMap pose_id, lm_id;
  std::vector<double*> v_ptr, ba_ptr, bg_ptr;      // per keyframe, null if the frame has no IMU blocks
  std::vector<double> w_kf;
  std::vector<char> w_known;
  std::vector<char> pose_const;                    // per keyframe: SetParameterBlockConstant was called on its pose block
  std::vector<char> vbb_const;                     // per keyframe, bit 0 / 1 / 2: ... on its velocity / accelerometer-bias / gyroscope-bias block
  // batches (insertion order preserved per type)
  PinVec<double> tc_l, tc_r; PinVec<int32_t> tc_lm, tc_kf; PinVec<double> tc_w;
  PinVec<double> tf_f, tf_o; PinVec<int32_t> tf_lm, tf_k1, tf_k2;
  PinVec<double> po_o, po_pw; PinVec<int32_t> po_kf, po_pi;
  PinVec<lvf_preint> imu_pre; PinVec<int32_t> imu_i, imu_j;
  std::vector<int32_t> pr_a, pr_b; std::vector<double> pr_t, pr_w, pr_v;
  std::vector<int> order_kind, order_idx;          // per residual block: which batch, which row
  lvf_camera left, right; bool have_left = false, have_right = false;
  double huber = -2.0;                             // -2 = not seen yet
  // The TwoFrame blocks' shape, proved block by block while they arrive (WindowBuilder::add): sorted by current keyframe, first keyframe
  // < current keyframe, one block per (landmark, current keyframe), one first keyframe per landmark — what Backend::BuildProblem always
  // produces (backend.cpp:105-140).  While it holds the device side is told so (lvf_two_frame_set_shape) and skips its own pass over the blocks.
  bool tf_shape_ok = true; int tf_last_k2 = -1;
  std::vector<int32_t> tf_per_k2;                  // blocks per current keyframe
  std::vector<int32_t> lm_first_kf, lm_last_k2;    // per landmark (index into lm_ptr): first keyframe of its TwoFrame blocks, current keyframe of the last one
};

// what can only be checked once every block is in: parameter blocks the device solver cannot hold constant individually.
// `constant` (recorder path) lists the blocks SetParameterBlockConstant was called on; NULL = ask the problem (walk path).
inline bool finish_window(ceres::Problem* problem, Window* w, const PtrMap* constant, const Fail& fail) {
  if (w->huber == -2.0) w->huber = 0.0;
  const int n_kf = (int)w->pose_ptr.size();
  // Poses and velocity/bias blocks (<= 4 per keyframe) always ask the problem itself: it stays right even when a caller reaches
  // SetParameterBlockConstant through a ceres::Problem& and the recorder's shadow never sees the call.  Only the ~10 k inverse-depth
  // blocks take the recorder's list, to keep 10 k Ceres map lookups off the solve.
  auto is_const = [&](double* p) { return problem->IsParameterBlockConstant(p); };
  auto lm_const = [&](double* p) { return constant ? constant->find(p) >= 0 : problem->IsParameterBlockConstant(p); };
  w->pose_const.assign(n_kf, 0);
  w->vbb_const.assign(n_kf, 0);
  if (!constant || constant->used)
    for (double* p : w->lm_ptr) if (lm_const(p)) return fail("constant inverse-depth blocks are not supported");
  for (int k = 0; k < n_kf; ++k) {
    w->pose_const[k] = is_const(w->pose_ptr[k]) ? 1 : 0;
    // constant velocity / bias blocks (Environment::Optimize holds the frame's and the previous frame's, environment.cpp:62-68)
    w->vbb_const[k] = (char)((w->v_ptr[k] && is_const(w->v_ptr[k]) ? 1 : 0) | (w->ba_ptr[k] && is_const(w->ba_ptr[k]) ? 2 : 0) | (w->bg_ptr[k] && is_const(w->bg_ptr[k]) ? 4 : 0));
  }
  return true;
}

// One residual block at a time into the device image of the window.  Used by BOTH paths: the walk over a finished ceres::Problem
// (build_window below) and the recorder that adapt::Problem feeds while Backend::BuildProblem is still adding blocks (Recorder), so that
// gpu::Solve finds the SoA payload ready instead of chasing 91 k heap objects through the Ceres accessors.
struct WindowBuilder {
  Window* w;
  std::string error;                              // first failure (the block's index and the rule it broke)
  const ceres::LossFunction* loss_seen = nullptr;  // BuildProblem shares ONE loss object: probe each distinct pointer once
  bool loss_seen_ok = false, loss_seen_any = false;
  explicit WindowBuilder(Window* w_) : w(w_) {}

  void add_pose_block(double* p) {
    if (w->pose_id.find(p) >= 0) return;
    w->pose_id.insert(p, (int)w->pose_ptr.size()); w->pose_ptr.push_back(p);
    w->v_ptr.push_back(nullptr); w->ba_ptr.push_back(nullptr); w->bg_ptr.push_back(nullptr);
    w->w_kf.push_back(1.0); w->w_known.push_back(0);
  }
  int kf_of(double* p) const { return w->pose_id.find(p); }
  int lm_of(double* p) {
    const int hit = w->lm_id.find(p);
    if (hit >= 0) return hit;
    const int id = (int)w->lm_ptr.size();
    w->lm_id.insert(p, id); w->lm_ptr.push_back(p);
    return id;
  }
  bool set_w(int kf, double wv) {
    if (w->w_known[kf] && w->w_kf[kf] != wv) return false;
    w->w_kf[kf] = wv; w->w_known[kf] = 1;
    return true;
  }
  static bool use_cam(const lvf_camera& c, lvf_camera* slot, bool* have) {
    if (!*have) { *slot = c; *have = true; return true; }
    return same_cam(*slot, c);
  }
  bool use_loss(const ceres::LossFunction* loss) {
    if (loss_seen_any && loss == loss_seen) return loss_seen_ok;
    const double a = probe_huber(loss);
    bool ok = a >= 0.0;
    if (ok) {
      if (w->huber == -2.0) w->huber = a;
      ok = w->huber == a;
    }
    loss_seen = loss; loss_seen_ok = ok; loss_seen_any = true;
    return ok;
  }
  static bool bind3(std::vector<double*>& slot, int kf, double* p) {
    if (slot[kf] && slot[kf] != p) return false;
    slot[kf] = p;
    return true;
  }
  bool err(size_t i, const char* why) {
    if (error.empty()) error = "residual block " + std::to_string(i) + ": " + why;      // only built on the failure path
    return false;
  }
  // block i of the problem (insertion order): cost function g, its loss and its parameter pointers
  bool add(size_t i, const GpuCostFunction* g, const ceres::LossFunction* loss, double* const* params) {
    switch (g->kind()) {
      case Kind::PoseOnly: {
        const auto* f = static_cast<const PoseOnlyReprojectionError*>(g);
        const int kf = kf_of(params[0]);
        if (kf < 0) return err(i, "pose block not registered");
        if (!set_w(kf, f->weight)) return err(i, "visual blocks of one keyframe must share one weight (frame->weights.visual)");
        if (!use_cam(f->cam, &w->left, &w->have_left)) return err(i, "all blocks must share the left camera");
        if (!use_loss(loss)) return err(i, "visual blocks must share one HuberLoss/TrivialLoss");
        w->order_kind.push_back(2); w->order_idx.push_back((int)w->po_kf.size());
        w->po_o.insert(w->po_o.end(), f->ob, f->ob + 2); w->po_pw.insert(w->po_pw.end(), f->pw, f->pw + 3);
        w->po_pi.push_back((int)w->po_kf.size()); w->po_kf.push_back(kf);
        break;
      }
      case Kind::TwoFrame: {
        const auto* f = static_cast<const TwoFrameReprojectionError*>(g);
        const int k1 = kf_of(params[1]), k2 = kf_of(params[2]);
        if (k1 < 0 || k2 < 0) return err(i, "pose block not registered");
        if (!set_w(k2, f->weight)) return err(i, "visual blocks of one keyframe must share one weight (frame->weights.visual)");
        if (!use_cam(f->left, &w->left, &w->have_left) || !use_cam(f->right, &w->right, &w->have_right)) return err(i, "all blocks must share the stereo pair");
        if (!use_loss(loss)) return err(i, "visual blocks must share one HuberLoss/TrivialLoss");
        w->order_kind.push_back(1); w->order_idx.push_back((int)w->tf_lm.size());
        w->tf_f.insert(w->tf_f.end(), f->first_ob, f->first_ob + 2); w->tf_o.insert(w->tf_o.end(), f->ob, f->ob + 2);
        const int lm = lm_of(params[0]);
        w->tf_lm.push_back(lm); w->tf_k1.push_back(k1); w->tf_k2.push_back(k2);
        if (w->tf_shape_ok) {
          if ((size_t)lm >= w->lm_first_kf.size()) { w->lm_first_kf.resize((size_t)lm + 1 + w->lm_first_kf.size() / 2, -1); w->lm_last_k2.resize(w->lm_first_kf.size(), -1); }
          if ((size_t)k2 >= w->tf_per_k2.size()) w->tf_per_k2.resize((size_t)k2 + 1, 0);
          const bool ok = k2 >= w->tf_last_k2 && k1 < k2 && w->lm_last_k2[lm] < k2 && (w->lm_first_kf[lm] < 0 || w->lm_first_kf[lm] == k1);
          w->tf_shape_ok = ok;
          w->tf_last_k2 = k2; w->lm_last_k2[lm] = k2; w->lm_first_kf[lm] = k1; ++w->tf_per_k2[k2];
        }
        break;
      }
      case Kind::TwoCamera: {
        const auto* f = static_cast<const TwoCameraReprojectionError*>(g);
        if (!use_cam(f->left, &w->left, &w->have_left) || !use_cam(f->right, &w->right, &w->have_right)) return err(i, "all blocks must share the stereo pair");
        if (!use_loss(loss)) return err(i, "visual blocks must share one HuberLoss/TrivialLoss");
        w->order_kind.push_back(0); w->order_idx.push_back((int)w->tc_lm.size());
        w->tc_l.insert(w->tc_l.end(), f->left_ob, f->left_ob + 2); w->tc_r.insert(w->tc_r.end(), f->right_ob, f->right_ob + 2);
        // the block carries its own weight (5 * frame->weights.visual at backend.cpp:123): handed to the device per block, no keyframe look-up
        w->tc_lm.push_back(lm_of(params[0])); w->tc_kf.push_back(0); w->tc_w.push_back(f->weight);
        break;
      }
      case Kind::Imu: {
        const auto* f = static_cast<const ImuError*>(g);
        const int ki = kf_of(params[0]), kj = kf_of(params[4]);
        if (ki < 0 || kj < 0) return err(i, "pose block not registered");
        if (loss && probe_huber(loss) != 0.0) return err(i, "a robust loss on ImuError is not supported (the reference passes NULL)");
        if (!bind3(w->v_ptr, ki, params[1]) || !bind3(w->ba_ptr, ki, params[2]) || !bind3(w->bg_ptr, ki, params[3]) ||
            !bind3(w->v_ptr, kj, params[5]) || !bind3(w->ba_ptr, kj, params[6]) || !bind3(w->bg_ptr, kj, params[7]))
          return err(i, "a keyframe is linked to two different velocity/bias blocks");
        w->order_kind.push_back(3); w->order_idx.push_back((int)w->imu_i.size());
        w->imu_pre.push_back(f->pre); w->imu_i.push_back(ki); w->imu_j.push_back(kj);
        break;
      }
      case Kind::PoseGraph: {
        const auto* f = static_cast<const PoseGraphError*>(g);
        const int ka = kf_of(params[0]), kb = kf_of(params[1]);
        if (ka < 0 || kb < 0) return err(i, "pose block not registered");
        if (loss && probe_huber(loss) != 0.0) return err(i, "a robust loss on PoseGraphError is not supported (the reference passes NULL)");
        w->order_kind.push_back(4); w->order_idx.push_back((int)w->pr_a.size());
        w->pr_a.push_back(ka); w->pr_b.push_back(kb); w->pr_t.insert(w->pr_t.end(), f->target, f->target + 7);
        w->pr_w.push_back(f->weight); w->pr_v.push_back(f->v);
        break;
      }
      case Kind::Pose: {
        const auto* f = static_cast<const PoseError*>(g);
        const int kb = kf_of(params[0]);
        if (kb < 0) return err(i, "pose block not registered");
        if (loss && probe_huber(loss) != 0.0) return err(i, "a robust loss on PoseError is not supported (the reference passes NULL)");
        w->order_kind.push_back(4); w->order_idx.push_back((int)w->pr_a.size());
        w->pr_a.push_back(-1); w->pr_b.push_back(kb); w->pr_t.insert(w->pr_t.end(), f->origin, f->origin + 7);
        w->pr_w.push_back(f->weight); w->pr_v.push_back(f->v);
        break;
      }
      case Kind::R: {
        const auto* f = static_cast<const RError*>(g);
        const int kb = kf_of(params[0]);
        if (kb < 0) return err(i, "pose block not registered");
        if (loss && probe_huber(loss) != 0.0) return err(i, "a robust loss on RError is not supported (the reference passes NULL)");
        w->order_kind.push_back(4); w->order_idx.push_back((int)w->pr_a.size());
        w->pr_a.push_back(-2); w->pr_b.push_back(kb); w->pr_t.insert(w->pr_t.end(), f->origin, f->origin + 7);
        w->pr_w.push_back(f->weight); w->pr_v.push_back(0.0);
        break;
      }
      default: return err(i, "lidar blocks cannot be mixed into a BA window");
    }
    return true;
  }
};

inline bool build_window(ceres::Problem* problem, const std::vector<BlockView>& blocks, Window* w, const Fail& fail) {
  // keyframes = 7-sized parameter blocks in the order the caller registered them (BuildProblem walks frames in time order)
  WindowBuilder wb(w);
  std::vector<double*> all;
  problem->GetParameterBlocks(&all);
  for (double* p : all)
    if (problem->ParameterBlockSize(p) == 7) wb.add_pose_block(p);
  if (w->pose_ptr.empty()) return fail("no pose parameter blocks in the problem");
  w->order_kind.reserve(blocks.size()); w->order_idx.reserve(blocks.size());
  w->lm_id.reserve(blocks.size() / 4);
  for (size_t i = 0; i < blocks.size(); ++i)
    if (!wb.add(i, blocks[i].g, blocks[i].loss, blocks[i].params)) return fail(wb.error);
  return finish_window(problem, w, nullptr, fail);
}

struct DeviceWindow {
  Handles h;
  lvf_batch *tc = nullptr, *tf = nullptr, *po = nullptr, *imu = nullptr, *prior = nullptr;
};

inline bool upload_window(lvf_ctx* ctx, ceres::Problem* problem, const Window& w, DeviceWindow* d, const Fail& fail) {
  const int n_kf = (int)w.pose_ptr.size(), n_lm = (int)w.lm_ptr.size();
  using clk = std::chrono::steady_clock;
  const bool timing = std::getenv("LVF_ADAPTER_TIMING") != nullptr;
  const auto u0 = clk::now();
  std::vector<double> poses(7 * (size_t)n_kf), vel(3 * (size_t)n_kf, 0.0), ba(vel), bg(vel), invd(n_lm);
  for (int k = 0; k < n_kf; ++k) {
    std::memcpy(&poses[7 * (size_t)k], w.pose_ptr[k], 56);
    if (w.v_ptr[k]) std::memcpy(&vel[3 * (size_t)k], w.v_ptr[k], 24);
    if (w.ba_ptr[k]) std::memcpy(&ba[3 * (size_t)k], w.ba_ptr[k], 24);
    if (w.bg_ptr[k]) std::memcpy(&bg[3 * (size_t)k], w.bg_ptr[k], 24);
  }
  for (int l = 0; l < n_lm; ++l) invd[l] = *w.lm_ptr[l];
  if (lvf_state_create_from(ctx, n_kf, n_lm, poses.data(), vel.data(), ba.data(), bg.data(), n_lm ? invd.data() : nullptr, w.w_kf.data(), &d->h.st) != LVF_OK) return fail(lvf_last_error());
  lvf_state* st = d->h.st;
  const auto u1 = clk::now();
  if (!w.tc_lm.empty()) {
    if (lvf_two_camera_create(ctx, &w.left, &w.right, (int)w.tc_lm.size(), w.tc_l.data(), w.tc_r.data(), w.tc_lm.data(), w.tc_kf.data(), &d->tc) != LVF_OK)
      return fail(lvf_last_error());
    d->h.keep(d->tc);
    if (lvf_two_camera_set_block_weights(d->tc, w.tc_w.data()) != LVF_OK) return fail(lvf_last_error());
  }
  if (!w.tf_lm.empty()) {
    if (lvf_two_frame_create(ctx, &w.left, &w.right, (int)w.tf_lm.size(), w.tf_f.data(), w.tf_o.data(), w.tf_lm.data(), w.tf_k1.data(), w.tf_k2.data(), &d->tf) != LVF_OK)
      return fail(lvf_last_error());
    d->h.keep(d->tf);
    if (w.tf_shape_ok) {             // BuildProblem's shape, proved while the blocks arrived: the device side need not walk them again
      std::vector<int32_t> per(w.tf_per_k2);
      per.resize((size_t)n_kf, 0);
      if (lvf_two_frame_set_shape(d->tf, n_kf, per.data()) != LVF_OK) return fail(lvf_last_error());
    }
  }
  if (!w.po_kf.empty()) {
    if (lvf_pose_only_create(ctx, &w.left, (int)w.po_kf.size(), w.po_o.data(), w.po_kf.data(), w.po_pi.data(), (int)w.po_kf.size(), w.po_pw.data(), &d->po) != LVF_OK)
      return fail(lvf_last_error());
    d->h.keep(d->po);
  }
  if (!w.imu_i.empty()) {
    if (lvf_imu_create(ctx, (int)w.imu_i.size(), w.imu_pre.data(), w.imu_i.data(), w.imu_j.data(), &d->imu) != LVF_OK) return fail(lvf_last_error());
    d->h.keep(d->imu);
  }
  if (!w.pr_b.empty()) {
    if (lvf_pose_prior_create(ctx, (int)w.pr_b.size(), w.pr_a.data(), w.pr_b.data(), w.pr_t.data(), w.pr_w.data(), w.pr_v.data(), &d->prior) != LVF_OK)
      return fail(lvf_last_error());
    d->h.keep(d->prior);
  }
  const auto u2 = clk::now();
  if (lvf_problem_create(ctx, st, d->tc, d->tf, d->po, d->imu, &d->h.prob) != LVF_OK) return fail(lvf_last_error());
  if (timing) std::fprintf(stderr, "  upload ms: state %.3f | batches %.3f | problem_create %.3f\n", 1e3 * std::chrono::duration<double>(u1 - u0).count(),
                           1e3 * std::chrono::duration<double>(u2 - u1).count(), 1e3 * std::chrono::duration<double>(clk::now() - u2).count());
  if (d->prior && lvf_problem_set_pose_priors(d->h.prob, d->prior) != LVF_OK) return fail(lvf_last_error());
  for (int k = 0; k < n_kf; ++k) {
    if (w.pose_const[k] && lvf_problem_set_pose_constant(d->h.prob, k, 1) != LVF_OK) return fail(lvf_last_error());
    if (w.vbb_const[k] && lvf_problem_set_vbb_constant(d->h.prob, k, w.vbb_const[k] & 1, (w.vbb_const[k] >> 1) & 1, (w.vbb_const[k] >> 2) & 1) != LVF_OK) return fail(lvf_last_error());
  }
  return true;
}

inline void to_lvf_options(const ceres::Solver::Options& o, double huber, lvf_solver_options* out) {
  lvf_solver_options_default(out);
  out->max_num_iterations = o.max_num_iterations;
  out->max_solver_time_in_seconds = o.max_solver_time_in_seconds >= 1e8 ? 0.0 : o.max_solver_time_in_seconds;
  out->huber_a = huber;
  out->initial_trust_region_radius = o.initial_trust_region_radius;
  out->function_tolerance = o.function_tolerance;
  out->gradient_tolerance = o.gradient_tolerance;
  out->parameter_tolerance = o.parameter_tolerance;
  out->min_relative_decrease = o.min_relative_decrease;
}

// ---- Relocator::UpdateNewSubmap's rotation solve (relocator.cpp:251-268): N x RelocateRError on ONE quaternion block (x, y, z, w) under
// EigenQuaternionParameterization, no loss; the whole LM loop is one device launch (lvf_relocate_rotation_solve), the quaternion is updated in place
inline bool solve_relocate_rotation(const ceres::Solver::Options& options, ceres::Problem* problem, const std::vector<BlockView>& blocks,
                                    ceres::Solver::Summary* summary, const Fail& fail) {
  ThreadContext& tc = thread_context();
  if (!tc.ctx) return fail("no device context: " + tc.error);
  double* q = blocks[0].params[0];
  std::vector<double> rel(7 * blocks.size()), un(7 * blocks.size());
  for (size_t i = 0; i < blocks.size(); ++i) {
    if (blocks[i].g->kind() != Kind::RelocateR) return fail("RelocateRError blocks mixed with other cost functions");
    if (blocks[i].params[0] != q) return fail("RelocateRError blocks on different parameter blocks");
    if (blocks[i].loss) return fail("a loss function on a RelocateRError block");
    const RelocateRError* e = static_cast<const RelocateRError*>(blocks[i].g);
    std::memcpy(&rel[7 * i], e->relocated, 56); std::memcpy(&un[7 * i], e->unrelocated, 56);
  }
  if (problem->ParameterBlockSize(q) != 4 || problem->GetParameterization(q) == nullptr) return fail("the rotation block must be a 4-block with EigenQuaternionParameterization");
  if (problem->IsParameterBlockConstant(q)) {
    summary->termination_type = ceres::CONVERGENCE; summary->num_residual_blocks = (int)blocks.size(); summary->num_residual_blocks_reduced = 0;
    summary->num_successful_steps = summary->num_unsuccessful_steps = 0; summary->message = "constant block"; return true;
  }
  lvf_solver_options o;
  to_lvf_options(options, 0.0, &o);
  lvf_solver_summary s;
  if (lvf_relocate_rotation_solve(tc.ctx, (int)blocks.size(), rel.data(), un.data(), q, &o, &s) != LVF_OK) return fail(lvf_last_error());
  summary->initial_cost = s.initial_cost; summary->final_cost = s.final_cost;
  summary->num_successful_steps = s.num_successful_steps; summary->num_unsuccessful_steps = s.num_unsuccessful_steps;
  summary->num_residual_blocks = summary->num_residual_blocks_reduced = (int)blocks.size();
  summary->num_parameter_blocks = summary->num_parameter_blocks_reduced = 1;
  if (s.termination == 2) return fail("the rotation solve failed (five invalid steps in a row)");
  summary->termination_type = s.termination == 0 ? ceres::CONVERGENCE : ceres::NO_CONVERGENCE;
  summary->message = "lvf_relocate_rotation_solve";
  return true;
}

inline bool solve_window(const ceres::Solver::Options& options, ceres::Problem* problem, const Window& w,
                         ceres::Solver::Summary* summary, const Fail& fail, double classify_seconds) {
  ThreadContext& tc = thread_context();
  if (!tc.ctx) return fail("no usable GPU context: " + tc.error);
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  const auto t0b = t0;
  DeviceWindow d;
  if (!upload_window(tc.ctx, problem, w, &d, fail)) return false;
  lvf_solver_options o;
  to_lvf_options(options, w.huber, &o);
  lvf_solver_summary s;
  const auto t1 = clk::now();
  if (lvf_problem_solve(d.h.prob, &o, &s) != LVF_OK) return fail(lvf_last_error());
  const auto t2 = clk::now();
  if (s.termination == 2) return fail("device LM failed (normal equations not positive definite at the smallest trust region)");
  // read back, then write IN PLACE into the caller's parameter arrays (frame->pose.data(), &landmark->inv_depth, ...)
  const int n_kf = (int)w.pose_ptr.size(), n_lm = (int)w.lm_ptr.size();
  std::vector<double> poses(7 * (size_t)n_kf), vel(3 * (size_t)n_kf), ba(vel), bg(vel), invd(n_lm);
  lvf_state* st = d.h.st;
  if (lvf_state_get(st, LVF_POSES, poses.data()) || lvf_state_get(st, LVF_VEL, vel.data()) || lvf_state_get(st, LVF_BA, ba.data()) ||
      lvf_state_get(st, LVF_BG, bg.data()) || (n_lm && lvf_state_get(st, LVF_INV_DEPTH, invd.data()))) return fail(lvf_last_error());
  for (int k = 0; k < n_kf; ++k) {
    std::memcpy(w.pose_ptr[k], &poses[7 * (size_t)k], 56);
    if (w.v_ptr[k]) std::memcpy(w.v_ptr[k], &vel[3 * (size_t)k], 24);
    if (w.ba_ptr[k]) std::memcpy(w.ba_ptr[k], &ba[3 * (size_t)k], 24);
    if (w.bg_ptr[k]) std::memcpy(w.bg_ptr[k], &bg[3 * (size_t)k], 24);
  }
  for (int l = 0; l < n_lm; ++l) *w.lm_ptr[l] = invd[l];
  summary->initial_cost = s.initial_cost; summary->final_cost = s.final_cost;
  summary->num_successful_steps = s.num_successful_steps; summary->num_unsuccessful_steps = s.num_unsuccessful_steps;
  summary->num_residual_blocks = summary->num_residual_blocks_reduced = s.num_residual_blocks;
  summary->num_parameter_blocks = summary->num_parameter_blocks_reduced = problem->NumParameterBlocks();
  summary->termination_type = s.termination == 0 ? ceres::CONVERGENCE : ceres::NO_CONVERGENCE;
  static const char* const kWhy[] = {"", "Gradient tolerance reached.", "Parameter tolerance reached.", "Function tolerance reached.",
                                    "Minimum trust region radius reached.", "Maximum number of iterations reached.",
                                    "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps.",
                                    "Maximum solver time reached.", "In-launch hand-over timed out and the un-chained retry could not run."};
  summary->message = std::string("sliding-window BA solved on device. ") + kWhy[s.termination_reason >= 0 && s.termination_reason <= 8 ? s.termination_reason : 0];
  summary->preprocessor_time_in_seconds += classify_seconds + std::chrono::duration<double>(t1 - t0).count();   // classify + upload (collect() is added by Solve)
  summary->minimizer_time_in_seconds = std::chrono::duration<double>(t2 - t1).count();
  summary->postprocessor_time_in_seconds = std::chrono::duration<double>(clk::now() - t2).count();
  if (std::getenv("LVF_ADAPTER_TIMING"))
    std::fprintf(stderr, "gpu::Solve (window) ms: classify %.3f | upload + configure %.3f | device loop %.3f | read back + write in place %.3f\n",
                 1e3 * classify_seconds, 1e3 * std::chrono::duration<double>(t1 - t0b).count(),
                 1e3 * std::chrono::duration<double>(t2 - t1).count(), 1e3 * summary->postprocessor_time_in_seconds);
  return true;
}

}  // namespace detail

// --------------------------------------------------------------------------------------------- recorder (fed while the problem is built)
// adapt::Problem (the reference-owned wrapper, adapt/problem.h:34-81) forwards every AddParameterBlock / AddResidualBlock /
// SetParameterBlockConstant to one of these before handing the call to ceres::Problem (INTEGRATION.md shows the three one-line hooks).
// Each lvio_fusion::gpu cost function's payload is appended to the window's SoA arrays right there — the object is hot in cache, it
// was created a few instructions earlier — so gpu::Solve uploads arrays it already has instead of walking 91 k heap cost functions
// through the Ceres accessors afterwards (6.4 of the 12.7 ms of an adapt::Solve call on the 50-keyframe window).  Anything the recorder
// cannot follow (a foreign cost function, a pose block used before it was registered, lidar blocks, a block removed again) only marks
// it unusable: gpu::Solve then falls back to the walk, which also produces the diagnostics.
class Recorder {
 public:
  Recorder() : builder_(&w_) {}
  Recorder(const Recorder&) = delete;
  Recorder& operator=(const Recorder&) = delete;
  void AddParameterBlock(double* values, int size) { if (size == 7) builder_.add_pose_block(values); }
  void AddResidualBlock(const ceres::CostFunction* cost_function, const ceres::LossFunction* loss_function, double* const* parameter_blocks, int /*num_parameter_blocks*/) {
    const size_t i = n_blocks_++;
    if (!usable_) return;
    const GpuCostFunction* g = detail::as_gpu(cost_function, cache_);
    if (!g) { usable_ = false; return; }
    const Kind k = g->kind();
    if (k == Kind::LidarPlaneRPZ || k == Kind::LidarPlaneYXY || k == Kind::PoseErrorRPZ || k == Kind::PoseErrorYXY || k == Kind::RelocateR) { usable_ = false; return; }   // scan-to-map / rotation problems: few blocks, walked
    if (!builder_.add(i, g, loss_function, parameter_blocks)) usable_ = false;
  }
  void SetParameterBlockConstant(double* values) { constant_.insert(values, 1); }
  void Invalidate() { usable_ = false; }            // SetParameterBlockVariable, RemoveResidualBlock, RemoveParameterBlock ...
  bool usable(ceres::Problem* problem) const { return usable_ && n_blocks_ > 0 && n_blocks_ == (size_t)problem->NumResidualBlocks(); }
  detail::Window& window() { return w_; }
  const detail::PtrMap& constant_blocks() const { return constant_; }
  size_t num_residual_blocks() const { return n_blocks_; }

 private:
  detail::Window w_;
  detail::WindowBuilder builder_;
  detail::CastCache cache_;
  detail::PtrMap constant_;
  size_t n_blocks_ = 0;
  bool usable_ = true;
};

// --------------------------------------------------------------------------------------------- the adapt::Solve body
inline void Solve(const ceres::Solver::Options& options, ceres::Problem* problem, ceres::Solver::Summary* summary, Recorder* recorder = nullptr) {
  const auto t0 = std::chrono::steady_clock::now();
  *summary = ceres::Solver::Summary();
  const detail::Fail fail{summary};
  const bool timing = std::getenv("LVF_ADAPTER_TIMING") != nullptr;
  if (recorder && recorder->usable(problem) && !std::getenv("LVF_ADAPTER_WALK")) {      // (LVF_ADAPTER_WALK=1: force the accessor walk, for A/B tests)
    // the window was assembled while the blocks were added: only the end-of-build checks remain
    detail::Window& w = recorder->window();
    if (w.pose_ptr.empty()) { fail("no pose parameter blocks in the problem"); return; }
    if (!detail::finish_window(problem, &w, &recorder->constant_blocks(), fail)) return;
    const double pre = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (timing) std::fprintf(stderr, "gpu::Solve ms: recorded window (%zu residual blocks captured while the problem was built) %.3f\n", recorder->num_residual_blocks(), 1e3 * pre);
    detail::solve_window(options, problem, w, summary, fail, pre);
    summary->total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return;
  }
  std::vector<detail::BlockView> blocks;
  std::vector<double*> block_params;
  if (!detail::collect(problem, &blocks, &block_params, fail)) return;
  summary->preprocessor_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (timing) std::fprintf(stderr, "gpu::Solve ms: collect (walk over %zu residual blocks) %.3f\n", blocks.size(), 1e3 * summary->preprocessor_time_in_seconds);
  if (blocks.empty()) {
    summary->termination_type = ceres::CONVERGENCE; summary->initial_cost = summary->final_cost = 0.0;
    summary->num_successful_steps = summary->num_unsuccessful_steps = 0; summary->num_residual_blocks = summary->num_residual_blocks_reduced = 0;
    summary->message = "empty problem";
    return;
  }
  if (blocks[0].g->kind() == Kind::RelocateR) { detail::solve_relocate_rotation(options, problem, blocks, summary, fail); summary->total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); return; }
  bool lidar = false;
  for (const auto& b : blocks) {
    const Kind k = b.g->kind();
    if (k == Kind::LidarPlaneRPZ || k == Kind::LidarPlaneYXY || k == Kind::PoseErrorRPZ || k == Kind::PoseErrorYXY) { lidar = true; break; }
  }
  if (lidar) detail::solve_lidar(options, problem, blocks, summary, fail);
  else {
    const auto c0 = std::chrono::steady_clock::now();
    detail::Window w;
    if (detail::build_window(problem, blocks, &w, fail))
      detail::solve_window(options, problem, w, summary, fail, std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count());
  }
  summary->total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// --------------------------------------------------------------------------------------------- batched Problem::Evaluate
// The upstream surface, on the GPU:  ceres::Problem::Evaluate(EvaluateOptions, double* cost, vector<double>* residuals,
// vector<double>* gradient, CRSMatrix* jacobian) for a sliding-window BA problem built from lvio_fusion::gpu cost functions.
//   cost      = 1/2 sum rho(|r_b|^2)                      (apply_loss_function = false: 1/2 sum |r_b|^2)
//   residuals = blocks in insertion order, with the loss function's Corrector applied when apply_loss_function is set
//   jacobian  = CRS, rows = residuals, columns = the parameter blocks in the order they were added (or options.parameter_blocks) in
//               LOCAL coordinates (pose blocks: 6 tangent columns of ProductParameterization(EigenQuaternion, Identity3)); blocks that
//               are constant in the problem keep their (empty) columns, blocks not listed in options.parameter_blocks are held constant
//               and get no columns — upstream's rules
//   gradient  = J^T r in the same column order.
// Any output pointer may be null.  options.residual_blocks must be empty (whole problem).  Values are produced on device
// (lvf_batch_evaluate_local / lvf_problem_gradient); the host only scatters them into the caller's containers.
inline bool Evaluate(ceres::Problem* problem, const ceres::Problem::EvaluateOptions& options, double* cost, std::vector<double>* residuals,
                     std::vector<double>* gradient, ceres::CRSMatrix* jacobian, std::string* error = nullptr) {
  ceres::Solver::Summary dummy;
  const detail::Fail fail{&dummy};
  auto bail = [&](const std::string& why = std::string()) { if (error) *error = why.empty() ? dummy.message : why; return false; };
  if (!options.residual_blocks.empty()) return bail("lvf: Evaluate on a subset of residual blocks is not supported");
  std::vector<detail::BlockView> blocks;
  std::vector<double*> block_params;
  if (!detail::collect(problem, &blocks, &block_params, fail)) return bail();
  ThreadContext& tc = thread_context();
  if (!tc.ctx) return bail(tc.error);
  detail::Window w;
  if (!detail::build_window(problem, blocks, &w, fail)) return bail();
  detail::DeviceWindow d;
  if (!detail::upload_window(tc.ctx, problem, w, &d, fail)) return bail();
  lvf_solver_options o;
  detail::to_lvf_options(ceres::Solver::Options(), options.apply_loss_function ? w.huber : 0.0, &o);
  if (cost && lvf_problem_cost(d.h.prob, &o, cost) != LVF_OK) return bail(lvf_last_error());
  if (!residuals && !gradient && !jacobian) return true;

  const int n_kf = (int)w.pose_ptr.size(), n_lm = (int)w.lm_ptr.size();
  // ---- column layout
  std::vector<double*> col_blocks = options.parameter_blocks;
  if (col_blocks.empty()) problem->GetParameterBlocks(&col_blocks);
  detail::PtrMap col_of;
  col_of.reserve(col_blocks.size());
  std::vector<int> col_start(col_blocks.size() + 1, 0);
  for (size_t i = 0; i < col_blocks.size(); ++i) {
    double* p = col_blocks[i];
    if (!problem->HasParameterBlock(p)) return bail("lvf: options.parameter_blocks names a block that is not in the problem");
    const int size = problem->ParameterBlockSize(p);
    const ceres::LocalParameterization* lp = problem->GetParameterization(p);
    const int local = lp ? lp->LocalSize() : size;
    if (size == 7 && local != 6) return bail("lvf: pose blocks must carry ProductParameterization(EigenQuaternionParameterization, IdentityParameterization(3))");
    if (size != 7 && local != size) return bail("lvf: only pose blocks may carry a local parameterization");
    col_of.insert(p, (int)i);
    col_start[i + 1] = col_start[i] + local;
  }
  const int num_cols = col_start.back();
  auto active_col = [&](double* p) {            // first column of block p, or -1 if it contributes no entries
    const int i = col_of.find(p);
    if (i < 0 || problem->IsParameterBlockConstant(p)) return -1;
    return col_start[i];
  };

  // ---- device evaluation, one call per functor batch
  lvf_batch* bs[5] = {d.tc, d.tf, d.po, d.imu, d.prior};
  const int nres[5] = {2, 2, 2, 15, 6};
  std::vector<double> r[5], J[5];
  int L[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < 5; ++k) {
    if (!bs[k]) continue;
    const size_t n = (size_t)lvf_batch_size(bs[k]);
    L[k] = lvf_batch_local_columns(bs[k]);
    r[k].resize(n * nres[k]);
    if (jacobian) J[k].resize(n * nres[k] * L[k]);
    if (lvf_batch_evaluate_local(bs[k], d.h.st, o.huber_a, r[k].data(), jacobian ? J[k].data() : nullptr) != LVF_OK) return bail(lvf_last_error());
  }
  // slice of each parameter block inside a batch row: TwoCamera [rho]; TwoFrame [rho | pose1 | pose2]; PoseOnly [pose];
  // ImuError [pose_i v_i ba_i bg_i pose_j v_j ba_j bg_j]; priors [pose_a | pose_b]
  static const int off_tc[1] = {0}, off_tf[3] = {0, 1, 7}, off_po[1] = {0}, off_imu[8] = {0, 6, 9, 12, 15, 21, 24, 27}, off_pg[2] = {0, 6}, off_p1[1] = {6};
  static const int wid_tc[1] = {1}, wid_tf[3] = {1, 6, 6}, wid_po[1] = {6}, wid_imu[8] = {6, 3, 3, 3, 6, 3, 3, 3}, wid_pg[2] = {6, 6}, wid_p1[1] = {6};
  if (residuals) residuals->clear();
  if (jacobian) { jacobian->rows.assign(1, 0); jacobian->cols.clear(); jacobian->values.clear(); jacobian->num_cols = num_cols; }
  int num_rows = 0;
  struct Piece { int col, off, wid; };
  for (size_t i = 0; i < blocks.size(); ++i) {
    const int k = w.order_kind[i], idx = w.order_idx[i];
    const Kind kind = blocks[i].g->kind();
    const int rows_here = kind == Kind::R ? 4 : nres[k];            // RError<4,7>: the device batch pads it to 6 rows
    const int* off = nullptr; const int* wid = nullptr; int nb = 0;
    switch (kind) {
      case Kind::TwoCamera: off = off_tc; wid = wid_tc; nb = 1; break;
      case Kind::TwoFrame: off = off_tf; wid = wid_tf; nb = 3; break;
      case Kind::PoseOnly: off = off_po; wid = wid_po; nb = 1; break;
      case Kind::Imu: off = off_imu; wid = wid_imu; nb = 8; break;
      case Kind::PoseGraph: off = off_pg; wid = wid_pg; nb = 2; break;
      default: off = off_p1; wid = wid_p1; nb = 1; break;          // PoseError / RError: the single pose sits in the batch's second slot
    }
    const double* rsrc = r[k].data() + (size_t)idx * nres[k];
    if (residuals) residuals->insert(residuals->end(), rsrc, rsrc + rows_here);
    if (jacobian) {
      Piece pc[8]; int np = 0;
      for (int b = 0; b < nb; ++b) {
        const int c = active_col(blocks[i].params[b]);
        if (c >= 0) pc[np++] = Piece{c, off[b], wid[b]};
      }
      std::sort(pc, pc + np, [](const Piece& a, const Piece& b) { return a.col < b.col; });
      for (int row = 0; row < rows_here; ++row) {
        const double* jrow = J[k].data() + ((size_t)idx * nres[k] + row) * L[k];
        for (int q = 0; q < np; ++q)
          for (int c = 0; c < pc[q].wid; ++c) { jacobian->cols.push_back(pc[q].col + c); jacobian->values.push_back(jrow[pc[q].off + c]); }
        jacobian->rows.push_back((int)jacobian->cols.size());
      }
    }
    num_rows += rows_here;
  }
  if (jacobian) jacobian->num_rows = num_rows;
  if (gradient) {
    std::vector<double> gc((size_t)15 * n_kf), gl((size_t)std::max(n_lm, 1));
    if (lvf_problem_gradient(d.h.prob, &o, gc.data(), n_lm ? gl.data() : nullptr) != LVF_OK) return bail(lvf_last_error());
    gradient->assign(num_cols, 0.0);
    auto put = [&](double* p, const double* src, int n) {
      if (!p) return;
      const int c = active_col(p);
      if (c >= 0) std::copy(src, src + n, gradient->begin() + c);
    };
    for (int kf = 0; kf < n_kf; ++kf) {
      put(w.pose_ptr[kf], &gc[(size_t)6 * kf], 6);
      const double* vb = &gc[(size_t)6 * n_kf + (size_t)9 * kf];
      put(w.v_ptr[kf], vb, 3); put(w.ba_ptr[kf], vb + 3, 3); put(w.bg_ptr[kf], vb + 6, 3);
    }
    for (int l = 0; l < n_lm; ++l) put(w.lm_ptr[l], &gl[l], 1);
  }
  return true;
}

// convenience form kept from round 1: cost + the RAW residual vector (loss applied in the cost only)
inline bool Evaluate(ceres::Problem* problem, double* cost, std::vector<double>* residuals, std::string* error = nullptr) {
  if (cost && !Evaluate(problem, ceres::Problem::EvaluateOptions(), cost, nullptr, nullptr, nullptr, error)) return false;
  if (!residuals) return true;
  ceres::Problem::EvaluateOptions raw;
  raw.apply_loss_function = false;
  return Evaluate(problem, raw, nullptr, residuals, nullptr, nullptr, error);
}

}  // namespace gpu
}  // namespace lvio_fusion
#endif  // LVF_CERES_ADAPTER_HPP_
