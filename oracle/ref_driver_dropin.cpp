// TEST INFRASTRUCTURE ONLY — the COMPILED DROP-IN: the reference's own
//   /root/reference/src/lvio_fusion/src/backend.cpp        (Backend::BuildProblem :96-183; the whole file is compiled, UNMODIFIED)
//   /root/reference/src/lvio_fusion/src/association.cpp    (FeatureAssociation::ScanToMapWithGround / WithSegmented :270-384, UNMODIFIED)
//   /root/reference/src/lvio_fusion/src/landmark.cpp, src/preintegration.cpp, src/projection.cpp (association.cpp's ImageProjection member)
// as translation units of their own (oracle/Makefile, target `dropin`), with include/reference_patch/ AHEAD of the reference's include
// directory: X::Create returns the MI355X library's tagged cost functions, adapt::Problem records its payload, adapt::Solve is
// lvio_fusion::gpu::Solve (include/lvf_ceres_adapter.hpp -> the C-ABI of include/lvf.h -> liblvf_hip.so).  Third-party headers are the
// stand-ins of oracle/ref_shim (Eigen / Sophus / PCL / OpenCV) and oracle/ref_shim_gpu (<ceres/ceres.h> = include/lvf_ceres_compat.h).
// So `Backend::BuildProblem -> adapt::Solve -> gpu::Solve` runs on the GPU FROM THE REFERENCE'S TEXT; this file only builds the object
// graph BuildProblem walks out of flat arrays (as oracle/ref_driver_backend.cpp does for the CPU pin), calls the reference's functions,
// sets the solver options Backend::Optimize sets (:206-209) and reads the caller-owned parameter arrays back.  tests/test_gpu_dropin.py
// compares the results with lvf_window_solve and with the oracle.  Nothing under lvio_fusion_amd/ links this library.
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <ceres/ceres.h>
#include <opencv2/opencv.hpp>
#include <pcl/common/common_headers.h>
#include <pcl/common/io.h>
#include <sophus/se3.hpp>

#define private public
#define protected public
#include "lvio_fusion/backend.h"
#include "lvio_fusion/ceres/imu_error.hpp"
#include "lvio_fusion/ceres/lidar_error.hpp"
#include "lvio_fusion/ceres/pose_error.hpp"
#include "lvio_fusion/ceres/visual_error.hpp"
#include "lvio_fusion/frontend.h"
#include "lvio_fusion/imu/tools.h"
#include "lvio_fusion/lidar/association.h"
#include "lvio_fusion/lidar/lidar.h"
#include "lvio_fusion/loop/pose_graph.h"
#include "lvio_fusion/manager.h"
#include "lvio_fusion/map.h"
#include "lvio_fusion/visual/feature.h"
#include "lvio_fusion/visual/landmark.h"
#undef private
#undef protected

// the reference calls ceres::Solve directly at mapping.cpp:277,291, pose_graph.cpp:206 and relocator.cpp:270 (all compiled into this library,
// driven by ref_driver_mapping.cpp with -DLVF_DROPIN_BUILD): it is the GPU here too
namespace ceres {
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) { lvio_fusion::gpu::Solve(options, problem, summary, nullptr); }
}  // namespace ceres

// statics the reference defines in .cpp files this build does not compile (src/visual/camera.cpp, src/imu/imu.cpp, src/lidar/lidar.cpp, src/estimator.cpp)
namespace lvio_fusion {
std::vector<Camera::Ptr> Camera::devices_;
double Camera::baseline = 1;
std::vector<Imu::Ptr> Imu::devices_;
std::vector<Lidar::Ptr> Lidar::devices_;
unsigned long Frame::current_frame_id = 0;
Frame::Frame() : id(0), time(0) {}                                                        // src/frame.cpp:9-17 (the driver sets the weights it needs)
Vector3d Frame::t() { std::abort(); }
// ---- referenced by functions of backend.cpp / association.cpp that the driver never calls; their homes are not compiled here
void Frame::RemoveFeature(visual::Feature::Ptr) { std::abort(); }                         // src/frame.cpp
void Frontend::UpdateCache() {}                                                           // src/frontend.cpp (not on the path; PoseGraph::ForwardUpdate calls it)
void Frontend::UpdateImu(const Bias&) { std::abort(); }
void Initializer::Initialize(double, double) { std::abort(); }                            // src/initializer.cpp
SE3d Map::ComputePose(double) { std::abort(); }
std::vector<Navsat::Ptr> Navsat::devices_;                                                // src/navsat.cpp
void Navsat::Optimize(const Section&) { std::abort(); }
void Navsat::QuickFix(double, double) { std::abort(); }
namespace imu {
void RePredictVel(Frames&, Frame::Ptr&) { std::abort(); }                                 // src/tools.cpp
void RecoverBias(Frames&) { std::abort(); }
}  // namespace imu
}  // namespace lvio_fusion
const double epsilon = 1e-3;
const int num_threads = 1;

using namespace lvio_fusion;

extern "C" {

struct lvd_camera { double fx, fy, cx, cy; double extrinsic[7]; };
// The window as flat arrays (the layout of oracle/ref_driver_backend.cpp's lvr_bp_input, plus the IMU state and samples).  Frames in ascending
// time; [first_active, n_frames) are the active keyframes handed to BuildProblem.
struct lvd_input {
  int n_frames, first_active;
  const double* time;            // [n_frames]
  const double* pose;            // [n_frames][7]  qx qy qz qw tx ty tz
  const double* w_visual;        // [n_frames]
  const unsigned char* good_imu; // [n_frames]
  int imu_initialized;
  const double* vel;             // [n_frames][3]  frame->Vw
  const double* ba;              // [n_frames][3]  frame->bias.linearized_ba
  const double* bg;              // [n_frames][3]
  const int* imu_ns;             // [n_frames] samples of frame k's pre-integration (frame k-1 -> k); 0 for frame 0
  const double* imu_samples;     // concatenated [sum ns][7] = dt, acc, gyr
  const double* imu_acc0;        // [n_frames][3]
  const double* imu_gyr0;        // [n_frames][3]
  const double* pre_ba;          // [n_frames][3]  biases frame k's pre-integration is linearised at (Preintegration::Create(bias))
  const double* pre_bg;          // [n_frames][3]
  const double* imu_noise4;      // ACC_N, GYR_N, ACC_W, GYR_W
  int n_lm;
  const long long* lm_id;        // [n_lm]
  const int* lm_birth;           // [n_lm]
  const double* lm_inv_depth;    // [n_lm]
  const double* lm_right_ob;     // [n_lm][2]
  int n_obs;
  const int* obs_lm;             // [n_obs]
  const int* obs_frame;          // [n_obs]
  const double* obs_xy;          // [n_obs][2]
};

const char* lvd_sources(void) {
  return "src/backend.cpp src/association.cpp src/projection.cpp src/landmark.cpp src/preintegration.cpp (unmodified, from /root/reference) + include/reference_patch + include/lvf_ceres_adapter.hpp";
}

// Backend::BuildProblem (the reference's text) -> adapt::Solve (reference_patch: gpu::Solve) with Backend::Optimize's options (:206-209;
// max_num_iterations from the caller, no time cap: parity needs a deterministic iteration count).  The results are read from where the reference
// keeps them: frame->pose, &landmark->inv_depth, frame->Vw, frame->bias.linearized_{ba,bg}.
// summary8 = {initial_cost, final_cost, successful steps, unsuccessful steps, residual blocks, termination_type, adapt::Problem::num_frames, recorder usable}
// times_ms4 (may be null) = {object graph built by this driver, Backend::BuildProblem, adapt::Solve, ~Problem + read-back}
int lvd_backend_solve(const lvd_camera* c0, const lvd_camera* c1, double baseline, const lvd_input* in, int max_num_iterations, double* pose_out,
                      double* inv_depth_out, double* vel_out, double* ba_out, double* bg_out, double* summary8, char* message, int message_cap, double* times_ms4) {
  const auto tick0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  Camera::devices_.clear();
  Camera::Create(c0->fx, c0->fy, c0->cx, c0->cy, SE3d(c0->extrinsic));
  Camera::Create(c1->fx, c1->fy, c1->cx, c1->cy, SE3d(c1->extrinsic));
  Camera::baseline = baseline;
  Imu::devices_.clear();
  if (in->imu_initialized) {
    Imu::Create(SE3d(), 0, 0, 0, 0, 9.81007);
    Imu::Ptr d = Imu::Get();                 // (Create's argument order is acc_n, acc_w, gyr_n, gyr_w — imu.h:49: set by name)
    d->ACC_N = in->imu_noise4[0]; d->GYR_N = in->imu_noise4[1]; d->ACC_W = in->imu_noise4[2]; d->GYR_W = in->imu_noise4[3];
    d->initialized = true;
  }
  std::vector<Frame::Ptr> frames((size_t)in->n_frames);
  size_t sample0 = 0;
  for (int k = 0; k < in->n_frames; ++k) {
    Frame::Ptr f(new Frame());
    f->id = (unsigned long)(k + 1); f->time = in->time[k];
    f->pose = SE3d(in->pose + 7 * k);
    f->weights.visual = in->w_visual[k];
    f->good_imu = in->good_imu[k] != 0;
    if (in->imu_initialized) {
      f->Vw = Vector3d(in->vel[3 * k], in->vel[3 * k + 1], in->vel[3 * k + 2]);
      f->bias = Bias(Vector3d(in->ba[3 * k], in->ba[3 * k + 1], in->ba[3 * k + 2]), Vector3d(in->bg[3 * k], in->bg[3 * k + 1], in->bg[3 * k + 2]));
      f->preintegration = imu::Preintegration::Create(Bias(Vector3d(in->pre_ba[3 * k], in->pre_ba[3 * k + 1], in->pre_ba[3 * k + 2]), Vector3d(in->pre_bg[3 * k], in->pre_bg[3 * k + 1], in->pre_bg[3 * k + 2])));
      const Vector3d a0(in->imu_acc0[3 * k], in->imu_acc0[3 * k + 1], in->imu_acc0[3 * k + 2]), g0(in->imu_gyr0[3 * k], in->imu_gyr0[3 * k + 1], in->imu_gyr0[3 * k + 2]);
      for (int s = 0; s < in->imu_ns[k]; ++s) {
        const double* q = in->imu_samples + 7 * (sample0 + s);
        f->preintegration->Append(q[0], Vector3d(q[1], q[2], q[3]), Vector3d(q[4], q[5], q[6]), a0, g0);
      }
      sample0 += (size_t)in->imu_ns[k];
    }
    if (k > 0) f->last_keyframe = frames[k - 1];
    frames[k] = f;
  }
  std::vector<visual::Landmark::Ptr> lms((size_t)in->n_lm);
  for (int l = 0; l < in->n_lm; ++l) {
    visual::Landmark::Ptr L = visual::Landmark::Create(in->lm_inv_depth[l]);
    L->id = (unsigned long)in->lm_id[l];
    cv::KeyPoint kp(cv::Point2f((float)in->lm_right_ob[2 * l], (float)in->lm_right_ob[2 * l + 1]), 1.0f);
    visual::Feature::Ptr right = visual::Feature::Create(frames[in->lm_birth[l]], kp, L);
    right->is_on_left_image = false;
    L->first_observation = right;
    lms[l] = L;
  }
  for (int i = 0; i < in->n_obs; ++i) {
    const int l = in->obs_lm[i], k = in->obs_frame[i];
    cv::KeyPoint kp(cv::Point2f((float)in->obs_xy[2 * i], (float)in->obs_xy[2 * i + 1]), 1.0f);
    visual::Feature::Ptr ft = visual::Feature::Create(frames[k], kp, lms[l]);
    frames[k]->features_left[lms[l]->id] = ft;          // Frame::AddFeature (src/frame.cpp:30-35)
    lms[l]->observations[frames[k]->id] = ft;           // Landmark::AddObservation (src/landmark.cpp:66-71)
  }
  Frames active;
  for (int k = in->first_active; k < in->n_frames; ++k) active[frames[k]->time] = frames[k];
  alignas(Backend) static unsigned char storage[sizeof(Backend)];      // (a Backend is never constructed: its constructor starts two threads; BuildProblem reads global_end_)
  std::memset(storage, 0, sizeof(storage));
  Backend* be = reinterpret_cast<Backend*>(storage);
  int rc = 0;
  const double t_graph = ms_since(tick0);
  double t_build = 0.0, t_solve = 0.0;
  const auto tick1 = std::chrono::steady_clock::now();
  {
    adapt::Problem problem;
    be->BuildProblem(active, problem);                  // <- the reference's text, adding gpu:: cost functions through the patched factories
    t_build = ms_since(tick1);
    const auto tick2 = std::chrono::steady_clock::now();
    ceres::Solver::Options options;                     // Backend::Optimize, backend.cpp:205-211
    options.linear_solver_type = ceres::SPARSE_SCHUR;
    options.num_threads = num_threads;
    options.max_num_iterations = max_num_iterations;
    ceres::Solver::Summary summary;
    const bool recorded = problem.recorder.usable(&problem);
    adapt::Solve(options, &problem, &summary);          // <- reference_patch/lvio_fusion/adapt/problem.h: gpu::Solve on the MI355X
    t_solve = ms_since(tick2);
    summary8[0] = summary.initial_cost; summary8[1] = summary.final_cost; summary8[2] = summary.num_successful_steps; summary8[3] = summary.num_unsuccessful_steps;
    summary8[4] = summary.num_residual_blocks; summary8[5] = (double)summary.termination_type; summary8[6] = problem.num_frames; summary8[7] = recorded ? 1.0 : 0.0;
    if (message && message_cap > 0) { std::strncpy(message, summary.message.c_str(), (size_t)message_cap - 1); message[message_cap - 1] = 0; }
    if (summary.termination_type == ceres::FAILURE) rc = 1;
  }
  for (int k = 0; k < in->n_frames; ++k) {
    std::memcpy(pose_out + 7 * k, frames[k]->pose.data(), 7 * sizeof(double));
    if (in->imu_initialized)
      for (int i = 0; i < 3; ++i) { vel_out[3 * k + i] = frames[k]->Vw[i]; ba_out[3 * k + i] = frames[k]->bias.linearized_ba[i]; bg_out[3 * k + i] = frames[k]->bias.linearized_bg[i]; }
  }
  for (int l = 0; l < in->n_lm; ++l) inv_depth_out[l] = lms[l]->inv_depth;
  if (times_ms4) { times_ms4[0] = t_graph; times_ms4[1] = t_build; times_ms4[2] = t_solve; times_ms4[3] = ms_since(tick1) - t_build - t_solve; }
  for (auto& f : frames) { f->features_left.clear(); f->last_keyframe.reset(); }
  for (auto& L : lms) { L->observations.clear(); L->first_observation.reset(); }
  return rc;
}

static void to_cloud(const float* pts, int n, int stride, PointICloud& c) {
  c.clear();
  for (int i = 0; i < n; ++i) {
    PointI p;
    p.x = pts[(size_t)i * stride]; p.y = pts[(size_t)i * stride + 1]; p.z = pts[(size_t)i * stride + 2]; p.intensity = pts[(size_t)i * stride + 3];
    c.push_back(p);
  }
}

// FeatureAssociation::ScanToMapWithGround (mode 0) / WithSegmented (mode 1) — the reference's text: its kd-tree loop over the stand-in exact search,
// LidarPlaneErrorRPZ/YXY::Create and PoseErrorRPZ/YXY::Create through the patched factories — then the solve of Mapping::Optimize (mapping.cpp:158-164:
// DENSE_QR, max_num_iterations 4) through adapt::Solve = gpu::Solve.  para6 (rpyxyz, the caller's LIVE array) is updated in place.
// summary4 = {final_cost, num_residual_blocks_reduced, termination_type, LidarError blocks}
int lvd_scan_to_map_solve(int mode, const float* scan, int n_scan, const float* map, int n_map, const double* frame_pose, const double* map_pose, double* para6,
                          double w_ground, double w_surf, double w_visual, int n_features_left, int relocate, double resolution, int max_num_iterations,
                          double* summary4, char* message, int message_cap) {
  Lidar::devices_.clear();
  static const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
  Lidar::Create(resolution, SE3d(ident));
  FeatureAssociation fa(1, 1, 1.0, 0.0, 0, 0.1, 0.0, 1.0, 0.0, 0.0);
  Frame::Ptr frame(new Frame()), map_frame(new Frame());
  frame->pose = SE3d(frame_pose); map_frame->pose = SE3d(map_pose);
  frame->weights.lidar_ground = w_ground; frame->weights.lidar_surf = w_surf; frame->weights.visual = w_visual;
  frame->feature_lidar = lidar::Feature::Create(); map_frame->feature_lidar = lidar::Feature::Create();
  to_cloud(scan, n_scan, 4, mode == 0 ? frame->feature_lidar->points_ground : frame->feature_lidar->points_surf);
  to_cloud(map, n_map, 4, mode == 0 ? map_frame->feature_lidar->points_ground : map_frame->feature_lidar->points_surf);
  for (int k = 0; k < n_features_left; ++k) frame->features_left[(unsigned long)k] = nullptr;      // only its size() is read (association.cpp:323, :381)
  adapt::Problem problem;
  if (mode == 0) fa.ScanToMapWithGround(frame, map_frame, para6, problem, relocate != 0);
  else fa.ScanToMapWithSegmented(frame, map_frame, para6, problem, relocate != 0);
  ceres::Solver::Options options;                       // mapping.cpp:158-161
  options.linear_solver_type = ceres::DENSE_QR;
  options.max_num_iterations = max_num_iterations;
  options.num_threads = num_threads;
  ceres::Solver::Summary summary;
  adapt::Solve(options, &problem, &summary);
  summary4[0] = summary.final_cost; summary4[1] = summary.num_residual_blocks_reduced; summary4[2] = (double)summary.termination_type;
  summary4[3] = problem.num_types[ProblemType::LidarError];
  if (message && message_cap > 0) { std::strncpy(message, summary.message.c_str(), (size_t)message_cap - 1); message[message_cap - 1] = 0; }
  return summary.termination_type == ceres::FAILURE ? 1 : 0;
}

}  // extern "C"
