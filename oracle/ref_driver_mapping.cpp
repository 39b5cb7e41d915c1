// ORACLE — TEST INFRASTRUCTURE ONLY.  Fourth driver of oracle/_ref/liblvf_ref.so (round 6): the reference's own CONTROL code
//   /root/reference/src/lvio_fusion/src/mapping.cpp      (Mapping::Optimize :139-191, BuildMapFrame :114-137, BuildOldMapFrame :78-112, MergeScan / ToWorld
//                                                          :193-220, Relocate :251-300 — the whole file is compiled, UNMODIFIED)
//   /root/reference/src/lvio_fusion/src/pose_graph.cpp   (PoseGraph::BuildProblem :163-199, Optimize :201-224, ForwardUpdate :227-252)
//   /root/reference/src/lvio_fusion/src/relocator.cpp    (Relocator::UpdateNewSubmap :247-282)
//   /root/reference/src/lvio_fusion/src/environment.cpp  (Environment::Optimize :18-115 — the third adapt::Solve call site: one free pose, PoseOnly blocks,
//                                                          one ImuError whose other seven parameter blocks are constant)
// as translation units of their own (oracle/Makefile) against the stand-in third-party headers of oracle/ref_shim/.  ceres::Solve is the DECLARED
// Levenberg-Marquardt loop (ref_shim/ceres/solve_shim.h, included HERE and nowhere else) over the recording ceres::Problem, so the reference's
// outer loops run END TO END: four passes x {ground, surf} of Mapping::Relocate with its score arithmetic, the per-keyframe chain of Mapping::Optimize
// with ForwardUpdate and ToWorld in between, the pose-graph solve and its section-wise forward updates, the rotation solve of UpdateNewSubmap.
// No reference source is copied: this file builds the object graph out of flat arrays and CALLS the reference's member functions.
// Stand-ins written here for code this build does not compile (declared, not pinned):
//   Map::GetKeyFrame / GetKeyFrames  (src/map.cpp:22-95: std::map range queries; src/map.cpp needs Eigen slerp / Sophus constructors the stand-in headers lack),
//   se32rpyxyz / rpyxyz2se3          (src/utility.cpp:27-40: the reference's own ceres::SE3ToRpyxyz / RpyxyzToSE3 of ceres/base.hpp:134-150 on SE3d::data()),
//   Frontend (never constructed: zeroed storage carrying `mutex` and `last_frame`, which PoseGraph::ForwardUpdate(transform, time) reads).
// tests/test_oracle_ref.py + tests/golden/make_ref_golden_mapping.py pin oracle/icp.h / oracle/loop.h (through oracle/pyoracle.py's compositions) and
// lvio_fusion_amd's scan-match / pose-graph paths to these results, live here and through tests/golden/ref_v5.npz on the GPU box.
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
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
#ifndef LVF_DROPIN_BUILD
#include <ceres/solve_shim.h>      // the CPU pin: ceres::Solve = the declared LM loop.  (The compiled drop-in defines ceres::Solve as gpu::Solve: ref_driver_dropin.cpp.)
#endif
#include <opencv2/opencv.hpp>
#include <pcl/common/common_headers.h>
#include <pcl/common/io.h>
#include <sophus/se3.hpp>

#define private public
#define protected public
#include "lvio_fusion/adapt/environment.h"
#include "lvio_fusion/visual/camera.h"
#include "lvio_fusion/visual/feature.h"
#include "lvio_fusion/visual/landmark.h"
#include "lvio_fusion/ceres/imu_error.hpp"
#include "lvio_fusion/frontend.h"
#include "lvio_fusion/lidar/association.h"
#include "lvio_fusion/lidar/lidar.h"
#include "lvio_fusion/lidar/mapping.h"
#include "lvio_fusion/loop/pose_graph.h"
#include "lvio_fusion/loop/relocator.h"
#include "lvio_fusion/map.h"
#undef private
#undef protected

namespace lvio_fusion {
// ---- stand-ins (see the header)
Frame::Ptr Map::GetKeyFrame(double time) {
  if (time < 0) return (--keyframes.end())->second;
  auto it = keyframes.lower_bound(time);
  if (it == keyframes.end()) return (--keyframes.end())->second;
  if (it == keyframes.begin()) return it->second;
  auto prev = it; --prev;
  return (time - prev->first > it->first - time) ? it->second : prev->second;      // the nearer of the two neighbours, the earlier one on a tie
}
Frames Map::GetKeyFrames(double start, double end, int num) {
  if (end == 0 && num == 0) { auto a = keyframes.lower_bound(start); return a == keyframes.end() ? Frames() : Frames(a, keyframes.end()); }
  if (num == 0) { return start > end ? Frames() : Frames(keyframes.lower_bound(start), keyframes.upper_bound(end)); }
  Frames out;
  if (end == 0) { auto it = keyframes.upper_bound(start); for (int i = 0; i < num && it != keyframes.end(); ++i) out.insert(*(it++)); }
  else if (start == 0) { auto it = keyframes.lower_bound(end); for (int i = 0; i < num && it != keyframes.begin(); ++i) out.insert(*(--it)); }
  return out;
}
void se32rpyxyz(const SE3d relative_i_j, double* rpyxyz) { ceres::SE3ToRpyxyz<double>(relative_i_j.data(), rpyxyz); }
SE3d rpyxyz2se3(const double* rpyxyz) { double se3[7]; ceres::RpyxyzToSE3<double>(rpyxyz, se3); return SE3d(se3); }
// ---- referenced by a function of pose_graph.cpp this driver never calls (PoseGraph::AddSection); its home, src/utility.cpp, is not compiled
double vectors_degree_angle(Vector3d, Vector3d) { std::abort(); }
// ---- referenced by Environment::Step (never called here); its home, src/frame.cpp, is not compiled
Observation Frame::GetObservation() { std::abort(); }
}  // namespace lvio_fusion

using namespace lvio_fusion;

namespace {
void to_cloud(const float* pts, int n, PointICloud& c) {
  c.clear();
  for (int i = 0; i < n; ++i) { PointI p; p.x = pts[4 * i]; p.y = pts[4 * i + 1]; p.z = pts[4 * i + 2]; p.intensity = pts[4 * i + 3]; c.push_back(p); }
}
void set_lidar(double resolution) {
  Lidar::devices_.clear();
  static const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
  Lidar::Create(resolution, SE3d(ident));
}
// frames with their body-frame feature clouds; n_ground[k] < 0: the frame carries no lidar feature
std::vector<Frame::Ptr> make_frames(int n, const double* time, const double* pose, const int* n_ground, const int* n_surf, const float* ground, const float* surf,
                                    double w_ground, double w_surf, double w_visual, const int* n_features_left) {
  std::vector<Frame::Ptr> frames((size_t)n);
  size_t g0 = 0, s0 = 0;
  for (int k = 0; k < n; ++k) {
    Frame::Ptr f(new Frame());
    f->id = (unsigned long)(k + 1); f->time = time[k]; f->pose = SE3d(pose + 7 * k);
    f->weights.lidar_ground = w_ground; f->weights.lidar_surf = w_surf; f->weights.visual = w_visual;
    if (n_ground[k] >= 0) {
      f->feature_lidar = lidar::Feature::Create();
      to_cloud(ground + 4 * g0, n_ground[k], f->feature_lidar->points_ground); g0 += (size_t)n_ground[k];
      to_cloud(surf + 4 * s0, n_surf[k], f->feature_lidar->points_surf); s0 += (size_t)n_surf[k];
    }
    if (n_features_left) for (int q = 0; q < n_features_left[k]; ++q) f->features_left[(unsigned long)q] = nullptr;      // only size() is read (association.cpp:323, :381)
    frames[(size_t)k] = f;
  }
  return frames;
}
// a Frontend that is never constructed (its constructor builds detectors and matchers): zeroed storage is a valid unlocked std::mutex and a null
// shared_ptr on this platform; PoseGraph::ForwardUpdate(transform, time) locks `mutex`, reads `last_frame` and calls UpdateCache() (a no-op stand-in)
Frontend::Ptr make_frontend(Frame::Ptr last_frame) {
  void* mem = std::calloc(1, sizeof(Frontend));
  Frontend* fe = reinterpret_cast<Frontend*>(mem);
  fe->last_frame = last_frame;
  return Frontend::Ptr(fe, [](Frontend* p) { p->last_frame.reset(); std::free(p); });
}
}  // namespace

extern "C" {

// Mapping::Optimize over frames [first_active, n): frames before first_active only enter the map (Mapping::ToWorld = MergeScan of their clouds at their
// poses).  All frames live in lvio_fusion::Map::Instance().keyframes; PoseGraph::ForwardUpdate moves every later keyframe after each frame's update.
// pose_out [n][7]: every frame's pose afterwards.  world_counts [n][2]: sizes of pointclouds_ground / pointclouds_surf per frame afterwards.
void lvr_mapping_optimize(int n, const double* time, const double* pose, const int* n_ground, const int* n_surf, const float* ground, const float* surf, int first_active,
                          double w_ground, double w_surf, double w_visual, const int* n_features_left, double resolution, double* pose_out, int* world_counts) {
  set_lidar(resolution);
  std::vector<Frame::Ptr> frames = make_frames(n, time, pose, n_ground, n_surf, ground, surf, w_ground, w_surf, w_visual, n_features_left);
  lvio_fusion::Map::Instance().Reset();
  for (auto& f : frames) lvio_fusion::Map::Instance().keyframes[f->time] = f;
  // Frontend::last_frame: the front end's newest frame, LATER than every keyframe the mapper works on (Backend::Optimize hands Mapping::Optimize the keyframes
  // up to end - window_size, backend.cpp:224-226) and not (yet) a keyframe; ForwardUpdate(transform, time) moves it along with the later keyframes
  Frame::Ptr newest(new Frame());
  newest->id = (unsigned long)(n + 1); newest->time = time[n - 1] + 0.25; newest->pose = SE3d(pose + 7 * (n - 1));
  PoseGraph::Instance().SetFrontend(make_frontend(newest));
  Mapping mapping;
  mapping.SetFeatureAssociation(FeatureAssociation::Ptr(new FeatureAssociation(1, 1, 1.0, 0.0, 0, 0.1, 0.0, 1.0, 0.0, 0.0)));
  for (int k = 0; k < first_active; ++k) mapping.ToWorld(frames[(size_t)k]);
  Frames active;
  for (int k = first_active; k < n; ++k) active[frames[(size_t)k]->time] = frames[(size_t)k];
  mapping.Optimize(active);                                     // <- the reference's text
  for (int k = 0; k < n; ++k) {
    std::memcpy(pose_out + 7 * k, frames[(size_t)k]->pose.data(), 7 * sizeof(double));
    world_counts[2 * k] = (int)mapping.pointclouds_ground[frames[(size_t)k]->time].size();
    world_counts[2 * k + 1] = (int)mapping.pointclouds_surf[frames[(size_t)k]->time].size();
  }
  PoseGraph::Instance().SetFrontend(nullptr);
  lvio_fusion::Map::Instance().Reset();
}

// Mapping::Relocate(last_frame = frames[old_index], current_frame, relative_o_c): frames are the OLD keyframes around the loop's old frame (all with lidar
// features; their world clouds come from Mapping::ToWorld), `cur_*` the current frame with loop_closure->relative_o_c = rel_in.  Returns the int score.
int lvr_mapping_relocate(int n, const double* time, const double* pose, const int* n_ground, const int* n_surf, const float* ground, const float* surf, int old_index,
                         const float* cur_ground, int cur_n_ground, const float* cur_surf, int cur_n_surf, const double* cur_pose, const double* rel_in,
                         double w_ground, double w_surf, double w_visual, double resolution, double* rel_out, double* map_pose_out, int* map_counts) {
  set_lidar(resolution);
  std::vector<Frame::Ptr> frames = make_frames(n, time, pose, n_ground, n_surf, ground, surf, w_ground, w_surf, w_visual, nullptr);
  lvio_fusion::Map::Instance().Reset();
  for (auto& f : frames) lvio_fusion::Map::Instance().keyframes[f->time] = f;
  Mapping mapping;
  mapping.SetFeatureAssociation(FeatureAssociation::Ptr(new FeatureAssociation(1, 1, 1.0, 0.0, 0, 0.1, 0.0, 1.0, 0.0, 0.0)));
  for (auto& f : frames) mapping.ToWorld(f);
  Frame::Ptr cur(new Frame());
  cur->id = (unsigned long)(n + 1); cur->time = time[n - 1] + 100.0; cur->pose = SE3d(cur_pose);
  cur->weights.lidar_ground = w_ground; cur->weights.lidar_surf = w_surf; cur->weights.visual = w_visual;
  cur->feature_lidar = lidar::Feature::Create();
  to_cloud(cur_ground, cur_n_ground, cur->feature_lidar->points_ground);
  to_cloud(cur_surf, cur_n_surf, cur->feature_lidar->points_surf);
  cur->loop_closure = loop::LoopClosure::Ptr(new loop::LoopClosure());
  cur->loop_closure->frame_old = frames[(size_t)old_index];
  cur->loop_closure->relative_o_c = SE3d(rel_in);
  SE3d rel;
  const int score = mapping.Relocate(frames[(size_t)old_index], cur, rel);      // <- the reference's text
  std::memcpy(rel_out, rel.data(), 7 * sizeof(double));
  if (map_pose_out) {           // what BuildOldMapFrame made (re-run: it has no side effects)
    Frame::Ptr mf(new Frame());
    mapping.BuildOldMapFrame(frames[(size_t)old_index], mf);
    std::memcpy(map_pose_out, mf->pose.data(), 7 * sizeof(double));
    map_counts[0] = (int)mf->feature_lidar->points_ground.size(); map_counts[1] = (int)mf->feature_lidar->points_surf.size();
  }
  cur->loop_closure.reset();
  lvio_fusion::Map::Instance().Reset();
  return score;
}

// PoseGraph::BuildProblem + Optimize.  Keyframes (time, pose) fill the map; sections: the A-times of the turning sections between the loop's old frame
// (submap_A) and the new sub-map's start frame (submap_B).  pose_out [n][7].  counts3 = {residual blocks, parameter blocks (calls), num_frames}.
// start_after: the pose Relocator::UpdateNewSubmap gives the sub-map's start frame BETWEEN BuildProblem and Optimize (relocator.cpp:214-216; may be null).
void lvr_pose_graph_optimize(int n, const double* time, const double* pose, const double* vw, int n_sections, const double* section_A, double submap_A, double submap_B,
                             const double* start_after, double* pose_out, double* vw_out, int* counts3, double* summary3) {
  lvio_fusion::Map::Instance().Reset();
  std::vector<Frame::Ptr> frames((size_t)n);
  for (int k = 0; k < n; ++k) {
    Frame::Ptr f(new Frame());
    f->id = (unsigned long)(k + 1); f->time = time[k]; f->pose = SE3d(pose + 7 * k);
    f->Vw = Vector3d(vw[3 * k], vw[3 * k + 1], vw[3 * k + 2]);
    frames[(size_t)k] = f; lvio_fusion::Map::Instance().keyframes[f->time] = f;
  }
  Atlas sections;
  for (int i = 0; i < n_sections; ++i) { Section s; s.A = section_A[i]; s.B = section_A[i]; s.C = section_A[i]; s.degree = 0; sections[s.A] = s; }
  Section submap; submap.A = submap_A; submap.B = submap_B; submap.C = time[n - 1]; submap.degree = 0;
  {
    adapt::Problem problem;
    PoseGraph::Instance().BuildProblem(sections, submap, problem);      // <- the reference's text
#ifdef LVF_DROPIN_BUILD
    counts3[0] = problem.NumResidualBlocks(); counts3[1] = problem.NumParameterBlocks(); counts3[2] = problem.num_frames;      // (distinct blocks: old, start, sections)
#else
    counts3[0] = problem.NumResidualBlocks(); counts3[1] = (int)problem.recorded_parameter_blocks().size(); counts3[2] = problem.num_frames;
#endif
    if (start_after) lvio_fusion::Map::Instance().GetKeyFrame(submap_B)->pose = SE3d(start_after);
    PoseGraph::Instance().Optimize(sections, submap, problem);          // <- the reference's text (ceres::Solve = solve_shim.h)
  }
  summary3[0] = summary3[1] = summary3[2] = 0.0;
  for (int k = 0; k < n; ++k) {
    std::memcpy(pose_out + 7 * k, frames[(size_t)k]->pose.data(), 7 * sizeof(double));
    for (int i = 0; i < 3; ++i) vw_out[3 * k + i] = frames[(size_t)k]->Vw[i];
  }
  lvio_fusion::Map::Instance().Reset();
}

// Relocator::UpdateNewSubmap(best_frame, new_submap_kfs): n frames of the new sub-map, each with loop_closure {frame_old (pose old_pose[k]), relative_o_c[k]}.
void lvr_update_new_submap(int n, const double* time, const double* pose, const double* old_pose, const double* relative_o_c, int best, double* pose_out) {
  std::vector<Frame::Ptr> frames((size_t)n), olds((size_t)n);
  Frames kfs;
  for (int k = 0; k < n; ++k) {
    Frame::Ptr f(new Frame()), o(new Frame());
    f->id = (unsigned long)(k + 1); f->time = time[k]; f->pose = SE3d(pose + 7 * k);
    o->id = (unsigned long)(1000 + k); o->time = time[k] - 1000.0; o->pose = SE3d(old_pose + 7 * k);
    f->loop_closure = loop::LoopClosure::Ptr(new loop::LoopClosure());
    f->loop_closure->frame_old = o; f->loop_closure->relative_o_c = SE3d(relative_o_c + 7 * k);
    frames[(size_t)k] = f; olds[(size_t)k] = o; kfs[f->time] = f;
  }
  alignas(Relocator) static unsigned char storage[sizeof(Relocator)];      // (never constructed: the constructor starts the detector thread; UpdateNewSubmap reads no member)
  std::memset(storage, 0, sizeof(storage));
  reinterpret_cast<Relocator*>(storage)->UpdateNewSubmap(frames[(size_t)best], kfs);      // <- the reference's text
  for (int k = 0; k < n; ++k) std::memcpy(pose_out + 7 * k, frames[(size_t)k]->pose.data(), 7 * sizeof(double));
  for (auto& f : frames) f->loop_closure.reset();
}

// Environment::Optimize (environment.cpp:18-115) on ONE keyframe `cur` with its predecessor `last`: PoseOnlyReprojectionError blocks for every feature of
// `cur` (world point = Landmark::ToWorld of a landmark born in frame `birth`), one ImuError (last -> cur) with every block but cur's pose constant, HuberLoss(1),
// ceres defaults (50 iterations), DENSE_QR; estimator_->mapping is null (the lidar half computes a local rpyxyz it never applies: environment.cpp:78-110).
// frames: [birth, last, cur] poses / vel / ba / bg; samples [ns][7] of the (last -> cur) pre-integration.  pose_out[7] = the returned pose.
void lvr_environment_optimize(const double* cam0_11, const double* cam1_11, double baseline, const double* pose3x7, const double* vel3x3, const double* ba3x3,
                              const double* bg3x3, double w_visual, int ns, const double* samples, const double* acc0, const double* gyr0, const double* noise4,
                              int n_lm, const double* inv_depth, const double* right_ob, const double* left_ob, double* pose_out) {
  Camera::devices_.clear();
  Camera::Create(cam0_11[0], cam0_11[1], cam0_11[2], cam0_11[3], SE3d(cam0_11 + 4));
  Camera::Create(cam1_11[0], cam1_11[1], cam1_11[2], cam1_11[3], SE3d(cam1_11 + 4));
  Camera::baseline = baseline;
  Imu::devices_.clear();
  Imu::Create(SE3d(), 0, 0, 0, 0, 9.81007);
  { Imu::Ptr d = Imu::Get(); d->ACC_N = noise4[0]; d->GYR_N = noise4[1]; d->ACC_W = noise4[2]; d->GYR_W = noise4[3]; d->initialized = true; }
  Frame::Ptr fr[3];
  for (int k = 0; k < 3; ++k) {
    fr[k] = Frame::Ptr(new Frame());
    fr[k]->id = (unsigned long)(k + 1); fr[k]->time = 30.0 + 0.5 * k; fr[k]->pose = SE3d(pose3x7 + 7 * k);
    fr[k]->weights.visual = w_visual; fr[k]->good_imu = true;
    fr[k]->Vw = Vector3d(vel3x3[3 * k], vel3x3[3 * k + 1], vel3x3[3 * k + 2]);
    fr[k]->bias = Bias(Vector3d(ba3x3[3 * k], ba3x3[3 * k + 1], ba3x3[3 * k + 2]), Vector3d(bg3x3[3 * k], bg3x3[3 * k + 1], bg3x3[3 * k + 2]));
    if (k > 0) fr[k]->last_keyframe = fr[k - 1];
  }
  fr[2]->preintegration = imu::Preintegration::Create(fr[1]->bias);
  {
    const Vector3d a0(acc0[0], acc0[1], acc0[2]), g0(gyr0[0], gyr0[1], gyr0[2]);
    for (int s_ = 0; s_ < ns; ++s_) { const double* q = samples + 7 * s_; fr[2]->preintegration->Append(q[0], Vector3d(q[1], q[2], q[3]), Vector3d(q[4], q[5], q[6]), a0, g0); }
  }
  std::vector<visual::Landmark::Ptr> lms((size_t)n_lm);
  for (int l = 0; l < n_lm; ++l) {
    visual::Landmark::Ptr L = visual::Landmark::Create(inv_depth[l]);
    L->id = (unsigned long)(7000 + l);
    cv::KeyPoint kr(cv::Point2f((float)right_ob[2 * l], (float)right_ob[2 * l + 1]), 1.0f);
    visual::Feature::Ptr right = visual::Feature::Create(fr[0], kr, L);
    right->is_on_left_image = false;
    L->first_observation = right;
    cv::KeyPoint kl(cv::Point2f((float)left_ob[2 * l], (float)left_ob[2 * l + 1]), 1.0f);
    visual::Feature::Ptr ft = visual::Feature::Create(fr[2], kl, L);
    fr[2]->features_left[L->id] = ft; L->observations[fr[2]->id] = ft;
    lms[(size_t)l] = L;
  }
  lvio_fusion::Map::Instance().Reset();
  for (int k = 0; k < 3; ++k) lvio_fusion::Map::Instance().keyframes[fr[k]->time] = fr[k];
  // Estimator: never constructed (its constructor reads a config file and builds the whole system); zeroed storage = null `mapping`
  void* emem = std::calloc(1, sizeof(Estimator));
  Environment::estimator_ = Estimator::Ptr(reinterpret_cast<Estimator*>(emem), [](Estimator* p) { std::free(p); });
  Environment::num_frames_per_env_ = 10;
  Environment* env = new Environment();                 // frames_ = the keyframes after a time in [0, 1): all three
  env->state_ = env->frames_.find(fr[2]->time);
  const SE3d result = env->Optimize();                  // <- the reference's text
  std::memcpy(pose_out, result.data(), 7 * sizeof(double));
  delete env;
  Environment::estimator_.reset();
  for (auto& L : lms) { L->observations.clear(); L->first_observation.reset(); }
  for (int k = 0; k < 3; ++k) { fr[k]->features_left.clear(); fr[k]->last_keyframe.reset(); }
  lvio_fusion::Map::Instance().Reset();
}

#ifndef LVF_DROPIN_BUILD
// ceres::Solve of the stand-in on a scan-to-map problem built by the reference's text (mode 0 / 1), for a direct comparison with oracle/icp.h's loop
void lvr_scan_to_map_solve(int mode, const float* scan, int n_scan, const float* map, int n_map, const double* frame_pose, const double* map_pose, double* para6,
                           double w_ground, double w_surf, double w_visual, int n_features_left, int relocate, double resolution, int max_num_iterations, double* summary6) {
  set_lidar(resolution);
  FeatureAssociation fa(1, 1, 1.0, 0.0, 0, 0.1, 0.0, 1.0, 0.0, 0.0);
  Frame::Ptr frame(new Frame()), map_frame(new Frame());
  frame->pose = SE3d(frame_pose); map_frame->pose = SE3d(map_pose);
  frame->weights.lidar_ground = w_ground; frame->weights.lidar_surf = w_surf; frame->weights.visual = w_visual;
  frame->feature_lidar = lidar::Feature::Create(); map_frame->feature_lidar = lidar::Feature::Create();
  to_cloud(scan, n_scan, mode == 0 ? frame->feature_lidar->points_ground : frame->feature_lidar->points_surf);
  to_cloud(map, n_map, mode == 0 ? map_frame->feature_lidar->points_ground : map_frame->feature_lidar->points_surf);
  for (int k = 0; k < n_features_left; ++k) frame->features_left[(unsigned long)k] = nullptr;
  adapt::Problem problem;
  if (mode == 0) fa.ScanToMapWithGround(frame, map_frame, para6, problem, relocate != 0);
  else fa.ScanToMapWithSegmented(frame, map_frame, para6, problem, relocate != 0);
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::DENSE_QR; options.max_num_iterations = max_num_iterations; options.num_threads = 1;
  ceres::Solver::Summary summary;
  adapt::Solve(options, &problem, &summary);
  summary6[0] = summary.initial_cost; summary6[1] = summary.final_cost; summary6[2] = summary.num_residual_blocks_reduced; summary6[3] = summary.num_iterations;
  summary6[4] = summary.num_successful_steps; summary6[5] = (double)summary.termination_type;
}

#endif

}  // extern "C"
