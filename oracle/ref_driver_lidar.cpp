// ORACLE — TEST INFRASTRUCTURE ONLY.  Second driver of oracle/_ref/liblvf_ref.so: the REFERENCE's own LiDAR front half, compiled UNMODIFIED from
//   /root/reference/src/lvio_fusion/src/projection.cpp        (ImageProjection: range image, ground marking, BFS segmentation)
//   /root/reference/src/lvio_fusion/src/association.cpp       (FeatureAssociation: Preprocess, AdjustDistortion, CalculateSmoothness,
//                                                              ExtractFeatures' picks, AlignScan, ScanToMapWithGround / WithSegmented)
// as translation units of their own (oracle/Makefile), against the stand-in containers under oracle/ref_shim/ (cv::Mat as a typed array,
// pcl::PointCloud as a vector, PCL filters / RANSAC as pass-throughs, KdTreeFLANN as the declared exact brute-force search, ceres::Problem
// as a recorder).  No reference source is copied: this file only CALLS the reference's member functions — the private ones too, which is
// why the class headers are included with `private` / `protected` opened (class layout is unchanged by that).
// What the reference leaves to files this build does not compile is stubbed HERE and never reached by the pinned calls:
// Frame::Frame() (src/frame.cpp:9-17 sets the default weights — the driver sets the weights it needs), Frame::t(), Map::ComputePose /
// GetKeyFrames (deskewing and the keyframe loop of AddScan).
// Used by tests/test_oracle_ref.py to pin oracle/extract.h and oracle/icp.h's association against the reference's text, live and through
// tests/golden/ref_v3.npz (tests/golden/make_ref_golden_lidar.py).
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
#include "lvio_fusion/lidar/association.h"
#include "lvio_fusion/lidar/lidar.h"
#include "lvio_fusion/map.h"
#undef private
#undef protected

namespace lvio_fusion {
std::vector<Lidar::Ptr> Lidar::devices_;                      // src/lidar/lidar.cpp
unsigned long Frame::current_frame_id = 0;
Frame::Frame() : id(0), time(0) {}                           // (src/frame.cpp is not compiled: see the header of this file)
Vector3d Frame::t() { std::abort(); }
SE3d Map::ComputePose(double) { std::abort(); }
}  // namespace lvio_fusion

using namespace lvio_fusion;

namespace {
void set_lidar(double resolution, const double* extrinsic7) {
  Lidar::devices_.clear();
  static const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
  Lidar::Create(resolution, SE3d(extrinsic7 ? extrinsic7 : ident));
}
void to_cloud(const float* pts, int n, int stride, PointICloud& c, bool with_intensity) {
  c.clear();
  for (int i = 0; i < n; ++i) {
    PointI p;
    p.x = pts[(size_t)i * stride]; p.y = pts[(size_t)i * stride + 1]; p.z = pts[(size_t)i * stride + 2];
    p.intensity = with_intensity ? pts[(size_t)i * stride + 3] : 0.0f;
    c.push_back(p);
  }
}
int from_cloud(const PointICloud& c, float* out) {
  for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
  return (int)c.size();
}
}  // namespace

extern "C" {

struct lvr_lidar_params { int num_scans, horizon_scan, ground_rows; float ang_res_y, ang_bottom, min_range, max_range, resolution; double cycle_time; };

// FeatureAssociation::Process step by step (association.cpp:86-95 -> :97-102, projection.cpp:26-40, association.cpp:104-111) with taps:
// (extrinsic7: see the first line of the body)  every array is caller-allocated for num_scans * horizon_scan entries (clouds: 4 floats per point, capacity n).
// counts6 = {n_filtered, n_segmented, n_ground_picks, n_surf_picks, label_count, 0}; orient3 = {start, end, diff} orientation.
void lvr_lidar_extract(const float* pts, int n, int stride, const lvr_lidar_params* P, float* filtered, float* range_mat, signed char* ground_mat,
                       int* label_mat, float* segmented, unsigned char* seg_ground, int* seg_col, float* seg_range, int* start_ring, int* end_ring,
                       float* curvature, float* ground_picks, float* surf_picks, int* counts6, float* orient3, const double* extrinsic7) {
  // extrinsic7 == NULL: identity — Sensor2Robot (association.cpp:236-247, an inline member: only reachable through ExtractFeatures) then returns
  // the picks unchanged; otherwise ground_picks / surf_picks are the picks taken through the reference's own Sensor2Robot
  set_lidar(P->resolution, extrinsic7);
  FeatureAssociation fa(P->num_scans, P->horizon_scan, P->ang_res_y, P->ang_bottom, P->ground_rows, P->cycle_time, P->min_range, P->max_range, 0.0, 0.0);
  const int npix = P->num_scans * P->horizon_scan;
  std::memset(fa.curvatures, 0, sizeof(float) * (size_t)npix);      // `new float[]` in the reference (association.h:23): declared as a fresh, zeroed allocation
  PointICloud points;
  to_cloud(pts, n, stride, points, false);
  fa.Preprocess(points);
  counts6[0] = from_cloud(points, filtered);
  ImageProjection& ip = *fa.projection_;
  SegmentedInfo info(P->num_scans, P->horizon_scan);
  PointICloud seg;
  ip.FindStartEndAngle(info, points);
  ip.ProjectPointCloud(info, points);
  ip.RemoveGround(info);
  ip.Segment(info, seg);
  for (int i = 0; i < P->num_scans; ++i)
    for (int j = 0; j < P->horizon_scan; ++j) {
      const size_t k = (size_t)i * P->horizon_scan + j;
      range_mat[k] = ip.range_mat.at<float>(i, j); ground_mat[k] = ip.ground_mat.at<int8_t>(i, j); label_mat[k] = ip.label_mat.at<int>(i, j);
    }
  counts6[4] = ip.label_count;
  ip.Clear();
  orient3[0] = info.start_orientation; orient3[1] = info.end_orientation; orient3[2] = info.orientation_diff;
  const int m = (int)seg.size();
  counts6[1] = m;
  for (int k = 0; k < m; ++k) { seg_ground[k] = info.ground_flag[k] ? 1 : 0; seg_col[k] = (int)info.col_ind[k]; seg_range[k] = info.range[k]; }
  for (int i = 0; i < P->num_scans; ++i) { start_ring[i] = info.start_ring_index[i]; end_ring[i] = info.end_ring_index[i]; }
  fa.AdjustDistortion(seg, info);
  fa.CalculateSmoothness(seg, info);
  from_cloud(seg, segmented);
  for (int k = 0; k < m; ++k) curvature[k] = fa.curvatures[k];
  Frame::Ptr frame(new Frame());
  fa.ExtractFeatures(seg, info, frame);       // PCL filters are pass-throughs here: feature_lidar holds the picks of association.cpp:185-208
  counts6[2] = from_cloud(frame->feature_lidar->points_ground, ground_picks);
  counts6[3] = from_cloud(frame->feature_lidar->points_surf, surf_picks);
  counts6[5] = 0;
  delete[] fa.curvatures; fa.curvatures = nullptr;      // (the reference never frees it)
}

// FeatureAssociation::AlignScan (association.cpp:39-64) on two stamped revolutions; returns 1 and the cut (xyz, intensity 0) or 0
int lvr_align_scan(const float* pc1, int n1, double stamp1, const float* pc2, int n2, double stamp2, double cycle_time, double time, float* out, int* n_out) {
  set_lidar(0.2, nullptr);
  FeatureAssociation fa(1, 1, 1.0, 0.0, 0, cycle_time, 0.0, 1.0, 0.0, 0.0);
  auto mk = [](const float* p, int n) { Point3Cloud::Ptr c(new Point3Cloud()); for (int i = 0; i < n; ++i) { Point3 q; q.x = p[4 * i]; q.y = p[4 * i + 1]; q.z = p[4 * i + 2]; c->push_back(q); } return c; };
  fa.raw_point_clouds_[stamp1] = mk(pc1, n1);
  fa.raw_point_clouds_[stamp2] = mk(pc2, n2);
  PointICloud o;
  const bool ok = fa.AlignScan(time, o);
  *n_out = ok ? from_cloud(o, out) : 0;
  return ok ? 1 : 0;
}

// FeatureAssociation::ScanToMapWithGround (mode 0, association.cpp:270-326) / ScanToMapWithSegmented (mode 1, :328-384): the reference builds its
// adapt::Problem; the recorded blocks are read back — LidarError blocks evaluated at para through CostFunction::Evaluate (residual + the three
// 1 x 1 Jacobians, in insertion order = scan order of the accepted points), the loss function's parameter, and the prior block if any.
// residuals / jacobians have room for n_scan blocks.  counts4 = {LidarError blocks, Other blocks, parameter blocks, 0}; prior4 = {r0, r1, r2, weight}.
void lvr_scan_to_map(int mode, const float* scan, int n_scan, const float* map, int n_map, const double* frame_pose, const double* map_pose, double* para6,
                     double w_ground, double w_surf, double w_visual, int n_features_left, int relocate, double resolution, double* residuals,
                     double* jacobians3, double* huber_a, int* counts4, double* prior4) {
  set_lidar(resolution, nullptr);
  FeatureAssociation fa(1, 1, 1.0, 0.0, 0, 0.1, 0.0, 1.0, 0.0, 0.0);
  Frame::Ptr frame(new Frame()), map_frame(new Frame());
  frame->pose = SE3d(frame_pose); map_frame->pose = SE3d(map_pose);
  frame->weights.lidar_ground = w_ground; frame->weights.lidar_surf = w_surf; frame->weights.visual = w_visual;
  frame->feature_lidar = lidar::Feature::Create(); map_frame->feature_lidar = lidar::Feature::Create();
  PointICloud& fs = mode == 0 ? frame->feature_lidar->points_ground : frame->feature_lidar->points_surf;
  PointICloud& ms = mode == 0 ? map_frame->feature_lidar->points_ground : map_frame->feature_lidar->points_surf;
  to_cloud(scan, n_scan, 4, fs, true);
  to_cloud(map, n_map, 4, ms, true);
  for (int k = 0; k < n_features_left; ++k) frame->features_left[(unsigned long)k] = nullptr;      // only its size() is read (association.cpp:323, :381)
  adapt::Problem problem;
  if (mode == 0) fa.ScanToMapWithGround(frame, map_frame, para6, problem, relocate != 0);
  else fa.ScanToMapWithSegmented(frame, map_frame, para6, problem, relocate != 0);
  int n_lidar = 0, n_other = 0;
  *huber_a = 0.0;
  prior4[0] = prior4[1] = prior4[2] = prior4[3] = 0.0;
  for (ceres::ResidualBlockId b : problem.recorded_blocks()) {
    const double* params[3] = {b->params[0], b->params[1], b->params[2]};
    if (problem.types[b] == ProblemType::LidarError) {
      double j0, j1, j2;
      double* jac[3] = {&j0, &j1, &j2};
      b->cost->Evaluate(params, residuals + n_lidar, jac);
      jacobians3[3 * n_lidar] = j0; jacobians3[3 * n_lidar + 1] = j1; jacobians3[3 * n_lidar + 2] = j2;
      if (auto* h = dynamic_cast<ceres::HuberLoss*>(b->loss)) *huber_a = h->a();
      ++n_lidar;
    } else {
      double r3[3];
      b->cost->Evaluate(params, r3, nullptr);
      prior4[0] = r3[0]; prior4[1] = r3[1]; prior4[2] = r3[2];
      ++n_other;
    }
  }
  counts4[0] = n_lidar; counts4[1] = n_other; counts4[2] = (int)problem.recorded_parameter_blocks().size(); counts4[3] = problem.num_types[ProblemType::LidarError];
}

}  // extern "C"
