// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for the PCL header of the same name: the INTERFACE association.cpp / projection.cpp name, with a
// pass-through body — the third-party algorithm itself is not part of the reference's text (its declared semantics live in oracle/cloud.h,
// oracle/knn.h).  With pass-through filters the clouds ExtractFeatures hands to PCL come out unchanged, which is how oracle/ref_driver_lidar.cpp
// reads the reference's own picks (association.cpp:185-208).
#pragma once
#include "../common/io.h"
#include "../sample_consensus/method_types.h"
#include "../sample_consensus/model_types.h"
namespace pcl {
// every point is an inlier of a z = 0 plane (pass-through: see above)
template <typename PointT>
class SACSegmentation {
 public:
  void setOptimizeCoefficients(bool) {}
  void setModelType(int) {}
  void setMethodType(int) {}
  void setDistanceThreshold(double) {}
  void setMaxIterations(int) {}
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
  void segment(PointIndices& inliers, ModelCoefficients& co) {
    inliers.indices.clear();
    if (in_) for (size_t i = 0; i < in_->points.size(); ++i) inliers.indices.push_back((int)i);
    co.values.assign({0.f, 0.f, 1.f, 0.f});
  }
 private:
  typename PointCloud<PointT>::ConstPtr in_;
};
}  // namespace pcl
