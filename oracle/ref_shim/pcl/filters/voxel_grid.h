// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for the PCL header of the same name: the INTERFACE association.cpp / projection.cpp name, with a
// pass-through body — the third-party algorithm itself is not part of the reference's text (its declared semantics live in oracle/cloud.h,
// oracle/knn.h).  With pass-through filters the clouds ExtractFeatures hands to PCL come out unchanged, which is how oracle/ref_driver_lidar.cpp
// reads the reference's own picks (association.cpp:185-208).
#pragma once
#include "../common/io.h"
namespace pcl {
template <typename PointT>
class VoxelGrid {
 public:
  void setLeafSize(float x, float y, float z) { leaf_[0] = x; leaf_[1] = y; leaf_[2] = z; }
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
  void filter(PointCloud<PointT>& out) { if (in_) out = *in_; }
  float leaf_[3] = {0, 0, 0};
 private:
  typename PointCloud<PointT>::ConstPtr in_;
};
}  // namespace pcl
