// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for the PCL header of the same name: the INTERFACE association.cpp / projection.cpp name, with a
// pass-through body — the third-party algorithm itself is not part of the reference's text (its declared semantics live in oracle/cloud.h,
// oracle/knn.h).  With pass-through filters the clouds ExtractFeatures hands to PCL come out unchanged, which is how oracle/ref_driver_lidar.cpp
// reads the reference's own picks (association.cpp:185-208).
#pragma once
#include "../common/io.h"
namespace pcl {
template <typename PointT>
class ExtractIndices {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
  void setIndices(const PointIndices::Ptr& i) { idx_ = i; }
  void setNegative(bool n) { neg_ = n; }
  void filter(PointCloud<PointT>& out) {
    if (!in_) return;
    std::vector<char> sel(in_->points.size(), 0);
    if (idx_) for (int k : idx_->indices) if (k >= 0 && (size_t)k < sel.size()) sel[k] = 1;
    PointCloud<PointT> r; r.header = in_->header;
    for (size_t i = 0; i < sel.size(); ++i) if ((sel[i] != 0) != neg_) r.push_back(in_->points[i]);
    out = r;
  }
 private:
  typename PointCloud<PointT>::ConstPtr in_; PointIndices::Ptr idx_; bool neg_ = false;
};
}  // namespace pcl
