// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/kdtree/kdtree_flann.h>: nearestKSearch with the DECLARED FLANN semantics (SURVEY.md 8c
// item 6: flann::L2_Simple<float> over (x, y, z), float accumulation in x, y, z order, exact, ascending d2; ties by ascending index) as a
// brute-force scan — so that the reference's own association loops (association.cpp:278-301, :336-359) can run as written.
#pragma once
#include <algorithm>
#include <limits>
#include "../common/io.h"
namespace pcl {
template <typename PointT>
class KdTreeFLANN {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
  int nearestKSearch(const PointT& p, int k, std::vector<int>& idx, std::vector<float>& d2) const {
    idx.clear(); d2.clear();
    if (!in_) return 0;
    std::vector<std::pair<float, int>> best;
    for (size_t i = 0; i < in_->points.size(); ++i) {
      const PointT& m = in_->points[i];
      const float dx = p.x - m.x, dy = p.y - m.y, dz = p.z - m.z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      best.emplace_back(d, (int)i);
      std::inplace_merge(best.begin(), best.end() - 1, best.end());
      if ((int)best.size() > k) best.pop_back();
    }
    for (auto& b : best) { d2.push_back(b.first); idx.push_back(b.second); }
    return (int)best.size();
  }
 private:
  typename PointCloud<PointT>::ConstPtr in_;
};
}  // namespace pcl
