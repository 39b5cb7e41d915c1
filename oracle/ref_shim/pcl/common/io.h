// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/common/io.h> / <pcl/filters/filter.h>: copyPointCloud (field-wise copy between
// point types; a PointXYZ source leaves intensity 0) and removeNaNFromPointCloud (drops points with a non-finite x, y or z, keeps order).
#pragma once
#include <cmath>
#include "common_headers.h"
namespace pcl {
template <typename A, typename B>
inline void copyPointCloud(const PointCloud<A>& in, PointCloud<B>& out) {
  out.header = in.header; out.width = in.width; out.height = in.height; out.is_dense = in.is_dense;
  out.points.resize(in.points.size());
  for (size_t i = 0; i < in.points.size(); ++i) { B b; b.x = in.points[i].x; b.y = in.points[i].y; b.z = in.points[i].z; out.points[i] = b; }
}
template <typename A>
inline void copyPointCloud(const PointCloud<A>& in, PointCloud<A>& out) { if (&in != &out) out = in; }
template <typename A>
inline void removeNaNFromPointCloud(const PointCloud<A>& in, PointCloud<A>& out, std::vector<int>& index) {
  std::vector<A> kept; index.clear();
  for (size_t i = 0; i < in.points.size(); ++i) {
    const A& p = in.points[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    kept.push_back(p); index.push_back((int)i);
  }
  out.header = in.header; out.points.swap(kept); out.width = (std::uint32_t)out.points.size(); out.height = 1; out.is_dense = true;
}
}  // namespace pcl
