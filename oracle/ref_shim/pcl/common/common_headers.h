// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/common/common_headers.h>: the point / cloud types common.h:38-43 typedefs, as plain
// containers (std::vector underneath, the element accessors / push_back / concatenation the reference's lidar code uses:
// projection.cpp, association.cpp:39-235, utility.h:70-96).  No algorithm inside.
#pragma once
#include <math.h>
#include <cstdint>
#include <memory>
#include <vector>
#include "../../boost/make_shared.hpp"
namespace pcl {
struct PointXYZ { union { float data[4]; struct { float x, y, z; }; }; PointXYZ() : data{0, 0, 0, 1} {} };
struct PointXYZI { union { float data[4]; struct { float x, y, z; }; }; union { struct { float intensity; }; float data_c[4]; }; PointXYZI() : data{0, 0, 0, 1}, data_c{0, 0, 0, 0} {} };
struct PointXYZRGB { union { float data[4]; struct { float x, y, z; }; }; union { struct { float rgb; }; struct { std::uint8_t b, g, r, a; }; float data_c[4]; }; PointXYZRGB() : data{0, 0, 0, 1}, data_c{0, 0, 0, 0} {} };
struct PCLHeader { std::uint32_t seq = 0; std::uint64_t stamp = 0; };
template <typename PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  typedef typename std::vector<PointT>::iterator iterator;
  typedef typename std::vector<PointT>::const_iterator const_iterator;
  PCLHeader header; std::vector<PointT> points; std::uint32_t width = 0, height = 0; bool is_dense = true;
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = height = 0; }
  void push_back(const PointT& p) { points.push_back(p); width = (std::uint32_t)points.size(); height = 1; }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  const_iterator begin() const { return points.begin(); }
  const_iterator end() const { return points.end(); }
  template <typename It> void insert(iterator pos, It first, It last) { points.insert(pos, first, last); width = (std::uint32_t)points.size(); height = 1; }
  PointCloud& operator+=(const PointCloud& o) { points.insert(points.end(), o.points.begin(), o.points.end()); width = (std::uint32_t)points.size(); height = 1; return *this; }
  PointCloud operator+(const PointCloud& o) const { PointCloud r = *this; r += o; return r; }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
struct PointIndices { typedef std::shared_ptr<PointIndices> Ptr; std::vector<int> indices; };
struct ModelCoefficients { typedef std::shared_ptr<ModelCoefficients> Ptr; std::vector<float> values; };
}  // namespace pcl
