// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/common/common_headers.h>: the point / cloud type names common.h:38-43
// typedefs.  No algorithm inside.
#pragma once
#include <vector>
namespace pcl {
struct PointXYZ { float x, y, z, pad; };
struct PointXYZI { float x, y, z, pad; float intensity, pad2[3]; };
struct PointXYZRGB { float x, y, z, pad; float rgb, pad2[3]; };
template <typename PointT> struct PointCloud { std::vector<PointT> points; };
}  // namespace pcl
