// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/common/common_headers.h>: the point / cloud type names common.h:38-43
// typedefs (+ the cloud fields utility.h's filter template names).  No algorithm inside.
#pragma once
#include <cstdint>
#include <vector>
namespace pcl {
struct PointXYZ { float x, y, z, pad; };
struct PointXYZI { float x, y, z, pad; float intensity, pad2[3]; };
struct PointXYZRGB { float x, y, z, pad; float rgb, pad2[3]; };
struct PCLHeader { std::uint32_t seq = 0; std::uint64_t stamp = 0; };
template <typename PointT> struct PointCloud { PCLHeader header; std::vector<PointT> points; std::uint32_t width = 0, height = 0; bool is_dense = true; };
}  // namespace pcl
