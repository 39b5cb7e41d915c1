// reference_patch/lvio_fusion/ceres/lidar_error.hpp — shadows src/lvio_fusion/include/lvio_fusion/ceres/lidar_error.hpp on the include path.
// LidarPlaneError (:10-40) is untouched; LidarPlaneErrorRPZ::Create (:65-69) and LidarPlaneErrorYXY::Create (:100-104) return the MI355X library's
// tagged cost functions.  rpyxyz stays the caller's LIVE pointer (:52,:87): the library reads it when the problem is solved.
#pragma once
#define LidarPlaneErrorRPZ LidarPlaneErrorRPZ_host
#define LidarPlaneErrorYXY LidarPlaneErrorYXY_host
#include_next "lvio_fusion/ceres/lidar_error.hpp"
#undef LidarPlaneErrorRPZ
#undef LidarPlaneErrorYXY

#include "lvf_ceres_adapter.hpp"

namespace lvio_fusion
{

class LidarPlaneErrorRPZ : public LidarPlaneErrorRPZ_host
{
public:
    using LidarPlaneErrorRPZ_host::LidarPlaneErrorRPZ_host;
    static ceres::CostFunction *Create(Vector3d p, Vector3d pa, Vector3d pb, Vector3d pc, SE3d Twc1, double *rpyxyz, double weight)
    {
        return gpu::LidarPlaneErrorRPZ::Create(p.data(), pa.data(), pb.data(), pc.data(), Twc1.data(), rpyxyz, weight);
    }
};

class LidarPlaneErrorYXY : public LidarPlaneErrorYXY_host
{
public:
    using LidarPlaneErrorYXY_host::LidarPlaneErrorYXY_host;
    static ceres::CostFunction *Create(Vector3d p, Vector3d pa, Vector3d pb, Vector3d pc, SE3d Twc1, double *rpyxyz, double weight)
    {
        return gpu::LidarPlaneErrorYXY::Create(p.data(), pa.data(), pb.data(), pc.data(), Twc1.data(), rpyxyz, weight);
    }
};

} // namespace lvio_fusion
