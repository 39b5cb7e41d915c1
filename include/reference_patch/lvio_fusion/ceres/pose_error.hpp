// reference_patch/lvio_fusion/ceres/pose_error.hpp — shadows src/lvio_fusion/include/lvio_fusion/ceres/pose_error.hpp on the include path.
// PoseGraphError::Create (:40-49, both overloads), PoseError::Create (:78-81), RError::Create (:103-106), PoseErrorRPZ::Create (:155-158) and
// PoseErrorYXY::Create (:183-186) and RelocateRError::Create (:215-218: the blocks of Relocator::UpdateNewSubmap's rotation solve, which gpu::Solve
// runs as ONE device launch, lvf_relocate_rotation_solve) return the MI355X library's tagged cost functions; TError keeps the reference's host class.
#pragma once
#define PoseGraphError PoseGraphError_host
#define PoseError PoseError_host
#define RError RError_host
#define PoseErrorRPZ PoseErrorRPZ_host
#define PoseErrorYXY PoseErrorYXY_host
#define RelocateRError RelocateRError_host
#include_next "lvio_fusion/ceres/pose_error.hpp"
#undef PoseGraphError
#undef PoseError
#undef RError
#undef PoseErrorRPZ
#undef PoseErrorYXY
#undef RelocateRError

#include "lvf_ceres_adapter.hpp"

namespace lvio_fusion
{

class PoseGraphError : public PoseGraphError_host
{
public:
    using PoseGraphError_host::PoseGraphError_host;
    static ceres::CostFunction *Create(SE3d last_pose, SE3d pose, double weight = 1, double v = 1)
    {
        return gpu::PoseGraphError::Create(last_pose.data(), pose.data(), weight, v);
    }
    static ceres::CostFunction *Create(SE3d relative_i_j, double weight = 1, double v = 1)
    {
        return gpu::PoseGraphError::Create(relative_i_j.data(), weight, v);
    }
};

class PoseError : public PoseError_host
{
public:
    using PoseError_host::PoseError_host;
    static ceres::CostFunction *Create(SE3d pose, double weight = 1, double v = 1)
    {
        return gpu::PoseError::Create(pose.data(), weight, v);
    }
};

class RError : public RError_host
{
public:
    using RError_host::RError_host;
    static ceres::CostFunction *Create(SE3d pose, double weight = 1)
    {
        return gpu::RError::Create(pose.data(), weight);
    }
};

class PoseErrorRPZ : public PoseErrorRPZ_host
{
public:
    using PoseErrorRPZ_host::PoseErrorRPZ_host;
    static ceres::CostFunction *Create(double *rpyxyz, double weight = 1)
    {
        return gpu::PoseErrorRPZ::Create(rpyxyz, weight);
    }
};

class PoseErrorYXY : public PoseErrorYXY_host
{
public:
    using PoseErrorYXY_host::PoseErrorYXY_host;
    static ceres::CostFunction *Create(double *rpyxyz, double weight = 1)
    {
        return gpu::PoseErrorYXY::Create(rpyxyz, weight);
    }
};

class RelocateRError : public RelocateRError_host
{
public:
    using RelocateRError_host::RelocateRError_host;
    static ceres::CostFunction *Create(SE3d relocated, SE3d unrelocated)
    {
        return gpu::RelocateRError::Create(relocated.data(), unrelocated.data());
    }
};

} // namespace lvio_fusion
