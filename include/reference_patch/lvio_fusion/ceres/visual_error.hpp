// reference_patch/lvio_fusion/ceres/visual_error.hpp — shadows src/lvio_fusion/include/lvio_fusion/ceres/visual_error.hpp on the include path.
// The reference's three functor classes come in UNCHANGED under *_host names (their templated operator() is what the outlier gate at
// src/backend.cpp:185-190 calls on doubles); the public names derive from them and replace ONLY the factories:
//   PoseOnlyReprojectionError::Create   (:66-70)    TwoFrameReprojectionError::Create (:98-102)    TwoCameraReprojectionError::Create (:128-132)
// which now return the MI355X library's tagged cost functions (include/lvf_ceres_adapter.hpp) — same argument types, order and meaning.
#pragma once
#define PoseOnlyReprojectionError PoseOnlyReprojectionError_host
#define TwoFrameReprojectionError TwoFrameReprojectionError_host
#define TwoCameraReprojectionError TwoCameraReprojectionError_host
#include_next "lvio_fusion/ceres/visual_error.hpp"
#undef PoseOnlyReprojectionError
#undef TwoFrameReprojectionError
#undef TwoCameraReprojectionError

#include <algorithm>

#include "lvf_ceres_adapter.hpp"

namespace lvio_fusion
{

// Camera = intrinsics + extrinsic (sensor.h:41-44, visual/camera.h:74); SE3d::data() = [qx,qy,qz,qw,tx,ty,tz] (ceres/base.hpp:29)
inline lvf_camera to_lvf(const Camera::Ptr &c)
{
    lvf_camera k{c->fx, c->fy, c->cx, c->cy, {}};
    std::copy(c->extrinsic.data(), c->extrinsic.data() + 7, k.extrinsic);
    return k;
}

class PoseOnlyReprojectionError : public PoseOnlyReprojectionError_host
{
public:
    using PoseOnlyReprojectionError_host::PoseOnlyReprojectionError_host;
    static ceres::CostFunction *Create(Vector2d ob, Vector3d pw, Camera::Ptr camera, double weight)
    {
        return gpu::PoseOnlyReprojectionError::Create(ob.data(), pw.data(), to_lvf(camera), weight);
    }
};

class TwoFrameReprojectionError : public TwoFrameReprojectionError_host
{
public:
    using TwoFrameReprojectionError_host::TwoFrameReprojectionError_host;
    static ceres::CostFunction *Create(Vector2d first_ob, Vector2d ob, Camera::Ptr left, Camera::Ptr right, double weight)
    {
        return gpu::TwoFrameReprojectionError::Create(first_ob.data(), ob.data(), to_lvf(left), to_lvf(right), weight);
    }
};

class TwoCameraReprojectionError : public TwoCameraReprojectionError_host
{
public:
    using TwoCameraReprojectionError_host::TwoCameraReprojectionError_host;
    static ceres::CostFunction *Create(Vector2d left_ob, Vector2d right_ob, Camera::Ptr left, Camera::Ptr right, double weight)
    {
        return gpu::TwoCameraReprojectionError::Create(left_ob.data(), right_ob.data(), to_lvf(left), to_lvf(right), weight);
    }
};

} // namespace lvio_fusion
