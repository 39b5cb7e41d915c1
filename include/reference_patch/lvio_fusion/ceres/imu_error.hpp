// reference_patch/lvio_fusion/ceres/imu_error.hpp — shadows src/lvio_fusion/include/lvio_fusion/ceres/imu_error.hpp on the include path.
// ImuError::Create (:115-118) snapshots the pre-integration (read-only during a solve: preintegration_ is only dereferenced inside Evaluate,
// :17-108) into the MI355X library's tagged cost function; ImuInitError / ImuInitGError (initialisation, not on the hot path) are untouched.
#pragma once
#define ImuError ImuError_host
#include_next "lvio_fusion/ceres/imu_error.hpp"
#undef ImuError

#include "lvf_ceres_adapter.hpp"

namespace lvio_fusion
{

class ImuError : public ImuError_host
{
public:
    using ImuError_host::ImuError_host;
    static ceres::CostFunction *Create(imu::Preintegration::Ptr pre)
    {
        lvf_preint s;
        s.sum_dt = pre->sum_dt;
        for (int i = 0; i < 3; ++i)
        {
            s.lin_ba[i] = pre->linearized_ba[i];
            s.lin_bg[i] = pre->linearized_bg[i];
            s.dp[i] = pre->delta_p[i];
            s.dv[i] = pre->delta_v[i];
        }
        s.dq[0] = pre->delta_q.x(); s.dq[1] = pre->delta_q.y(); s.dq[2] = pre->delta_q.z(); s.dq[3] = pre->delta_q.w();
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c)
            {
                s.jac[15 * r + c] = pre->jacobian(r, c);
                s.cov[15 * r + c] = pre->covariance(r, c);
            }
        return gpu::ImuError::Create(s);
    }
};

} // namespace lvio_fusion
