// reference_patch/lvio_fusion/adapt/problem.h — shadows src/lvio_fusion/include/lvio_fusion/adapt/problem.h (:11-88) on the include path.
// The reference's header comes in UNCHANGED (#include_next) with its `adapt` namespace renamed to `adapt_host`; ProblemType and
// init_num_types stay where they are.  lvio_fusion::adapt::Problem derives from the reference's class — types, num_types, num_frames,
// GetTypes are the reference's own — and adds what INTEGRATION.md §2 describes:
//   hook 1  AddResidualBlock      also hands the block to the recorder (the window's SoA payload is captured while BuildProblem runs),
//   hook 2  AddParameterBlock     also registers pose blocks with the recorder,
//   hook 3  SetParameterBlockConstant / Variable   shadow the base methods (pose_graph.cpp, environment.cpp call them on an adapt::Problem),
// and adapt::Solve (:83-88) calls lvio_fusion::gpu::Solve instead of ceres::Solve.
#pragma once
#define adapt adapt_host
#include_next "lvio_fusion/adapt/problem.h"
#undef adapt

#include "lvf_ceres_adapter.hpp"

namespace lvio_fusion
{
namespace adapt
{

class Problem : public adapt_host::Problem
{
public:
    template <typename... Ts>
    void AddResidualBlock(ProblemType type, ceres::CostFunction *cost_function, ceres::LossFunction *loss_function, double *x0, Ts *... xs)
    {
        adapt_host::Problem::AddResidualBlock(type, cost_function, loss_function, x0, xs...);
        double *const params[] = {x0, xs...};
        recorder.AddResidualBlock(cost_function, loss_function, params, 1 + (int)sizeof...(xs));          // [hook 1 of 3]
    }

    void AddParameterBlock(double *values, int size)
    {
        adapt_host::Problem::AddParameterBlock(values, size);
        recorder.AddParameterBlock(values, size);                                                         // [hook 2 of 3]
    }

    void AddParameterBlock(double *values, int size, ceres::LocalParameterization *local_parameterization)
    {
        adapt_host::Problem::AddParameterBlock(values, size, local_parameterization);
        recorder.AddParameterBlock(values, size);                                                         // [hook 2 of 3]
    }

    void SetParameterBlockConstant(double *values)                                                        // [hook 3 of 3]
    {
        ceres::Problem::SetParameterBlockConstant(values);
        recorder.SetParameterBlockConstant(values);
    }

    void SetParameterBlockVariable(double *values)
    {
        ceres::Problem::SetParameterBlockVariable(values);
        recorder.Invalidate();
    }

    gpu::Recorder recorder; // the window's SoA payload, captured while BuildProblem adds the blocks
};

inline void Solve(const ceres::Solver::Options &options, adapt::Problem *problem, ceres::Solver::Summary *summary)
{
    gpu::Solve(options, problem, summary, &problem->recorder); // was: ceres::Solve(options, problem, summary);
}

} // namespace adapt
} // namespace lvio_fusion
