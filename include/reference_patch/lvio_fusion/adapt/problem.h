// reference_patch/lvio_fusion/adapt/problem.h — shadows src/lvio_fusion/include/lvio_fusion/adapt/problem.h (:11-88) on the include path.
// Same enum, same class, same members and method names (interface mirror: the reference's translation units compile against it unchanged);
// the differences are the three recorder hooks of INTEGRATION.md §2 and adapt::Solve -> lvio_fusion::gpu::Solve (:83-88).
#ifndef lvio_fusion_PROBLEM_H
#define lvio_fusion_PROBLEM_H

#include "lvio_fusion/common.h"

#include <ceres/ceres.h>

#include "lvf_ceres_adapter.hpp"

namespace lvio_fusion
{

enum class ProblemType
{
    VisualError,
    WeakError,
    LidarError,
    NavsatError,
    PoseError,
    ImuError,
    Other
};

const std::map<ProblemType, int> init_num_types = {
    {ProblemType::VisualError, 0}, {ProblemType::WeakError, 0}, {ProblemType::LidarError, 0}, {ProblemType::NavsatError, 0},
    {ProblemType::PoseError, 0},   {ProblemType::ImuError, 0},  {ProblemType::Other, 0}};

namespace adapt
{

class Problem : public ceres::Problem
{
public:
    template <typename... Ts>
    void AddResidualBlock(ProblemType type, ceres::CostFunction *cost_function, ceres::LossFunction *loss_function, double *x0, Ts *... xs)
    {
        ceres::ResidualBlockId id = ceres::Problem::AddResidualBlock(cost_function, loss_function, x0, xs...);
        types[id] = type;
        num_types[type]++;
        double *const params[] = {x0, xs...};                                                             // [hook 1 of 3]
        recorder.AddResidualBlock(cost_function, loss_function, params, 1 + (int)sizeof...(xs));
    }

    void AddParameterBlock(double *values, int size)
    {
        ceres::Problem::AddParameterBlock(values, size);
        recorder.AddParameterBlock(values, size);                                                         // [hook 2 of 3]
    }

    void AddParameterBlock(double *values, int size, ceres::LocalParameterization *local_parameterization)
    {
        if (size == SE3d::num_parameters)
        {
            num_frames++;
        }
        ceres::Problem::AddParameterBlock(values, size, local_parameterization);
        recorder.AddParameterBlock(values, size);                                                         // [hook 2 of 3]
    }

    // (shadow the base methods: pose_graph.cpp / environment.cpp call them on an adapt::Problem object)
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

    std::map<ProblemType, int> GetTypes(double *para)
    {
        std::vector<ceres::ResidualBlockId> residual_blocks;
        GetResidualBlocksForParameterBlock(para, &residual_blocks);
        std::map<ProblemType, int> result = init_num_types;
        for (auto i : residual_blocks)
        {
            result[types[i]]++;
        }
        return result;
    }

    int num_frames = 0;
    std::unordered_map<ceres::ResidualBlockId, ProblemType> types;
    std::map<ProblemType, int> num_types = init_num_types;
    gpu::Recorder recorder; // the window's SoA payload, captured while BuildProblem adds the blocks
};

inline void Solve(const ceres::Solver::Options &options, adapt::Problem *problem, ceres::Solver::Summary *summary)
{
    gpu::Solve(options, problem, summary, &problem->recorder); // was: ceres::Solve(options, problem, summary);
}

} // namespace adapt
} // namespace lvio_fusion

#endif // lvio_fusion_PROBLEM_H
