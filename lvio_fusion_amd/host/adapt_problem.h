// adapt_problem.h — the reference's thin Problem wrapper, restated for the adapter self-test and as the template of the
// integration patch: same members and method names as lvio_fusion::adapt::Problem
// (src/lvio_fusion/include/lvio_fusion/adapt/problem.h:11-88), with adapt::Solve routed to the MI355X library instead of
// ceres::Solve (:83-88).  Interface mirror only — the bodies are bookkeeping.
#pragma once
#include <map>
#include <unordered_map>
#include <vector>

#include "lvf_ceres_adapter.hpp"

namespace lvio_fusion {

enum class ProblemType { VisualError, WeakError, LidarError, NavsatError, PoseError, ImuError, Other };

inline std::map<ProblemType, int> zero_type_counts() {
  std::map<ProblemType, int> m;
  for (ProblemType t : {ProblemType::VisualError, ProblemType::WeakError, ProblemType::LidarError, ProblemType::NavsatError,
                        ProblemType::PoseError, ProblemType::ImuError, ProblemType::Other}) m[t] = 0;
  return m;
}

namespace adapt {

class Problem : public ceres::Problem {
 public:
  template <typename... Ts>
  void AddResidualBlock(ProblemType type, ceres::CostFunction* cost_function, ceres::LossFunction* loss_function, double* x0, Ts*... xs) {
    const ceres::ResidualBlockId id = ceres::Problem::AddResidualBlock(cost_function, loss_function, x0, xs...);
    types[id] = type;
    ++num_types[type];
    double* const params[] = {x0, xs...};                                                       // [integration hook 1 of 3]
    recorder.AddResidualBlock(cost_function, loss_function, params, 1 + (int)sizeof...(xs));
  }
  void AddParameterBlock(double* values, int size) { ceres::Problem::AddParameterBlock(values, size); recorder.AddParameterBlock(values, size); }
  void AddParameterBlock(double* values, int size, ceres::LocalParameterization* local_parameterization) {
    if (size == 7) ++num_frames;   // SE3d::num_parameters
    ceres::Problem::AddParameterBlock(values, size, local_parameterization);
    recorder.AddParameterBlock(values, size);                                                   // [integration hook 2 of 3]
  }
  // (shadows ceres::Problem's: backend.cpp / pose_graph.cpp call it on an adapt::Problem object)
  void SetParameterBlockConstant(double* values) { ceres::Problem::SetParameterBlockConstant(values); recorder.SetParameterBlockConstant(values); }   // [hook 3 of 3]
  void SetParameterBlockVariable(double* values) { ceres::Problem::SetParameterBlockVariable(values); recorder.Invalidate(); }
  gpu::Recorder recorder;      // the window's SoA payload, captured while the blocks are added
  std::map<ProblemType, int> GetTypes(double* para) {
    std::vector<ceres::ResidualBlockId> ids;
    GetResidualBlocksForParameterBlock(para, &ids);
    std::map<ProblemType, int> out = zero_type_counts();
    for (auto id : ids) ++out[types[id]];
    return out;
  }
  int num_frames = 0;
  std::unordered_map<ceres::ResidualBlockId, ProblemType> types;
  std::map<ProblemType, int> num_types = zero_type_counts();
};

// the one-line change of the integration: ceres::Solve -> lvio_fusion::gpu::Solve
inline void Solve(const ceres::Solver::Options& options, adapt::Problem* problem, ceres::Solver::Summary* summary) {
  gpu::Solve(options, problem, summary, &problem->recorder);
}

}  // namespace adapt
}  // namespace lvio_fusion
