// ORACLE — TEST INFRASTRUCTURE ONLY.  <ceres/ceres.h> of the COMPILED DROP-IN (oracle/ref_driver_dropin.cpp, `make -C oracle dropin`):
// the reference's backend.cpp / association.cpp / landmark.cpp / preintegration.cpp are compiled UNMODIFIED against
//   * include/lvf_ceres_compat.h — the slice of the Ceres public API the MI355X adapter (include/lvf_ceres_adapter.hpp) walks
//     (ceres::Problem with its accessors, Solver::Options / Summary); in a catkin workspace this is the real <ceres/ceres.h>;
//   * ref_shim/ceres/autodiff_shim.h — AutoDiffCostFunction over the stand-in Jet, for the reference's functors that stay on the host
//     (the *_host classes behind include/reference_patch/, NavsatError, ImuInitGError ...).
// ceres::Solve — which the reference calls directly at mapping.cpp:277,291 and pose_graph.cpp:206 — is declared here and DEFINED in the
// drop-in driver as a call of lvio_fusion::gpu::Solve: there is no CPU solver in this build either.
#pragma once
#include "lvf_ceres_compat.h"
#include <ceres/autodiff_shim.h>

namespace ceres {
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary);
}  // namespace ceres
