// scan_match_dev.hpp — device-side records shared by the association (knn_kernels.hip), the 3-DoF solve (icp_kernels.hip) and the
// batched scan-to-map update (scan_match.hip): Mapping::Relocate's per-candidate loop (src/lvio_fusion/src/mapping.cpp:251-300) for MANY
// candidates at once (src/lvio_fusion/src/relocator.cpp:196-206), one chain of launches with the candidate in blockIdx.y.
#pragma once
#include <cstdint>
#include "lvf_internal.hpp"

namespace lvf {

struct GridP { float ox, oy, oz, cell, inv_cell; int nx, ny, nz; };
struct LevelP { GP<const float4> sorted; GP<const int> cell_start; GridP g; };
struct LevelsP { LevelP l[LVF_MAX_GRID_LEVELS]; int n; };   // l[0] = finest ... l[n-1] = coarsest (cell >= gate radius / 2)

// LM state of one scan-to-map sub-problem (ceres::Solve's TrustRegionMinimizer on 3 unknowns: oracle/icp.h), device-resident
struct IcpDev {
  double x[3], x0[3], xc[3];
  double radius, decrease;
  double acc[11];            // H lower (00,10,11,20,21,22), g (3), cost — at x ; [10] = number of valid correspondences (first pass)
  double accC[11];           // the same sums at the candidate xc (one pass per LM iteration: an accepted candidate's linearisation is already there)
  double cost_cand;          // (unused since the candidate pass carries its Jacobian: kept for the record layout's readers)
  double cost_cur, initial_cost, model;
  double rpyxyz[6];
  int done, iters, successes, nvalid, first;
  int invalid_run;           // consecutive invalid steps (solver failure or model_cost_change <= 0): 5 end the solve
  unsigned ticket;           // workgroups that have finished the running k_icp_eval (the last one does the scalar tail)
  int count_valid;           // 1: nvalid is taken from acc[10] by the first step (batched chain: nobody else counts)
  double h0[3];              // Jacobi scaling of this solve: diag(J^T J) at its first linearisation, frozen (Ceres default jacobi_scaling; mapping.cpp:159-163 solves with default options)
};

struct IcpArgs {
  double Twc1[7];
  double weight, huber, prior_w;
  double function_tolerance, gradient_tolerance, parameter_tolerance, min_relative_decrease;
  int mode, max_iters;
};

// One candidate of a batched scan-to-map update.  Everything Mapping::Relocate keeps on its stack between the two sub-problems of an
// outer iteration lives here, so nothing crosses PCIe until the records are read back.
struct SmDev {
  double map_pose[7], minv[7], pose[7], linv[7];
  double rpyxyz[6];
  float tf[8];               // pose.cast<float>() for the association (association.cpp:287)
  IcpDev icp;
  double score[2], initial_cost[2], final_cost[2];      // [0] ground, [1] surf: of the LAST outer iteration (mapping.cpp:279-280, :293-294)
  int nres[2], iters[2], succ[2];
  int has[2];                // the sub-problem exists (map and scan given, map not empty)
  int has_last;
  double relative_o_c[7];
  int score_int;
};

// association of one (candidate, sub-problem): table entry [2 * candidate + sub]
struct KnnJob {
  GP<const float4> scan; int Q; LevelsP L; float thr;
  GP<int> idx; GP<float> d2; GP<uint8_t> valid;
  GP<const float4> map_raw; GP<double> corr;      // correspondences P | PA | N, SoA [3][Q] each
};
struct IcpJob { int Q; GP<const double> P, PA, N; GP<const uint8_t> valid; IcpArgs args; };

constexpr int kIcpMaxBlocks = 128;   // k_icp_eval grid cap (grid-stride above it)

// launchers (the kernels stay in their translation units: the association is compiled without FMA contraction)
int launch_knn3_batch(hipStream_t q, const KnnJob* jobs, const SmDev* devs, int n, int sub, int max_Q);      // association + correspondence build
int launch_knn3_build(lvf_map* m, lvf_scan* sc, const double* pose, float thr, double* corr);                 // the same for one (map, scan) pair
int launch_icp_eval_batch(hipStream_t q, const IcpJob* jobs, SmDev* devs, int n, int sub, int max_Q, bool candidate, bool last);   // candidate: the pass at xc (decision + next step) instead of the opening pass at x; last: cost only
LevelsP levels_of(const lvf_map* m);

}  // namespace lvf
