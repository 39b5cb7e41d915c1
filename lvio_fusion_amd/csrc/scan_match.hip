// scan_match.hip — one frame's scan-to-map update as Mapping::Optimize / Mapping::Relocate run it
// (src/lvio_fusion/src/mapping.cpp:147-178 and :251-300): per outer iteration
//     rpyxyz = se32rpyxyz(map_pose^-1 * pose)
//     ground sub-problem (pitch, roll, z)  -> pose = map_pose * rpyxyz2se3(rpyxyz)
//     surf   sub-problem (yaw, x, y)       -> pose = map_pose * rpyxyz2se3(rpyxyz)      (same rpyxyz array, re-associated)
// Each sub-problem is one lvf_icp_solve (association + 3-DoF LM on device); only the 6-double rpyxyz crosses PCIe between
// them.  This is the unit of work config 5 shards one-per-GPU (loop-closure candidates, src/relocator.cpp:196-206).
#include <algorithm>
#include "host_se3.hpp"
#include "lvf_internal.hpp"

using namespace lvf;

extern "C" {

void lvf_scan_match_options_default(lvf_scan_match_options* o, double resolution) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->thr_ground = (float)(resolution * resolution * 100.0);   // association.cpp:285
  o->thr_surf = (float)(resolution * resolution * 25.0);      // association.cpp:343
  o->weight_ground = 1.0; o->weight_surf = 0.01;              // frame.cpp:14-15
  o->huber_surf = 0.1;                                        // association.cpp:330
  o->prior_weight = 0.0;                                      // relocate mode
  o->outer_iterations = 1;                                    // Mapping::Optimize; Mapping::Relocate uses 4
  o->max_num_iterations = 4;                                  // mapping.cpp:161
}

int lvf_scan_match(lvf_map* map_ground, lvf_scan* scan_ground, lvf_map* map_surf, lvf_scan* scan_surf, const double* map_pose,
                   const double* frame_pose, const double* last_pose, const lvf_scan_match_options* opt, lvf_scan_match_result* out) {
  LVF_REQUIRE(map_pose && frame_pose && opt && out, "lvf_scan_match: null argument");
  LVF_REQUIRE((map_ground != nullptr) == (scan_ground != nullptr) && (map_surf != nullptr) == (scan_surf != nullptr),
              "lvf_scan_match: a map and its scan must be given together");
  LVF_REQUIRE(opt->outer_iterations >= 1 && opt->max_num_iterations >= 0, "lvf_scan_match: bad options");
  std::memset(out, 0, sizeof(*out));
  double pose[7], minv[7];
  std::memcpy(pose, frame_pose, sizeof(pose));
  hse3::inv(map_pose, minv);
  for (int it = 0; it < opt->outer_iterations; ++it) {
    double rel[7], rpyxyz[6], d[7];
    hse3::mul(minv, pose, rel);
    hse3::to_rpyxyz(rel, rpyxyz);
    if (map_ground && map_ground->M > 0) {
      const lvf_icp_options o{0, opt->thr_ground, opt->weight_ground, 0.0, opt->prior_weight, opt->max_num_iterations};
      LVF_TRY(lvf_icp_solve(map_ground, scan_ground, map_pose, pose, rpyxyz, &o, &out->ground));
      hse3::from_rpyxyz(rpyxyz, d);
      hse3::mul(map_pose, d, pose);
      const int nr = out->ground.num_residual_blocks;    // mapping.cpp:279-280
      out->score_ground = std::min((double)nr / 10, 20.0) - (nr > 0 ? 2 * out->ground.final_cost / nr : 0.0);
    }
    if (map_surf && map_surf->M > 0) {
      const lvf_icp_options o{1, opt->thr_surf, opt->weight_surf, opt->huber_surf, opt->prior_weight, opt->max_num_iterations};
      LVF_TRY(lvf_icp_solve(map_surf, scan_surf, map_pose, pose, rpyxyz, &o, &out->surf));
      hse3::from_rpyxyz(rpyxyz, d);
      hse3::mul(map_pose, d, pose);
      const int nr = out->surf.num_residual_blocks;      // mapping.cpp:293-294
      out->score_surf = std::min((double)nr / 10, 30.0) - (nr > 0 ? 2 * out->surf.final_cost / nr : 0.0);
    }
  }
  std::memcpy(out->pose, pose, sizeof(pose));
  out->score = (int)(out->score_ground + out->score_surf);   // int Mapping::Relocate(...) truncates
  if (last_pose) {
    double linv[7];
    hse3::inv(last_pose, linv);
    hse3::mul(linv, pose, out->relative_o_c);                // mapping.cpp:298
  } else {
    std::memcpy(out->relative_o_c, pose, sizeof(pose));
  }
  return LVF_OK;
}

}  // extern "C"
