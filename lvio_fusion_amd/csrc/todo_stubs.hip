// todo_stubs.hip — entry points of include/lvf.h whose device implementation lands later this round.
#include "lvf_internal.hpp"
using namespace lvf;
extern "C" {
#define LVF_TODO(name) do { set_error(name ": not implemented yet"); return LVF_ERR_STATE; } while (0)
int lvf_preintegrate(lvf_ctx*, int, const int32_t*, const double*, const double*, const double*, const double*, const double*, const double*, lvf_preint*) { LVF_TODO("lvf_preintegrate"); }
int lvf_icp_solve(lvf_map*, lvf_scan*, const double*, double*, const lvf_icp_options*, lvf_icp_summary*) { LVF_TODO("lvf_icp_solve"); }
void lvf_solver_options_default(lvf_solver_options* o) {
  if (!o) return;
  o->max_num_iterations = 50; o->max_solver_time_in_seconds = 0.0; o->huber_a = 1.0;
  o->initial_trust_region_radius = 1e4; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8; o->min_relative_decrease = 1e-3;
}
int lvf_problem_create(lvf_ctx*, lvf_state*, lvf_batch*, lvf_batch*, lvf_batch*, lvf_batch*, lvf_problem**) { LVF_TODO("lvf_problem_create"); }
int lvf_problem_destroy(lvf_problem*) { return LVF_OK; }
int lvf_problem_set_pose_constant(lvf_problem*, int, int) { LVF_TODO("lvf_problem_set_pose_constant"); }
int lvf_problem_cost(lvf_problem*, const lvf_solver_options*, double*) { LVF_TODO("lvf_problem_cost"); }
int lvf_problem_lm_iteration(lvf_problem*, const lvf_solver_options*, double*, double*, double*, double*, int*) { LVF_TODO("lvf_problem_lm_iteration"); }
int lvf_problem_solve(lvf_problem*, const lvf_solver_options*, lvf_solver_summary*) { LVF_TODO("lvf_problem_solve"); }
int lvf_problem_reduced_dim(lvf_problem*) { return -1; }
int lvf_problem_download_reduced(lvf_problem*, double*, double*) { LVF_TODO("lvf_problem_download_reduced"); }
}
