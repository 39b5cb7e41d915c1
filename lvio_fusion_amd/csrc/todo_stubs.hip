// todo_stubs.hip — entry points of include/lvf.h whose device implementation lands later this round.
#include "lvf_internal.hpp"
using namespace lvf;
extern "C" {
#define LVF_TODO(name) do { set_error(name ": not implemented yet"); return LVF_ERR_STATE; } while (0)
int lvf_icp_solve(lvf_map*, lvf_scan*, const double*, double*, const lvf_icp_options*, lvf_icp_summary*) { LVF_TODO("lvf_icp_solve"); }
}
