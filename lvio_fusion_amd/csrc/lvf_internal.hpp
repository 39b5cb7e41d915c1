// lvf_internal.hpp — host-side objects behind the opaque C-ABI handles of include/lvf.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lvf.h"
#include "lvf_math.hpp"

namespace lvf {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define LVF_HIP(call)                                                          \
  do {                                                                         \
    hipError_t e__ = (call);                                                   \
    if (e__ != hipSuccess) return ::lvf::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)
#define LVF_REQUIRE(cond, ...)                                                 \
  do {                                                                         \
    if (!(cond)) { ::lvf::set_error(__VA_ARGS__); return LVF_ERR_INVALID; }    \
  } while (0)
#define LVF_TRY(expr)                                                          \
  do {                                                                         \
    int rc__ = (expr);                                                         \
    if (rc__ != LVF_OK) return rc__;                                           \
  } while (0)

template <typename T>
struct DevBuf {  // owning device buffer
  T* p = nullptr;
  size_t n = 0;
  size_t cap = 0;   // allocated elements (>= n); ensure()/assign() only ever grow it
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t count) {
    if (p) { (void)hipFree(p); p = nullptr; }
    n = count; cap = count;
    if (count == 0) return LVF_OK;
    LVF_HIP(hipMalloc(&p, count * sizeof(T)));
    return LVF_OK;
  }
  // grow-only resize (contents are NOT preserved across a growth): the persistent-window path re-uses buffers across ticks
  int ensure(size_t count) {
    if (count <= cap && (p || count == 0)) { n = count; return LVF_OK; }
    const size_t want = count + count / 4 + 16;
    if (p) { (void)hipFree(p); p = nullptr; }
    LVF_HIP(hipMalloc(&p, want * sizeof(T)));
    cap = want; n = count;
    return LVF_OK;
  }
  int assign(const T* host, size_t count, hipStream_t s) {
    LVF_TRY(ensure(count));
    if (count) LVF_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    return LVF_OK;
  }
  int upload(const T* host, size_t count, hipStream_t s) {
    LVF_TRY(alloc(count));
    if (count) LVF_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    return LVF_OK;
  }
};

}  // namespace lvf

struct lvf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cu = 256;
};

struct lvf_state {
  lvf_ctx* ctx = nullptr;
  int n_kf = 0, n_lm = 0;
  lvf::DevBuf<double> poses, vel, ba, bg, inv_depth, w_visual;
};

enum lvf_batch_kind { LVF_K_POSE_ONLY = 0, LVF_K_TWO_FRAME = 1, LVF_K_TWO_CAMERA = 2, LVF_K_IMU = 3, LVF_K_LIDAR = 4, LVF_K_POSE_PRIOR = 5 };

struct lvf_batch {
  lvf_ctx* ctx = nullptr;
  int kind = 0;
  int n = 0;
  int n_res = 0;             // residuals per block
  int n_blocks = 0;          // parameter blocks per residual block
  int block_size[8] = {0};
  bool evaluated = false, have_jac = false;
  bool sorted_by_kf = false; // blocks ordered by (current) keyframe
  int min_n_kf = 0, min_n_lm = 0;  // smallest state the index arrays are valid for
  std::vector<int32_t> host_kf1, host_kf2;  // two-frame batches keep their keyframe indices for the solver's work list
  lvf::CamD cam_a{}, cam_b{};  // visual: cam_a = left / cam0, cam_b = right
  // inputs (device)
  lvf::DevBuf<double> ob_a, ob_b;        // [n][2] each
  lvf::DevBuf<int32_t> idx_a, idx_b, idx_c;
  lvf::DevBuf<double> table;             // pose-only: pw[n_pw][3]
  int n_table = 0;
  // lidar
  int lidar_mode = 0;
  double lidar_weight = 1.0;
  double Twc1[7] = {0, 0, 0, 1, 0, 0, 0};
  lvf::DevBuf<double> lp, lpa, lnrm;     // SoA [3][n]
  lvf::DevBuf<char> icp_dev;             // device LM state of lvf_lidar_solve
  // imu
  lvf::DevBuf<double> pre;               // [n][467] flattened lvf_preint
  lvf::DevBuf<double> sqrt_info;         // [n][225]
  // outputs (device)
  lvf::DevBuf<double> res;
  lvf::DevBuf<double> jac[8];
};

#define LVF_MAX_GRID_LEVELS 4
struct lvf_map {
  struct Level {
    lvf::DevBuf<float4> sorted;      // cell-sorted points, .w = bitcast original index
    lvf::DevBuf<int> cell_start;     // ncells + 1
    float ox = 0, oy = 0, oz = 0, cell = 1, inv_cell = 1;
    int nx = 1, ny = 1, nz = 1;
    Level() = default;
    Level(Level&& o) noexcept { *this = std::move(o); }
    Level& operator=(Level&& o) noexcept {
      std::swap(sorted.p, o.sorted.p); std::swap(sorted.n, o.sorted.n);
      std::swap(cell_start.p, o.cell_start.p); std::swap(cell_start.n, o.cell_start.n);
      ox = o.ox; oy = o.oy; oz = o.oz; cell = o.cell; inv_cell = o.inv_cell; nx = o.nx; ny = o.ny; nz = o.nz;
      return *this;
    }
  };
  lvf_ctx* ctx = nullptr;
  int M = 0;
  lvf::DevBuf<float4> raw;           // original order (x,y,z,.) — gather source for plane fitting
  int n_levels = 0;
  Level levels[LVF_MAX_GRID_LEVELS]; // [0] = finest ... [n_levels-1] = coarsest (cell = gate radius / 2)
};

struct lvf_scan {
  lvf_ctx* ctx = nullptr;
  int Q = 0;
  lvf::DevBuf<float4> pts;           // body-frame scan points
  lvf::DevBuf<int> idx;              // [Q][3]
  lvf::DevBuf<float> d2;             // [Q][3]
  lvf::DevBuf<uint8_t> valid;        // [Q]
  bool searched = false;
  lvf::DevBuf<double> corr;          // ICP correspondences: p | pa | n, each SoA [3][Q]
  lvf::DevBuf<char> icp_dev;         // device-resident LM state of lvf_icp_solve
};

struct lvf_cloud {
  lvf_ctx* ctx = nullptr;
  int n = 0;
  lvf::DevBuf<float4> pts;           // x, y, z, intensity  (pcl::PointXYZI payload, 16 B on device)
};

struct lvf_problem;
namespace lvf {
// (re)derives a problem's dimensions from its state's CURRENT n_kf / n_lm, grows its work buffers if needed and rebuilds the
// TwoFrame work list; called by lvf_problem_create and, every tick, by the persistent window (window.hip)
int problem_configure(lvf_problem* p);
int device_exclusive_scan_i32(lvf_ctx* ctx, const int* in, int n, int* out);
int compact_points(lvf_ctx* ctx, const float4* pts, int n, const int* flags_dev, lvf_cloud** out);
// kernels / launchers implemented in the .hip translation units
int launch_pose_only(lvf_batch* b, const lvf_state* st, bool want_j);
int launch_two_frame(lvf_batch* b, const lvf_state* st, bool want_j);
int launch_two_camera(lvf_batch* b, const lvf_state* st, bool want_j);
int launch_lidar_normals(lvf_batch* b, const double* d_pb, const double* d_pc);
int launch_lidar_plane(lvf_batch* b, const double* rpyxyz_host, bool want_j);
int launch_imu_sqrt_info(lvf_batch* b);
int launch_imu(lvf_batch* b, const lvf_state* st, bool want_j);
int launch_pose_prior(lvf_batch* b, const lvf_state* st, bool want_j);
void make_camd(const lvf_camera& c, CamD& d);
}  // namespace lvf
