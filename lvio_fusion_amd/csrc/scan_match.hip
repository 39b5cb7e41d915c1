// scan_match.hip — scan-to-map updates as Mapping::Optimize / Mapping::Relocate run them
// (src/lvio_fusion/src/mapping.cpp:147-178 and :251-300), for ONE frame or for MANY loop-closure candidates at once
// (src/lvio_fusion/src/relocator.cpp:196-206).  Per outer iteration and candidate
//     rpyxyz = se32rpyxyz(map_pose^-1 * pose)
//     ground sub-problem (pitch, roll, z)  -> pose = map_pose * rpyxyz2se3(rpyxyz)
//     surf   sub-problem (yaw, x, y)       -> pose = map_pose * rpyxyz2se3(rpyxyz)      (same rpyxyz array, re-associated)
// A sub-problem is an association (k_knn3) + the correspondence build + up to max_num_iterations Levenberg-Marquardt steps.
//
// MI355X design: a candidate is a CHAIN of small launches (a few hundred points each in a relocalisation) — latency, not throughput — so
// the candidates run side by side in the same launches (blockIdx.y = candidate, per-candidate jobs in device tables) and everything the
// reference keeps on its stack between sub-problems (pose, rpyxyz, scores) lives in a device record (SmDev): the pose is composed on the
// device by one thread per candidate (k_sm_step), the float transform of the next association is taken from there, and nothing crosses
// PCIe until the records are read back at the end — one read-back per call instead of one per sub-problem.  lvf_scan_match is the batch of
// one.  Eight configs[4] candidates: 5.8 ms one after the other (16 read-backs each) -> see DESIGN.md for the batched figure.
#include <algorithm>
#include <vector>
#include "host_se3.hpp"
#include "lvf_internal.hpp"
#include "scan_match_dev.hpp"

namespace lvf {

// ---- SE3 / rpyxyz on device: the same formulas as host_se3.hpp (Sophus SE3d product / inverse on unit quaternions, utility.cpp:27-40)
namespace dse3 {
__device__ inline void normalize4(double q[4]) {
  const double s = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] *= s;
}
__device__ inline void rotate(const double q_in[4], const double p[3], double o[3]) {
  double q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
  normalize4(q);
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double cx = y * p[2] - z * p[1], cy = z * p[0] - x * p[2], cz = x * p[1] - y * p[0];
  const double dx = y * cz - z * cy, dy = z * cx - x * cz, dz = x * cy - y * cx;
  o[0] = p[0] + 2.0 * (w * cx + dx); o[1] = p[1] + 2.0 * (w * cy + dy); o[2] = p[2] + 2.0 * (w * cz + dz);
}
__device__ inline void quat_mul(const double a[4], const double b[4], double o[4]) {
  const double aw = a[3], ax = a[0], ay = a[1], az = a[2], bw = b[3], bx = b[0], by = b[1], bz = b[2];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by - ax * bz + ay * bw + az * bx;
  o[2] = aw * bz + ax * by - ay * bx + az * bw;
}
__device__ inline void mul(const double A[7], const double B[7], double C[7]) {
  double q[4], t[3];
  quat_mul(A, B, q);
  normalize4(q);
  rotate(A, B + 4, t);
  for (int k = 0; k < 4; ++k) C[k] = q[k];
  for (int k = 0; k < 3; ++k) C[4 + k] = A[4 + k] + t[k];
}
__device__ inline void to_rpyxyz(const double T[7], double r[6]) {
  const double w = T[3], x = T[0], y = T[1], z = T[2];
  r[0] = atan2(2.0 * (x * y + w * z), 1.0 - 2.0 * (y * y + z * z));
  r[1] = asin(2.0 * (w * y - x * z));
  r[2] = atan2(2.0 * (y * z + w * x), 1.0 - 2.0 * (x * x + y * y));
  r[3] = T[4]; r[4] = T[5]; r[5] = T[6];
}
__device__ inline void from_rpyxyz(const double r[6], double T[7]) {
  const double hz = r[0] / 2.0, hy = r[1] / 2.0, hx = r[2] / 2.0;
  const double cz = cos(hz), sz = sin(hz), cy = cos(hy), sy = sin(hy), cx = cos(hx), sx = sin(hx);
  double q[4];
  q[3] = cz * cy * cx + sz * sy * sx;
  q[0] = cz * cy * sx - sz * sy * cx;
  q[1] = cz * sy * cx + sz * cy * sx;
  q[2] = sz * cy * cx - cz * sy * sx;
  normalize4(q);
  for (int k = 0; k < 4; ++k) T[k] = q[k];
  T[4] = r[3]; T[5] = r[4]; T[6] = r[5];
}
}  // namespace dse3

struct SmStepArgs {
  int n;
  int prev, next;            // sub-problem to close / to open (0 ground, 1 surf, -1 none)
  int first_of_outer;        // `next` opens an outer iteration: rpyxyz is re-derived from the pose (mapping.cpp:154, :264-266)
  double prior_w;            // > 0: the PoseErrorRPZ / YXY block counts as a residual block (Summary::num_residual_blocks_reduced)
  double cap[2];             // 20 / 30 (mapping.cpp:279, :293)
};

// One thread per candidate between two sub-problems: what Mapping::Relocate does on the host between two ceres::Solve calls.
__global__ __launch_bounds__(64) void k_sm_step(SmDev* __restrict__ devs, const SmStepArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.n) return;
  SmDev& d = devs[c];
  if (a.prev >= 0 && d.has[a.prev]) {
    const int m = a.prev;
    const int i0 = m == 0 ? 1 : 0, i1 = m == 0 ? 2 : 3, i2 = m == 0 ? 5 : 4;
    d.rpyxyz[i0] = d.icp.x[0]; d.rpyxyz[i1] = d.icp.x[1]; d.rpyxyz[i2] = d.icp.x[2];
    double dT[7], pose[7];
    dse3::from_rpyxyz(d.rpyxyz, dT);
    dse3::mul(d.map_pose, dT, pose);                               // frame->pose = map_frame->pose * rpyxyz2se3(rpyxyz)
    for (int k = 0; k < 7; ++k) d.pose[k] = pose[k];
    const int nr = d.icp.nvalid + (a.prior_w > 0.0 ? 1 : 0);
    d.nres[m] = nr; d.iters[m] = d.icp.iters; d.succ[m] = d.icp.successes; d.initial_cost[m] = d.icp.initial_cost; d.final_cost[m] = d.icp.cost_cur;
    d.score[m] = fmin((double)nr / 10, a.cap[m]) - (nr > 0 ? 2 * d.icp.cost_cur / nr : 0.0);
  }
  if (a.next >= 0) {
    if (a.first_of_outer) {
      double rel[7], r[6];
      dse3::mul(d.minv, d.pose, rel);
      dse3::to_rpyxyz(rel, r);
      for (int k = 0; k < 6; ++k) d.rpyxyz[k] = r[k];
    }
    IcpDev& h = d.icp;
    const int m = a.next;
    const int i0 = m == 0 ? 1 : 0, i1 = m == 0 ? 2 : 3, i2 = m == 0 ? 5 : 4;
    h.x[0] = h.x0[0] = d.rpyxyz[i0]; h.x[1] = h.x0[1] = d.rpyxyz[i1]; h.x[2] = h.x0[2] = d.rpyxyz[i2];
    h.xc[0] = h.xc[1] = h.xc[2] = 0.0;
    for (int k = 0; k < 6; ++k) h.rpyxyz[k] = d.rpyxyz[k];
    h.radius = 1e4; h.decrease = 2.0;
    for (int k = 0; k < 11; ++k) { h.acc[k] = 0.0; h.accC[k] = 0.0; }
    h.cost_cand = 0.0; h.cost_cur = 0.0; h.initial_cost = 0.0; h.model = 0.0;
    h.done = d.has[m] ? 0 : 1; h.iters = 0; h.successes = 0; h.nvalid = 0; h.first = 1; h.invalid_run = 0; h.ticket = 0u; h.count_valid = 1;
    for (int k = 0; k < 7; ++k) d.tf[k] = (float)d.pose[k];        // Sophus SE3d::cast<float>()
  } else {
    if (d.has_last) dse3::mul(d.linv, d.pose, d.relative_o_c);     // mapping.cpp:298
    else for (int k = 0; k < 7; ++k) d.relative_o_c[k] = d.pose[k];
    d.score_int = (int)(d.score[0] + d.score[1]);                  // int Mapping::Relocate(...) truncates
  }
}

}  // namespace lvf

using namespace lvf;

namespace {
struct SmJobView {
  lvf_map* map[2]; lvf_scan* scan[2];
  const double *map_pose, *frame_pose, *last_pose;
};

// the whole update for n candidates of one context; results[n]
int scan_match_run(lvf_ctx* ctx, const SmJobView* jobs, int n, const lvf_scan_match_options* opt, lvf_scan_match_result* out) {
  LVF_TRY(lvf::enter(ctx));
  hipStream_t q = ctx->stream;
  // host tables in ONE pinned block: records | knn jobs | icp jobs
  const size_t b_dev = (size_t)n * sizeof(SmDev), b_knn = (size_t)2 * n * sizeof(KnnJob), b_icp = (size_t)2 * n * sizeof(IcpJob);
  HostPin<char> stage;
  StreamWaitGuard stage_guard(q);          // an error between the table's upload and the wait below must not hand the pinned block back while the copy runs
  LVF_TRY(stage.reserve(b_dev + b_knn + b_icp));
  std::memset(stage.p, 0, b_dev + b_knn + b_icp);
  SmDev* hd = reinterpret_cast<SmDev*>(stage.p);
  KnnJob* hk = reinterpret_cast<KnnJob*>(stage.p + b_dev);
  IcpJob* hi = reinterpret_cast<IcpJob*>(stage.p + b_dev + b_knn);
  int max_Q[2] = {0, 0};
  bool any[2] = {false, false};
  for (int c = 0; c < n; ++c) {
    const SmJobView& J = jobs[c];
    SmDev& d = hd[c];
    std::memcpy(d.map_pose, J.map_pose, sizeof(d.map_pose));
    std::memcpy(d.pose, J.frame_pose, sizeof(d.pose));
    hse3::inv(J.map_pose, d.minv);
    d.has_last = J.last_pose ? 1 : 0;
    if (J.last_pose) hse3::inv(J.last_pose, d.linv);
    for (int m = 0; m < 2; ++m) {
      lvf_map* mp = J.map[m]; lvf_scan* sc = J.scan[m];
      d.has[m] = (mp && sc && mp->M > 0) ? 1 : 0;
      KnnJob& k = hk[2 * c + m]; IcpJob& ij = hi[2 * c + m];
      IcpArgs& a = ij.args;
      std::memcpy(a.Twc1, J.map_pose, sizeof(a.Twc1));
      a.weight = m == 0 ? opt->weight_ground : opt->weight_surf; a.huber = m == 0 ? 0.0 : opt->huber_surf; a.prior_w = opt->prior_weight;
      a.function_tolerance = 1e-6; a.gradient_tolerance = 1e-10; a.parameter_tolerance = 1e-8; a.min_relative_decrease = 1e-3;
      a.mode = m; a.max_iters = opt->max_num_iterations;
      if (!d.has[m]) continue;
      const int Q = sc->Q;
      if (sc->corr.n < (size_t)9 * std::max(Q, 1)) LVF_TRY(sc->corr.alloc((size_t)9 * std::max(Q, 1)));
      k.scan = sc->pts.p; k.Q = Q; k.L = levels_of(mp); k.thr = m == 0 ? opt->thr_ground : opt->thr_surf;
      k.idx = sc->idx.p; k.d2 = sc->d2.p; k.valid = sc->valid.p; k.map_raw = mp->raw.p; k.corr = sc->corr.p;
      ij.Q = Q; ij.P = sc->corr.p; ij.PA = ij.P + (size_t)3 * Q; ij.N = ij.PA + (size_t)3 * Q; ij.valid = sc->valid.p;
      max_Q[m] = std::max(max_Q[m], Q); any[m] = true;
      sc->searched = true;
    }
  }
  DevBuf<char> tab;
  LVF_TRY(tab.alloc(b_dev + b_knn + b_icp));
  LVF_HIP(hipMemcpyAsync(tab.p, stage.p, b_dev + b_knn + b_icp, hipMemcpyHostToDevice, q));
  SmDev* dd = reinterpret_cast<SmDev*>(tab.p);
  const KnnJob* dk = reinterpret_cast<const KnnJob*>(tab.p + b_dev);
  const IcpJob* di = reinterpret_cast<const IcpJob*>(tab.p + b_dev + b_knn);
  SmStepArgs sa{};
  sa.n = n; sa.prior_w = opt->prior_weight; sa.cap[0] = 20.0; sa.cap[1] = 30.0;
  const dim3 gs((n + 63) / 64);
  int prev = -1;
  for (int it = 0; it < opt->outer_iterations; ++it) {
    bool first = true;
    for (int m = 0; m < 2; ++m) {
      if (!any[m]) continue;
      sa.prev = prev; sa.next = m; sa.first_of_outer = first ? 1 : 0;
      hipLaunchKernelGGL(k_sm_step, gs, dim3(64), 0, q, dd, sa);
      first = false;
      LVF_TRY(launch_knn3_batch(q, dk, dd, n, m, max_Q[m]));            // association.cpp:287-301 / :345-359 and the correspondences, :303-314 / :361-372
      LVF_TRY(launch_icp_eval_batch(q, di, dd, n, m, max_Q[m], false, false));            // the pass at x: linearisation + first step
      for (int li = 0; li < opt->max_num_iterations; ++li)
        LVF_TRY(launch_icp_eval_batch(q, di, dd, n, m, max_Q[m], true, li + 1 == opt->max_num_iterations));      // one pass per LM iteration, at its candidate
      prev = m;
    }
    if (first) {     // no sub-problem at all: rpyxyz still follows the pose, nothing to solve
      break;
    }
  }
  sa.prev = prev; sa.next = -1; sa.first_of_outer = 0;
  hipLaunchKernelGGL(k_sm_step, gs, dim3(64), 0, q, dd, sa);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(stage.p, tab.p, b_dev, hipMemcpyDeviceToHost, q));
  LVF_HIP(hipStreamSynchronize(q));
  stage_guard.dismiss();
  for (int c = 0; c < n; ++c) {
    const SmDev& d = hd[c];
    lvf_scan_match_result& r = out[c];
    std::memset(&r, 0, sizeof(r));
    std::memcpy(r.pose, d.pose, sizeof(r.pose));
    std::memcpy(r.relative_o_c, d.relative_o_c, sizeof(r.relative_o_c));
    r.score_ground = d.score[0]; r.score_surf = d.score[1]; r.score = d.score_int;
    lvf_icp_summary* s[2] = {&r.ground, &r.surf};
    for (int m = 0; m < 2; ++m) {
      s[m]->initial_cost = d.initial_cost[m]; s[m]->final_cost = d.final_cost[m]; s[m]->num_residual_blocks = d.nres[m];
      s[m]->num_iterations = d.iters[m]; s[m]->num_successful_steps = d.succ[m];
    }
  }
  return LVF_OK;
}
}  // namespace

extern "C" {

void lvf_scan_match_options_default(lvf_scan_match_options* o, double resolution) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->thr_ground = (float)(resolution * resolution * 100.0);   // association.cpp:285
  o->thr_surf = (float)(resolution * resolution * 25.0);      // association.cpp:343
  o->weight_ground = 1.0; o->weight_surf = (double)0.01f;     // frame.cpp:14-15 — Weights' fields are floats (adapt/weights.h:10-12): the functors receive float(0.01)
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
  lvf_ctx* ctx = map_ground ? map_ground->ctx : (map_surf ? map_surf->ctx : nullptr);
  if (!ctx) {          // no cloud at all: the reference skips both sub-problems and the pose stays
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->pose, frame_pose, sizeof(out->pose));
    if (last_pose) { double linv[7]; hse3::inv(last_pose, linv); hse3::mul(linv, frame_pose, out->relative_o_c); }
    else std::memcpy(out->relative_o_c, frame_pose, sizeof(out->pose));
    return LVF_OK;
  }
  for (const lvf_map* m : {map_ground, map_surf}) LVF_REQUIRE(!m || m->ctx == ctx, "lvf_scan_match: handles of different contexts");
  for (const lvf_scan* s : {scan_ground, scan_surf}) LVF_REQUIRE(!s || s->ctx == ctx, "lvf_scan_match: handles of different contexts");
  const SmJobView j{{map_ground, map_surf}, {scan_ground, scan_surf}, map_pose, frame_pose, last_pose};
  return scan_match_run(ctx, &j, 1, opt, out);
}

int lvf_scan_match_batch(lvf_ctx* ctx, const lvf_scan_match_job* jobs, int n, const lvf_scan_match_options* opt, int relocate_base_score,
                         lvf_scan_match_result* results, int* best) {
  LVF_REQUIRE(ctx && opt && results && n >= 0 && (n == 0 || jobs), "lvf_scan_match_batch: null argument");
  LVF_REQUIRE(opt->outer_iterations >= 1 && opt->max_num_iterations >= 0, "lvf_scan_match_batch: bad options");
  if (best) *best = -1;
  if (n == 0) return LVF_OK;
  std::vector<SmJobView> v((size_t)n);
  for (int c = 0; c < n; ++c) {
    const lvf_scan_match_job& J = jobs[c];
    LVF_REQUIRE((J.map_ground != nullptr) == (J.scan_ground != nullptr) && (J.map_surf != nullptr) == (J.scan_surf != nullptr),
                "lvf_scan_match_batch: candidate %d: a map and its scan must be given together", c);
    for (const lvf_map* m : {J.map_ground, J.map_surf}) LVF_REQUIRE(!m || m->ctx == ctx, "lvf_scan_match_batch: candidate %d: handle of another context", c);
    for (const lvf_scan* s : {J.scan_ground, J.scan_surf}) LVF_REQUIRE(!s || s->ctx == ctx, "lvf_scan_match_batch: candidate %d: handle of another context", c);
    // a scan handle carries ONE set of outputs (idx / d2 / valid / correspondences): it may appear once in the whole batch, whatever its role
    LVF_REQUIRE(!J.scan_ground || J.scan_ground != J.scan_surf, "lvf_scan_match_batch: candidate %d uses one scan handle as ground AND surf (its outputs are per scan)", c);
    for (int e = 0; e < c; ++e)
      for (const lvf_scan* a : {J.scan_ground, J.scan_surf})
        LVF_REQUIRE(!a || (a != jobs[e].scan_ground && a != jobs[e].scan_surf),
                    "lvf_scan_match_batch: candidates %d and %d share a scan handle (its outputs are per scan)", e, c);
    v[c] = SmJobView{{J.map_ground, J.map_surf}, {J.scan_ground, J.scan_surf}, J.map_pose, J.frame_pose, J.has_last_pose ? J.last_pose : nullptr};
  }
  LVF_TRY(scan_match_run(ctx, v.data(), n, opt, results));
  if (best) {          // relocator.cpp:196-206: score - base > 0 qualifies, `>=` keeps the LATER of equal scores
    double max_score = -1.0;
    for (int c = 0; c < n; ++c) {
      const double s = (double)(results[c].score - relocate_base_score);
      if (s > 0.0 && s >= max_score) { max_score = s; *best = c; }
    }
  }
  return LVF_OK;
}

}  // extern "C"
