// window.hip — a PERSISTENT sliding window (the problem-assembly step immediately before the hot path).
//
// The reference rebuilds its whole Ceres problem from the pointer graph twice per backend tick
// (Backend::BuildProblem, src/lvio_fusion/src/backend.cpp:96-183, called from Optimize :206 and UpdateFrontend :261): one
// heap-allocated functor per observation, std::map walks over frames -> features -> landmarks.  Here the window is kept as
// flat host arrays that are updated INCREMENTALLY when the front-end adds a keyframe / landmark / observation or the window
// slides, and the device SoA batches, parameter state and solver work buffers live across ticks (grow-only buffers, no
// hipMalloc / hipFree per tick).  lvf_window_solve assembles the block lists with BuildProblem's rules —
//   feature of frame k, landmark born in k            -> TwoCameraReprojectionError (weight 5 w_k)            :117-125
//   landmark's birth frame older than the window       -> PoseOnlyReprojectionError on landmark->ToWorld()    :126-131
//   otherwise                                          -> TwoFrameReprojectionError(inv_depth, birth, k)      :132-140
//   good_imu frame after a good_imu frame              -> ImuError                                            :143-161
//   no IMU block and < 20 near visual blocks           -> PoseGraphError(last, k, 100, 0) / PoseError(k, 100, 0)  :163-178
// ("near" = !Camera::Far: depth in cam0 <= 50 baselines, include/lvio_fusion/visual/camera.h:38-41) — in frame order, features
// by ascending landmark id (features_left is a std::map keyed by landmark id), uploads them with one copy per array and runs
// the device LM loop; results are read back into the host mirror.
#include <algorithm>
#include <chrono>
#include <iterator>
#include <map>
#include <unordered_map>
#include "host_se3.hpp"
#include "lvf_internal.hpp"

namespace lvf {
struct alignas(64) LmHot { double right_ob[2]; double pw[3]; int64_t birth_kf; int bpos; int slot; int fixed; int pad; };
static_assert(sizeof(LmHot) == 64, "one cache line per landmark");
}  // namespace lvf

struct lvf_window {
  struct Obs { int64_t lm_id; int lm; double ob[2]; };  // lm = index into lms
  struct Kf {
    int64_t id = 0;
    double pose[7], vel[3] = {0, 0, 0}, ba[3] = {0, 0, 0}, bg[3] = {0, 0, 0};
    double w_visual = 1.0;
    bool good_imu = false, has_pre = false;
    lvf_preint pre;
    unsigned pre_ver = 0;                                // identifies the CONTENT of `pre` (lvf_window::pre_counter at the time it last changed)
    std::vector<Obs> obs;                                // kept sorted by landmark id (BuildProblem's iteration order) lazily
    bool sorted = true;
    bool d_dirty = true;                                 // device-side assembly: the frame's segment of the feature arena is stale
    size_t d_off = 0, d_cap = 0;                         // ... its place there (elements)
    void put(const Obs& o) { if (!obs.empty() && o.lm_id <= obs.back().lm_id) sorted = false; obs.push_back(o); d_dirty = true; }
    void sort_unique() {                                 // std::map semantics: ascending key, a re-inserted key overwrites
      if (sorted) return;
      std::stable_sort(obs.begin(), obs.end(), [](const Obs& a, const Obs& b) { return a.lm_id < b.lm_id; });
      size_t o = 0;
      for (size_t i = 0; i < obs.size(); ++i) {
        if (i + 1 < obs.size() && obs[i + 1].lm_id == obs[i].lm_id) continue;   // keep the last of equal keys
        obs[o++] = obs[i];
      }
      obs.resize(o);
      sorted = true;
    }
  };
  struct Lm {
    int64_t id = 0, birth_kf = 0;
    double left_ob[2], right_ob[2], inv_depth = 0.0;
    bool fixed = false;                                  // birth frame left the window: world point frozen
    double pw[3] = {0, 0, 0};
    int slot = -1;                                       // dense index in the current assembly (-1: not in the problem)
    bool alive = true;                                   // false: a free entry of `lms` (indices are STABLE: Obs::lm and the device tables keep them)
  };
  lvf_ctx* ctx = nullptr;
  lvf_camera left, right;
  lvf_window_options opt;
  std::vector<Kf> kfs;                                   // active keyframes, oldest first
  std::unordered_map<int64_t, int> kf_index;             // id -> position in kfs
  std::unordered_map<int64_t, std::array<double, 7>> departed;   // poses of frames that left the window (for ToWorld)
  std::vector<Lm> lms;                                   // entries are re-used through lm_free, never moved
  std::vector<int> lm_free;
  // device-side assembly keeps the landmark table RESIDENT: only entries the host changed since the last tick are sent (the solved
  // inverse depths are written into the table by the device itself)
  std::vector<int> lm_dirty; std::vector<uint8_t> lm_dirty_flag; bool lmtab_valid = false;
  void mark_lm(int i) {
    if (lm_dirty_flag.size() <= (size_t)i) lm_dirty_flag.resize((size_t)i + 1 + lm_dirty_flag.size() / 2, 0);
    if (!lm_dirty_flag[i]) { lm_dirty_flag[i] = 1; lm_dirty.push_back(i); }
  }
  int n_lm_live = 0;
  std::unordered_map<int64_t, int> lm_index;
  // device side, persistent across ticks
  lvf_state* st = nullptr;
  lvf_batch *tc = nullptr, *tf = nullptr, *po = nullptr, *imu = nullptr, *prior = nullptr;
  lvf_batch* rej = nullptr;            // PoseOnly batch of the outlier gate (lvf_window_reject_outliers), persistent
  lvf_state* rej_st = nullptr;         // the window's poses with unit visual weights
  lvf::DevBuf<uint8_t> rej_flags;
  // pinned staging for the per-tick block lists (observations / indices per functor type)
  lvf::HostPin<double> h_state;      // poses | vel | ba | bg | w_visual | inv_depth (read-back staging)
  // device-side assembly (opt.device_assembly): resident feature arena + per-tick tables and scratch
  lvf::DevBuf<int32_t> d_obs_lm; lvf::DevBuf<double> d_obs_xy; size_t arena_end = 0;
  lvf::DevBuf<unsigned char> d_lmtab, d_frames, d_cls, d_used;
  lvf::DevBuf<int> d_wgcnt, d_wgbase, d_slot, d_slot_lm, d_counts;
  lvf::DevBuf<double> d_lm_invd_out;
  lvf::HostPin<int> h_counts;
  // ImuError information matrices across ticks: slot f of `sq_prev` holds sqrt_info of the pre-integration version sq_ver[f] (what the
  // previous tick left in the IMU batch); a factor whose version is found there is copied instead of factored again (k_imu_sqrt_info_cached)
  lvf::DevBuf<double> sq_prev; std::vector<unsigned> sq_ver; lvf::DevBuf<int> d_sq_src; unsigned pre_counter = 0; bool sq_valid = false;
  hipEvent_t ev_counts = nullptr;                        // recorded behind the counters' copy: the host waits for it, not for the work queued after it
  std::vector<lvf::LmHot> hot;                           // per-tick compact copy of what the feature walk reads of a landmark
  lvf::HostPin<unsigned char> h_stage;                   // the tick's packed upload (records + plain segments)
  lvf::DevBuf<unsigned char> d_stage;
  lvf_problem* prob = nullptr;
  // assembly of the last solve
  std::vector<int> slot_lm;                              // dense landmark slot -> index into lms
  int n_tc = 0, n_tf = 0, n_po = 0, n_imu = 0, n_prior = 0, n_lm_problem = 0;
  ~lvf_window() {
    if (prob) lvf_problem_destroy(prob);
    for (lvf_batch* b : {tc, tf, po, imu, prior, rej}) if (b) lvf_batch_destroy(b);
    if (st) lvf_state_destroy(st);
    if (rej_st) lvf_state_destroy(rej_st);
    if (ev_counts) (void)hipEventDestroy(ev_counts);
  }
};

namespace lvf {

// the TwoCamera block weight as the reference forms it (backend.cpp:123: `5 * frame->weights.visual`, float arithmetic: adapt/weights.h:10)
static inline double two_camera_weight(double w_visual) { return (double)(5.0f * (float)w_visual); }

// Landmark::ToWorld (src/lvio_fusion/src/landmark.cpp:15-19): Pixel2Robot through the RIGHT camera, then the birth pose
static void to_world(const lvf_camera& right, const double rob[2], double inv_depth, const double birth_pose[7], double pw[3]) {
  const double d = 1.0 / inv_depth;
  const double ps[3] = {(rob[0] - right.cx) * d / right.fx, (rob[1] - right.cy) * d / right.fy, d};
  double pb[3], r[3];
  hse3::rotate(right.extrinsic, ps, pb);
  for (int k = 0; k < 3; ++k) pb[k] += right.extrinsic[4 + k];
  hse3::rotate(birth_pose, pb, r);
  for (int k = 0; k < 3; ++k) pw[k] = r[k] + birth_pose[4 + k];
}
template <typename T>
static int put(DevBuf<T>& buf, const std::vector<T>& v, hipStream_t s) { return buf.assign(v.data(), v.size(), s); }

// Keeps the host mirror bounded over a long run (the reference removes a landmark once its last observation is gone:
// Landmark::RemoveObservation / Map::RemoveLandmark): landmarks no active keyframe observes any more are dropped and Obs::lm is
// re-indexed; poses of departed frames are kept only while a live landmark was born there.
static void prune(lvf_window* w) {
  const size_t nl = w->lms.size();
  std::vector<char> seen(nl, 0);
  for (const lvf_window::Kf& f : w->kfs)
    for (const lvf_window::Obs& o : f.obs) seen[o.lm] = 1;
  for (size_t i = 0; i < nl; ++i) {
    lvf_window::Lm& l = w->lms[i];
    if (l.alive && !seen[i]) {                 // the entry becomes free; nothing is moved, so every index held elsewhere stays valid
      w->lm_index.erase(l.id);
      l.alive = false; l.fixed = false; l.slot = -1;
      w->mark_lm((int)i);
      w->lm_free.push_back((int)i);
      --w->n_lm_live;
    }
  }
  if (!w->departed.empty()) {
    std::unordered_map<int64_t, char> anchor;
    for (const lvf_window::Lm& l : w->lms) if (l.alive) anchor[l.birth_kf] = 1;
    for (auto it = w->departed.begin(); it != w->departed.end();) it = anchor.count(it->first) ? std::next(it) : w->departed.erase(it);
  }
}

// landmark->ToWorld() (src/lvio_fusion/src/landmark.cpp:15-19) for every landmark of the mirror: frozen points as stored, live ones
// from their birth keyframe's CURRENT pose and inverse depth.  lm_birth_pos[i] = position of the birth keyframe in kfs (-1: not active).
// Rk[9 k] = rotation of keyframe k (row-major), Rk[9 n_kf] = rotation of the right camera's extrinsic (same construction as in
// landmarks_to_world: the device-side assembly multiplies with exactly these matrices)
static void landmarks_to_world_prepare(const lvf_window* w, double* Rk) {
  const int n_kf = (int)w->kfs.size();
  auto rot_of = [](const double q[4], double R[9]) {
    double e0[3] = {1, 0, 0}, e1[3] = {0, 1, 0}, e2[3] = {0, 0, 1}, c0[3], c1[3], c2[3];
    hse3::rotate(q, e0, c0); hse3::rotate(q, e1, c1); hse3::rotate(q, e2, c2);
    R[0] = c0[0]; R[1] = c1[0]; R[2] = c2[0]; R[3] = c0[1]; R[4] = c1[1]; R[5] = c2[1]; R[6] = c0[2]; R[7] = c1[2]; R[8] = c2[2];
  };
  for (int k = 0; k < n_kf; ++k) rot_of(w->kfs[k].pose, Rk + (size_t)9 * k);
  rot_of(w->right.extrinsic, Rk + (size_t)9 * n_kf);
}
static void landmarks_to_world(const lvf_window* w, std::vector<double>& lm_pw, std::vector<int>& lm_birth_pos) {
  const int n_kf = (int)w->kfs.size();
  auto rot_of = [](const double q[4], double R[9]) {
    double e0[3] = {1, 0, 0}, e1[3] = {0, 1, 0}, e2[3] = {0, 0, 1}, c0[3], c1[3], c2[3];
    hse3::rotate(q, e0, c0); hse3::rotate(q, e1, c1); hse3::rotate(q, e2, c2);
    R[0] = c0[0]; R[1] = c1[0]; R[2] = c2[0]; R[3] = c0[1]; R[4] = c1[1]; R[5] = c2[1]; R[6] = c0[2]; R[7] = c1[2]; R[8] = c2[2];
  };
  std::vector<double> Rk((size_t)9 * n_kf);
  for (int k = 0; k < n_kf; ++k) rot_of(w->kfs[k].pose, &Rk[(size_t)9 * k]);
  double Re[9];
  rot_of(w->right.extrinsic, Re);
  const double* te = w->right.extrinsic + 4;
  const size_t nl = w->lms.size();
  lm_pw.assign((size_t)3 * nl, 0.0);
  lm_birth_pos.assign(nl, -1);
  const int64_t first_id = w->kfs.front().id;
  std::vector<int64_t> ids(n_kf);                          // ascending: lvf_window_add_keyframe only accepts increasing ids
  for (int k = 0; k < n_kf; ++k) ids[k] = w->kfs[k].id;
  for (size_t i = 0; i < nl; ++i) {
    const lvf_window::Lm& l = w->lms[i];
    if (!l.alive) continue;
    if (l.fixed) { std::memcpy(&lm_pw[3 * i], l.pw, 24); continue; }
    if (l.birth_kf < first_id) continue;
    // position of the birth frame: the window's ids ascend, so a binary search over <= a few dozen ids (a hash lookup per landmark
    // was most of this function: 10 k lookups per tick)
    const int64_t* ib = std::lower_bound(ids.data(), ids.data() + n_kf, l.birth_kf);
    if (ib == ids.data() + n_kf || *ib != l.birth_kf) continue;
    const int bp = (int)(ib - ids.data());
    lm_birth_pos[i] = bp;
    const double d = 1.0 / l.inv_depth;
    const double ps[3] = {(l.right_ob[0] - w->right.cx) * d / w->right.fx, (l.right_ob[1] - w->right.cy) * d / w->right.fy, d};
    const double pb[3] = {Re[0] * ps[0] + Re[1] * ps[1] + Re[2] * ps[2] + te[0], Re[3] * ps[0] + Re[4] * ps[1] + Re[5] * ps[2] + te[1],
                          Re[6] * ps[0] + Re[7] * ps[1] + Re[8] * ps[2] + te[2]};
    const double* R = &Rk[(size_t)9 * bp];
    const double* t = w->kfs[bp].pose + 4;
    lm_pw[3 * i] = R[0] * pb[0] + R[1] * pb[1] + R[2] * pb[2] + t[0];
    lm_pw[3 * i + 1] = R[3] * pb[0] + R[4] * pb[1] + R[5] * pb[2] + t[1];
    lm_pw[3 * i + 2] = R[6] * pb[0] + R[7] * pb[1] + R[8] * pb[2] + t[2];
  }
}

// What the per-feature walk of lvf_window_solve needs of a landmark, in ONE cache line (the walk is bound by its random accesses:
// through Lm + two side arrays it touched four lines per feature)
static void landmarks_hot(const lvf_window* w, std::vector<LmHot>& hot) {
  std::vector<double> lm_pw;
  std::vector<int> lm_birth_pos;
  landmarks_to_world(w, lm_pw, lm_birth_pos);
  const size_t nl = w->lms.size();
  hot.resize(nl);
  for (size_t i = 0; i < nl; ++i) {
    const lvf_window::Lm& l = w->lms[i];
    LmHot& h = hot[i];
    h.right_ob[0] = l.right_ob[0]; h.right_ob[1] = l.right_ob[1]; h.pw[0] = lm_pw[3 * i]; h.pw[1] = lm_pw[3 * i + 1]; h.pw[2] = lm_pw[3 * i + 2];
    h.birth_kf = l.birth_kf; h.bpos = lm_birth_pos[i]; h.slot = -1; h.fixed = l.fixed ? 1 : 0; h.pad = 0;
  }
}

// ================================================================================================ device-side assembly
// The block lists of a tick assembled ON THE DEVICE from resident tables, so that the host neither walks ~10^5 features nor uploads
// ~4 MB of lists per tick:
//   feature arena   obs_lm[i] (landmark INDEX, stable: lvf_window::lms entries never move), obs_xy[i]; one segment per keyframe, in
//                   BuildProblem's order (ascending landmark id); only new / changed keyframes are uploaded
//   landmark table  LmDev[nl] (first observation, inverse depth, birth keyframe id, frozen world point) — re-sent every tick (64 B each)
//   frame table     FrameDev[n_kf] (id, segment, rotation + translation for Landmark::ToWorld, the Camera::Far row) — every tick
// k_da_classify : one thread per feature -> block type (BuildProblem's rules), per-workgroup and per-keyframe counts, landmarks in use
// k_da_scan     : one workgroup: output offsets per workgroup and type (features keep their order: keyframe by keyframe, ascending
//                 landmark id), dense landmark slots (ascending landmark index), inverse depths into the state
// k_da_emit     : one thread per feature -> its block's SoA entries
// The host reads back 4 + 2 n_kf counters (block totals, TwoFrame blocks and near visual blocks per keyframe) for the work list, the
// IMU / prior decisions and the launch shapes.
struct LmDev { double right_ob[2]; double pw[3]; double inv_depth; long long birth_kf; int fixed; int alive; };
// w_tc: the weight backend.cpp:123 hands TwoCameraReprojectionError::Create for a landmark born in this frame, `5 * frame->weights.visual` — a FLOAT
// product (adapt/weights.h:10 declares the weights float), widened to double by the call: (double)(5.0f * (float)w_visual)
struct FrameDev { long long id; int start, cnt; long long arena_off; double zrow[3], zoff; double R[9], t[3]; double w_tc; };
static_assert(sizeof(LmDev) == 64 && sizeof(FrameDev) % 8 == 0, "device table records");
constexpr int kDaMaxKf = 64;
struct DaArgs {
  int n_kf, total, nl, n_wg;
  const FrameDev* frames; const int32_t* obs_lm; const double2* obs_xy; const LmDev* lm;
  double Re[9], te[3], fx, fy, cx, cy, far_z;      // right camera (Landmark::ToWorld goes through it), Camera::Far threshold
  unsigned char* cls; unsigned char* used; int* wgcnt; int* wgbase; int* slot; int* slot_lm; int* counts;   // counts: [0..3] tc, tf, po, n_lm | tf per kf [64] | near per kf [64]
  double* state_invd; double* lm_invd_out;
  double2 *tc_l, *tc_r; int32_t *tc_lm, *tc_kf; double* tc_w;
  double2 *tf_f, *tf_o; int32_t *tf_lm, *tf_k1, *tf_k2;
  double2* po_o; double* po_pw; int32_t *po_kf, *po_pi;
};
// frame of global feature index g (frames' segments concatenated); s_start[n_kf] = total
__device__ __forceinline__ int da_frame_of(const int* s_start, int n_kf, int g) {
  int lo = 0, hi = n_kf - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_start[mid] <= g) lo = mid; else hi = mid - 1; }
  return lo;
}
__device__ __forceinline__ int da_birth_pos(const long long* s_id, int n_kf, long long id) {
  int lo = 0, hi = n_kf - 1;
  while (lo <= hi) { const int mid = (lo + hi) >> 1; if (s_id[mid] == id) return mid; if (s_id[mid] < id) lo = mid + 1; else hi = mid - 1; }
  return -1;
}
// Landmark::ToWorld (landmark.cpp:15-19) with the matrices the host expanded (same operation order as landmarks_to_world)
__device__ __forceinline__ void da_to_world(const DaArgs& a, const LmDev& L, const FrameDev& B, double pw[3]) {
  const double d = 1.0 / L.inv_depth;
  const double ps[3] = {(L.right_ob[0] - a.cx) * d / a.fx, (L.right_ob[1] - a.cy) * d / a.fy, d};
  const double pb[3] = {a.Re[0] * ps[0] + a.Re[1] * ps[1] + a.Re[2] * ps[2] + a.te[0], a.Re[3] * ps[0] + a.Re[4] * ps[1] + a.Re[5] * ps[2] + a.te[1],
                        a.Re[6] * ps[0] + a.Re[7] * ps[1] + a.Re[8] * ps[2] + a.te[2]};
  pw[0] = B.R[0] * pb[0] + B.R[1] * pb[1] + B.R[2] * pb[2] + B.t[0];
  pw[1] = B.R[3] * pb[0] + B.R[4] * pb[1] + B.R[5] * pb[2] + B.t[1];
  pw[2] = B.R[6] * pb[0] + B.R[7] * pb[1] + B.R[8] * pb[2] + B.t[2];
}
__global__ __launch_bounds__(256) void k_da_classify(DaArgs a) {
  __shared__ int s_start[kDaMaxKf + 1];
  __shared__ long long s_id[kDaMaxKf];
  __shared__ int s_cnt[3], s_tf[kDaMaxKf], s_near[kDaMaxKf];
  const int tid = threadIdx.x, g = blockIdx.x * 256 + tid;
  if (tid < a.n_kf) { s_start[tid] = a.frames[tid].start; s_id[tid] = a.frames[tid].id; s_tf[tid] = 0; s_near[tid] = 0; }
  if (tid == 0) s_start[a.n_kf] = a.total;
  if (tid < 3) s_cnt[tid] = 0;
  __syncthreads();
  if (g < a.total) {
    const int k = da_frame_of(s_start, a.n_kf, g);
    const FrameDev F = a.frames[k];
    const int lm = a.obs_lm[F.arena_off + (g - F.start)];
    const LmDev L = a.lm[lm];
    int c = 0;                                       // 0 none / 1 TwoCamera / 2 TwoFrame / 3 PoseOnly
    if (L.birth_kf == F.id) c = 1;
    else {
      const int bpos = da_birth_pos(s_id, a.n_kf, L.birth_kf);
      double pw[3];
      if (bpos < 0) { if (L.fixed) { c = 3; pw[0] = L.pw[0]; pw[1] = L.pw[1]; pw[2] = L.pw[2]; } }
      else { c = 2; da_to_world(a, L, a.frames[bpos], pw); }
      if (c && !(F.zrow[0] * pw[0] + F.zrow[1] * pw[1] + F.zrow[2] * pw[2] + F.zoff > a.far_z)) atomicAdd(&s_near[k], 1);      // !Camera::Far
      if (c == 2) atomicAdd(&s_tf[k], 1);
    }
    a.cls[g] = (unsigned char)c;
    if (c) atomicAdd(&s_cnt[c - 1], 1);
    if (c == 1 || c == 2) a.used[lm] = 1;
  }
  __syncthreads();
  if (tid < 3) a.wgcnt[3 * blockIdx.x + tid] = s_cnt[tid];
  if (tid < a.n_kf) {
    if (s_tf[tid]) atomicAdd(&a.counts[4 + tid], s_tf[tid]);
    if (s_near[tid]) atomicAdd(&a.counts[4 + kDaMaxKf + tid], s_near[tid]);
  }
}
__global__ __launch_bounds__(1024) void k_da_scan(DaArgs a) {
  // One workgroup, two barriers per 16 k landmarks: the scans run inside waves (shuffles), a Hillis-Steele scan of 1024 LDS entries
  // (20 workgroup barriers each, 16 waves) and a per-thread walk over runs of 64-byte landmark records made this launch 35 us.
  __shared__ int s_cnt[16 * 16];                    // used landmarks per (chunk, wave) of a pass, then their exclusive prefix
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (1) exclusive scan of the per-workgroup block counts: wave t takes type t, 64 workgroups at a time with a running carry
  if (wv < 3) {
    int carry = 0;
    for (int sb = 0; sb < a.n_wg; sb += 8 * 64) {      // 512 workgroups per pass, their counts requested up front
      int v[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) { const int i = sb + 64 * g + lane; const int t = a.wgcnt[3 * min(i, a.n_wg - 1) + wv]; v[g] = i < a.n_wg ? t : 0; }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int i = sb + 64 * g + lane;
        const int inc = wave_incl_scan(v[g]);
        if (i < a.n_wg) a.wgbase[3 * i + wv] = carry + inc - v[g];
        carry += __shfl(inc, 63);
      }
    }
    if (lane == 0) a.counts[wv] = carry;
  }
  // (2) dense landmark slots in ascending landmark index (thread = landmark: coalesced; rank = ballot prefix + wave prefix + chunk prefix).
  // 16 k landmarks per pass, the `used` flags and inverse depths of a pass requested up front (a load inside the per-1024 loop is waited
  // for before the next one is issued)
  constexpr int kG = 16;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  int total = 0;
  for (int sb = 0; sb < a.nl; sb += kG * 1024) {
    unsigned ubits = 0;
    double invd[kG];
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      const int l = sb + g * 1024 + tid, lc = min(l, a.nl - 1);
      const unsigned char u = a.used[lc];                // (read unconditionally: behind `l < a.nl &&` each of the 16 loads sat in its own branch and was waited for there)
      ubits |= ((l < a.nl) & (u != 0)) ? (1u << g) : 0u;
      invd[g] = a.lm[lc].inv_depth;
    }
    int rank[kG];
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      const unsigned long long m = __ballot((ubits >> g) & 1u);
      rank[g] = __popcll(m & below);
      if (lane == 0) s_cnt[g * 16 + wv] = __popcll(m);
    }
    __syncthreads();
    if (wv == 0) {
      int carry = 0;
#pragma unroll
      for (int base = 0; base < kG * 16; base += 64) {
        const int v = s_cnt[base + lane];
        const int inc = wave_incl_scan(v);
        s_cnt[base + lane] = carry + inc - v;
        carry += __shfl(inc, 63);
      }
      if (lane == 0) s_total = carry;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      const int l = sb + g * 1024 + tid;
      if (l < a.nl) {
        a.lm_invd_out[l] = invd[g];
        if ((ubits >> g) & 1u) { const int at = total + s_cnt[g * 16 + wv] + rank[g]; a.slot[l] = at; a.slot_lm[at] = l; a.state_invd[at] = invd[g]; }
        else a.slot[l] = -1;
      }
    }
    total += s_total;
    __syncthreads();
  }
  if (tid == 0) a.counts[3] = total;
}
__global__ __launch_bounds__(256) void k_da_emit(DaArgs a) {
  __shared__ int s_start[kDaMaxKf + 1];
  __shared__ long long s_id[kDaMaxKf];
  __shared__ int s_wave[4][3];
  const int tid = threadIdx.x, g = blockIdx.x * 256 + tid, lane = tid & 63, wv = tid >> 6;
  if (tid < a.n_kf) { s_start[tid] = a.frames[tid].start; s_id[tid] = a.frames[tid].id; }
  if (tid == 0) s_start[a.n_kf] = a.total;
  const int c = g < a.total ? a.cls[g] : 0;
  // rank of this feature among the workgroup's features of the same type, in feature order
  int rank = 0;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int t = 1; t <= 3; ++t) {
    const unsigned long long m = __ballot(c == t);
    if (c == t) rank = __popcll(m & below);
    if (lane == 0) s_wave[wv][t - 1] = __popcll(m);
  }
  __syncthreads();
  if (!c) return;
  int idx = a.wgbase[3 * blockIdx.x + (c - 1)] + rank;
  for (int w = 0; w < wv; ++w) idx += s_wave[w][c - 1];
  const int k = da_frame_of(s_start, a.n_kf, g);
  const FrameDev F = a.frames[k];
  const long long i = F.arena_off + (g - F.start);
  const int lm = a.obs_lm[i];
  const double2 ob = a.obs_xy[i];
  const LmDev L = a.lm[lm];
  if (c == 1) {
    a.tc_l[idx] = ob; a.tc_r[idx] = make_double2(L.right_ob[0], L.right_ob[1]); a.tc_lm[idx] = a.slot[lm]; a.tc_kf[idx] = k; a.tc_w[idx] = F.w_tc;
  } else if (c == 2) {
    a.tf_f[idx] = make_double2(L.right_ob[0], L.right_ob[1]); a.tf_o[idx] = ob; a.tf_lm[idx] = a.slot[lm];
    a.tf_k1[idx] = da_birth_pos(s_id, a.n_kf, L.birth_kf); a.tf_k2[idx] = k;
  } else {
    a.po_o[idx] = ob; a.po_pw[3 * idx] = L.pw[0]; a.po_pw[3 * idx + 1] = L.pw[1]; a.po_pw[3 * idx + 2] = L.pw[2]; a.po_kf[idx] = k; a.po_pi[idx] = idx;
  }
}
// after the solve: the landmarks' inverse depths back into landmark-index order (the host mirror is indexed that way)
// ... and into the resident landmark table, which the next tick's assembly reads
__global__ __launch_bounds__(256) void k_da_scatter_invd(int n_lm, const int* __restrict__ slot_lm, const double* __restrict__ state_invd, double* __restrict__ lm_invd_out,
                                                         LmDev* __restrict__ lmtab) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < n_lm) { const int l = slot_lm[s]; const double v = state_invd[s]; lm_invd_out[l] = v; lmtab[l].inv_depth = v; }
}

// the read-back mirror of k_window_unpack's plain segments: device arrays -> one contiguous staging block (then ONE device-to-host copy
// instead of five, each a submission of its own)
struct PackArgs { unsigned char* stage; int n_segs; const unsigned char* src[8]; size_t off[8]; unsigned words[8]; };
__global__ __launch_bounds__(256) void k_window_pack(PackArgs a) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k >= a.n_segs) break;
    const uint4* src = reinterpret_cast<const uint4*>(a.src[k]);
    uint4* dst = reinterpret_cast<uint4*>(a.stage + a.off[k]);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < a.words[k]; i += gridDim.x * 256) dst[i] = src[i];
  }
}

// flags[i] = 1 iff |residual_i| > max_err  (residual pairs of a PoseOnly pass with unit weights = pixel errors)
__global__ __launch_bounds__(256) void k_flag_outliers(int n, const double2* __restrict__ res, double max_err, uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double2 r = res[i];
  flags[i] = sqrt(r.x * r.x + r.y * r.y) > max_err ? 1 : 0;      // Vector2d::norm() > 10 (backend.cpp:239)
}

// ---- the tick's upload: ONE host-to-device copy of a packed staging buffer, unpacked on the device.
// The block lists used to go up as ~25 separate copies; each is a blit launch of its own (~5 us) behind ~10 us of submission latency,
// and together they kept the GPU waiting for 0.33 ms per tick.  The host now writes one 48-byte record per block —
//   TwoCamera {left ob, right ob, landmark slot, keyframe}   TwoFrame {first ob, ob, landmark slot, k1, k2}   PoseOnly {ob, pw, keyframe, table row}
// — into one pinned buffer, followed by the small plain arrays (state, IMU pre-integrations and indices, priors); k_window_unpack turns
// the records into the batches' SoA arrays and copies the plain segments to their buffers.
struct TcRec { double l[2], r[2]; int32_t lm, kf; double w; };      // w: the block's weight (two_camera_weight)
struct TfRec { double f[2], o[2]; int32_t lm, k1, k2, pad0; };
struct PoRec { double o[2], pw[3]; int32_t kf, pi; };
static_assert(sizeof(TcRec) == 48 && sizeof(TfRec) == 48 && sizeof(PoRec) == 48, "staging records are 48 bytes");
constexpr int kMaxSegs = 96;            // (the whole UnpackArgs block stays under the 4 KB kernel-argument limit)
struct UnpackArgs {
  const unsigned char* stage;
  int ntc, ntf, npo; size_t off_tc, off_tf, off_po;
  double2 *tc_l, *tc_r; int32_t *tc_lm, *tc_kf; double* tc_w;
  double2 *tf_f, *tf_o; int32_t *tf_lm, *tf_k1, *tf_k2;
  double2* po_o; double* po_pw; int32_t *po_kf, *po_pi;
  int n_segs; unsigned char* seg_dst[kMaxSegs]; size_t seg_off[kMaxSegs]; unsigned seg_words[kMaxSegs];   // 16-byte words (sizes are padded up)
  int g_tc, g_tf, g_po, g_seg;
  unsigned char* zero_dst[2]; unsigned zero_words[2];      // buffers cleared by the same launch (16-byte words)
  // changed entries of the resident landmark table: record j (64 bytes at rec_off + 64 j) goes to entry idx[j] (int32 at idx_off + 4 j)
  int n_lm_dirty; size_t lm_idx_off, lm_rec_off; unsigned char* lmtab;
};
__global__ __launch_bounds__(256) void k_window_unpack(UnpackArgs a) {
  int b = blockIdx.x;
  const int t = threadIdx.x;
  if (b < a.g_tc) {
    const int i = b * 256 + t;
    if (i < a.ntc) { const TcRec r = reinterpret_cast<const TcRec*>(a.stage + a.off_tc)[i]; a.tc_l[i] = make_double2(r.l[0], r.l[1]); a.tc_r[i] = make_double2(r.r[0], r.r[1]); a.tc_lm[i] = r.lm; a.tc_kf[i] = r.kf; if (a.tc_w) a.tc_w[i] = r.w; }
    return;
  }
  b -= a.g_tc;
  if (b < a.g_tf) {
    const int i = b * 256 + t;
    if (i < a.ntf) {
      const TfRec r = reinterpret_cast<const TfRec*>(a.stage + a.off_tf)[i];
      a.tf_f[i] = make_double2(r.f[0], r.f[1]); a.tf_o[i] = make_double2(r.o[0], r.o[1]); a.tf_lm[i] = r.lm; a.tf_k1[i] = r.k1; a.tf_k2[i] = r.k2;
    }
    return;
  }
  b -= a.g_tf;
  if (b < a.g_po) {
    const int i = b * 256 + t;
    if (i < a.npo) {
      const PoRec r = reinterpret_cast<const PoRec*>(a.stage + a.off_po)[i];
      a.po_o[i] = make_double2(r.o[0], r.o[1]); a.po_pw[3 * i] = r.pw[0]; a.po_pw[3 * i + 1] = r.pw[1]; a.po_pw[3 * i + 2] = r.pw[2]; a.po_kf[i] = r.kf; a.po_pi[i] = r.pi;
    }
    return;
  }
  b -= a.g_po;
  // plain segments: workgroup b strides over every segment (static indices only: no scratch)
#pragma unroll
  for (int k = 0; k < kMaxSegs; ++k) {
    if (k >= a.n_segs) break;
    const uint4* src = reinterpret_cast<const uint4*>(a.stage + a.seg_off[k]);
    uint4* dst = reinterpret_cast<uint4*>(a.seg_dst[k]);
    for (unsigned i = (unsigned)b * 256 + t; i < a.seg_words[k]; i += (unsigned)a.g_seg * 256) dst[i] = src[i];
  }
  {
    const int32_t* idx = reinterpret_cast<const int32_t*>(a.stage + a.lm_idx_off);
    const uint4* rec = reinterpret_cast<const uint4*>(a.stage + a.lm_rec_off);
    for (unsigned i = (unsigned)b * 256 + t; i < 4u * (unsigned)a.n_lm_dirty; i += (unsigned)a.g_seg * 256)
      reinterpret_cast<uint4*>(a.lmtab)[4 * (size_t)idx[i >> 2] + (i & 3)] = rec[i];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    uint4* dst = reinterpret_cast<uint4*>(a.zero_dst[k]);
    for (unsigned i = (unsigned)b * 256 + t; i < a.zero_words[k]; i += (unsigned)a.g_seg * 256) dst[i] = make_uint4(0, 0, 0, 0);
  }
}

}  // namespace lvf

using namespace lvf;

extern "C" {

void lvf_window_options_default(lvf_window_options* o) {
  if (!o) return;
  o->baseline = 0.537;                 // |t_cam1 - t_cam0| of the KITTI rig (config/kitti.yaml); Camera::baseline
  o->weak_visual_threshold = 20;       // backend.cpp:166
  o->prior_weight = 100.0; o->prior_v = 0.0;   // backend.cpp:170,175
  o->device_assembly = 1;
}

int lvf_window_create(lvf_ctx* ctx, const lvf_camera* left, const lvf_camera* right, const lvf_window_options* opt, lvf_window** out) {
  LVF_REQUIRE(ctx && left && right && out, "lvf_window_create: null argument");
  auto* w = new lvf_window();
  w->ctx = ctx; w->left = *left; w->right = *right;
  if (opt) w->opt = *opt; else lvf_window_options_default(&w->opt);
  *out = w;
  return LVF_OK;
}
int lvf_window_destroy(lvf_window* w) { delete w; return LVF_OK; }

int lvf_window_add_keyframe(lvf_window* w, int64_t kf_id, const double* pose7, double w_visual) {
  LVF_REQUIRE(w && pose7, "lvf_window_add_keyframe: null argument");
  LVF_REQUIRE(w->kfs.empty() || kf_id > w->kfs.back().id, "lvf_window_add_keyframe: keyframe ids must increase (got %lld after %lld)", (long long)kf_id,
              (long long)(w->kfs.empty() ? 0 : w->kfs.back().id));
  lvf_window::Kf k;
  k.id = kf_id; std::memcpy(k.pose, pose7, 56); k.w_visual = w_visual;
  w->kf_index[kf_id] = (int)w->kfs.size();
  w->kfs.push_back(std::move(k));
  return LVF_OK;
}
int lvf_window_set_imu(lvf_window* w, int64_t kf_id, const double* vel3, const double* ba3, const double* bg3, const lvf_preint* pre) {
  LVF_REQUIRE(w && vel3 && ba3 && bg3, "lvf_window_set_imu: null argument");
  auto it = w->kf_index.find(kf_id);
  LVF_REQUIRE(it != w->kf_index.end(), "lvf_window_set_imu: keyframe %lld is not in the window", (long long)kf_id);
  lvf_window::Kf& k = w->kfs[it->second];
  std::memcpy(k.vel, vel3, 24); std::memcpy(k.ba, ba3, 24); std::memcpy(k.bg, bg3, 24);
  const bool had = k.has_pre;
  k.good_imu = true; k.has_pre = pre != nullptr;
  if (pre && !(had && std::memcmp(&k.pre, pre, sizeof(lvf_preint)) == 0)) { k.pre = *pre; k.pre_ver = ++w->pre_counter; }      // (an unchanged pre-integration keeps its cached matrix)
  return LVF_OK;
}
int lvf_window_add_landmark(lvf_window* w, int64_t lm_id, int64_t birth_kf_id, const double* left_ob2, const double* right_ob2, double inv_depth) {
  LVF_REQUIRE(w && left_ob2 && right_ob2, "lvf_window_add_landmark: null argument");
  LVF_REQUIRE(!w->lm_index.count(lm_id), "lvf_window_add_landmark: landmark %lld exists", (long long)lm_id);
  auto it = w->kf_index.find(birth_kf_id);
  LVF_REQUIRE(it != w->kf_index.end(), "lvf_window_add_landmark: birth keyframe %lld is not in the window", (long long)birth_kf_id);
  LVF_REQUIRE(inv_depth != 0.0, "lvf_window_add_landmark: zero inverse depth");
  lvf_window::Lm l;
  l.id = lm_id; l.birth_kf = birth_kf_id; l.inv_depth = inv_depth;
  std::memcpy(l.left_ob, left_ob2, 16); std::memcpy(l.right_ob, right_ob2, 16);
  int idx;
  if (!w->lm_free.empty()) { idx = w->lm_free.back(); w->lm_free.pop_back(); w->lms[idx] = l; }
  else { idx = (int)w->lms.size(); w->lms.push_back(l); }
  w->mark_lm(idx);
  w->lm_index[lm_id] = idx;
  ++w->n_lm_live;
  // the landmark's own left feature in its birth frame (features_left[lm] of the first frame -> the TwoCamera block)
  lvf_window::Obs o; o.lm_id = lm_id; o.lm = idx; o.ob[0] = left_ob2[0]; o.ob[1] = left_ob2[1];
  w->kfs[it->second].put(o);
  return LVF_OK;
}
int lvf_window_add_observation(lvf_window* w, int64_t lm_id, int64_t kf_id, const double* ob2) {
  LVF_REQUIRE(w && ob2, "lvf_window_add_observation: null argument");
  auto il = w->lm_index.find(lm_id);
  LVF_REQUIRE(il != w->lm_index.end(), "lvf_window_add_observation: unknown landmark %lld", (long long)lm_id);
  auto ik = w->kf_index.find(kf_id);
  LVF_REQUIRE(ik != w->kf_index.end(), "lvf_window_add_observation: keyframe %lld is not in the window", (long long)kf_id);
  LVF_REQUIRE(w->lms[il->second].birth_kf < kf_id, "lvf_window_add_observation: observation in or before the birth frame");
  lvf_window::Obs o; o.lm_id = lm_id; o.lm = il->second; o.ob[0] = ob2[0]; o.ob[1] = ob2[1];
  w->kfs[ik->second].put(o);
  return LVF_OK;
}
int lvf_window_remove_observation(lvf_window* w, int64_t lm_id, int64_t kf_id) {
  LVF_REQUIRE(w, "lvf_window_remove_observation: null window");
  auto ik = w->kf_index.find(kf_id);
  LVF_REQUIRE(ik != w->kf_index.end(), "lvf_window_remove_observation: keyframe %lld is not in the window", (long long)kf_id);
  auto& v = w->kfs[ik->second].obs;
  v.erase(std::remove_if(v.begin(), v.end(), [&](const lvf_window::Obs& o) { return o.lm_id == lm_id; }), v.end());
  w->kfs[ik->second].d_dirty = true;
  return LVF_OK;
}

// Map::GetKeyFrames(finished) (src/map.cpp:49-55): frames older than first_active_kf_id leave the problem.  Landmarks born in a
// departing frame keep their world point at the departing frame's final pose and their last inverse depth (they become
// PoseOnly blocks, backend.cpp:126-131).
int lvf_window_slide(lvf_window* w, int64_t first_active_kf_id) {
  LVF_REQUIRE(w, "lvf_window_slide: null window");
  size_t drop = 0;
  while (drop < w->kfs.size() && w->kfs[drop].id < first_active_kf_id) ++drop;
  if (drop == 0) return LVF_OK;
  for (size_t k = 0; k < drop; ++k) {
    std::array<double, 7> p;
    std::memcpy(p.data(), w->kfs[k].pose, 56);
    w->departed[w->kfs[k].id] = p;
  }
  for (size_t i = 0; i < w->lms.size(); ++i) {
    lvf_window::Lm& l = w->lms[i];
    if (l.alive && !l.fixed && l.birth_kf < first_active_kf_id) {
      auto it = w->departed.find(l.birth_kf);
      if (it != w->departed.end()) { to_world(w->right, l.right_ob, l.inv_depth, it->second.data(), l.pw); l.fixed = true; w->mark_lm((int)i); }
    }
  }
  w->kfs.erase(w->kfs.begin(), w->kfs.begin() + drop);
  w->kf_index.clear();
  for (size_t k = 0; k < w->kfs.size(); ++k) w->kf_index[w->kfs[k].id] = (int)k;
  prune(w);
  return LVF_OK;
}

int lvf_window_set_pose(lvf_window* w, int64_t kf_id, const double* pose7) {
  LVF_REQUIRE(w && pose7, "lvf_window_set_pose: null argument");
  auto ik = w->kf_index.find(kf_id);
  LVF_REQUIRE(ik != w->kf_index.end(), "lvf_window_set_pose: keyframe %lld is not in the window", (long long)kf_id);
  std::memcpy(w->kfs[ik->second].pose, pose7, 56);
  return LVF_OK;
}
int lvf_window_get_pose(const lvf_window* w, int64_t kf_id, double* pose7) {
  LVF_REQUIRE(w && pose7, "lvf_window_get_pose: null argument");
  auto ik = w->kf_index.find(kf_id);
  if (ik != w->kf_index.end()) { std::memcpy(pose7, w->kfs[ik->second].pose, 56); return LVF_OK; }
  auto id = w->departed.find(kf_id);
  LVF_REQUIRE(id != w->departed.end(), "lvf_window_get_pose: unknown keyframe %lld", (long long)kf_id);
  std::memcpy(pose7, id->second.data(), 56);
  return LVF_OK;
}
int lvf_window_get_imu(const lvf_window* w, int64_t kf_id, double* vel3, double* ba3, double* bg3) {
  LVF_REQUIRE(w, "lvf_window_get_imu: null window");
  auto ik = w->kf_index.find(kf_id);
  LVF_REQUIRE(ik != w->kf_index.end(), "lvf_window_get_imu: keyframe %lld is not in the window", (long long)kf_id);
  const lvf_window::Kf& k = w->kfs[ik->second];
  if (vel3) std::memcpy(vel3, k.vel, 24);
  if (ba3) std::memcpy(ba3, k.ba, 24);
  if (bg3) std::memcpy(bg3, k.bg, 24);
  return LVF_OK;
}
int lvf_window_get_inv_depth(const lvf_window* w, int64_t lm_id, double* inv_depth) {
  LVF_REQUIRE(w && inv_depth, "lvf_window_get_inv_depth: null argument");
  auto il = w->lm_index.find(lm_id);
  LVF_REQUIRE(il != w->lm_index.end(), "lvf_window_get_inv_depth: unknown landmark %lld", (long long)lm_id);
  *inv_depth = w->lms[il->second].inv_depth;
  return LVF_OK;
}
int lvf_window_counts(const lvf_window* w, int32_t* counts8) {
  LVF_REQUIRE(w && counts8, "lvf_window_counts: null argument");
  counts8[0] = (int)w->kfs.size(); counts8[1] = w->n_lm_problem; counts8[2] = w->n_tc; counts8[3] = w->n_tf; counts8[4] = w->n_po;
  counts8[5] = w->n_imu; counts8[6] = w->n_prior; counts8[7] = w->n_lm_live;
  return LVF_OK;
}


// lvf_window_solve with the block lists assembled on the device (see "device-side assembly" above).
}  // extern "C"
static int window_solve_device(lvf_window* w, const lvf_solver_options* o, lvf_solver_summary* summary) {
  lvf_ctx* ctx = w->ctx;
  hipStream_t s = ctx->stream;
  const int n_kf = (int)w->kfs.size();
  static const bool timing = getenv("LVF_WINDOW_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const auto t_begin = now();
  const size_t nl = w->lms.size();
  // ---- persistent device objects
  if (!w->st) {
    LVF_TRY(lvf_state_create(ctx, 0, 0, &w->st));
    const double z2[2] = {0, 0}; const int32_t z = 0; const double id7[7] = {0, 0, 0, 1, 0, 0, 0};
    LVF_TRY(lvf_two_camera_create(ctx, &w->left, &w->right, 0, z2, z2, &z, &z, &w->tc));
    LVF_TRY(lvf_two_frame_create(ctx, &w->left, &w->right, 0, z2, z2, &z, &z, &z, &w->tf));
    LVF_TRY(lvf_pose_only_create(ctx, &w->left, 0, z2, &z, &z, 0, id7, &w->po));
    LVF_TRY(lvf_imu_create(ctx, 0, nullptr, nullptr, nullptr, &w->imu));
    LVF_TRY(lvf_pose_prior_create(ctx, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &w->prior));
  }
  lvf_state* st = w->st;
  // ---- feature arena: every frame owns a segment; new / changed frames are (re)sent, a frame that outgrew its segment moves to the
  // end, and when the end is reached everything is laid out again from the start
  size_t n_obs = 0, need_new = 0;
  for (lvf_window::Kf& f : w->kfs) {
    f.sort_unique();
    n_obs += f.obs.size();
    if (f.d_cap < f.obs.size()) { f.d_dirty = true; need_new += f.obs.size() + f.obs.size() / 4 + 64; }
  }
  LVF_REQUIRE(n_obs < (size_t)1 << 30, "lvf_window_solve: too many features");
  if (w->arena_end + need_new > w->d_obs_lm.cap) {       // re-layout (also the first tick): capacity for twice the window
    size_t total_cap = 0;
    for (const lvf_window::Kf& f : w->kfs) total_cap += f.obs.size() + f.obs.size() / 4 + 64;
    const size_t want = std::max<size_t>(2 * total_cap, 4096);
    if (want > w->d_obs_lm.cap) { LVF_TRY(w->d_obs_lm.ensure(want)); LVF_TRY(w->d_obs_xy.ensure(2 * want)); }
    w->arena_end = 0;
    for (lvf_window::Kf& f : w->kfs) { f.d_cap = 0; f.d_dirty = true; }
  }
  for (lvf_window::Kf& f : w->kfs)
    if (f.d_cap < f.obs.size()) { f.d_off = w->arena_end; f.d_cap = f.obs.size() + f.obs.size() / 4 + 64; w->arena_end += f.d_cap; }
  // ---- staging: frame table | landmark table | state | dirty feature segments (one copy, one unpack launch)
  size_t dirty_obs = 0; int dirty_frames = 0;
  for (const lvf_window::Kf& f : w->kfs) if (f.d_dirty) { dirty_obs += f.obs.size(); ++dirty_frames; }
  const bool direct_segments = 2 * dirty_frames + 16 > kMaxSegs;      // (a full re-layout of a long window: copy those segments one by one)
  const size_t stage_bound = up16((size_t)n_kf * sizeof(FrameDev)) + up16(nl * sizeof(LmDev)) + up16(nl * 4) + 64 + up16((size_t)17 * n_kf * 8) + 8 * 64 + dirty_obs * 24 + (size_t)dirty_frames * 64 + 4096 +
                             up16((size_t)n_kf * 467 * 8) + (size_t)n_kf * 16 + 64;
  LVF_TRY(w->h_stage.reserve(stage_bound));
  LVF_TRY(w->d_stage.ensure(stage_bound));
  unsigned char* hs = w->h_stage.p;
  UnpackArgs ua{};
  size_t cur = 0;
  auto seg = [&](void* dst, size_t bytes) -> unsigned char* {
    unsigned char* at = hs + cur;
    if (bytes) { ua.seg_dst[ua.n_segs] = static_cast<unsigned char*>(dst); ua.seg_off[ua.n_segs] = cur; ua.seg_words[ua.n_segs] = (unsigned)((bytes + 15) / 16); ++ua.n_segs; }
    cur += up16(bytes);
    return at;
  };
  const unsigned char* lmtab_before = w->d_lmtab.p;
  LVF_TRY(w->d_frames.ensure((size_t)n_kf * sizeof(FrameDev) + 16)); LVF_TRY(w->d_lmtab.ensure(nl * sizeof(LmDev) + 16));
  if (w->d_lmtab.p != lmtab_before) w->lmtab_valid = false;          // re-allocated: the resident table is gone
  // frame table (also what the IMU / prior decisions below need of a frame)
  std::vector<double> Rk((size_t)9 * n_kf + 9);
  landmarks_to_world_prepare(w, Rk.data());
  double inv_e[7];
  hse3::inv(w->left.extrinsic, inv_e);
  FrameDev* fd = reinterpret_cast<FrameDev*>(seg(w->d_frames.p, (size_t)n_kf * sizeof(FrameDev)));
  {
    int start = 0;
    for (int k = 0; k < n_kf; ++k) {
      const lvf_window::Kf& f = w->kfs[k];
      FrameDev& d = fd[k];
      d.id = f.id; d.start = start; d.cnt = (int)f.obs.size(); d.arena_off = (long long)f.d_off;
      start += d.cnt;
      double inv_pose[7];
      hse3::inv(f.pose, inv_pose);
      double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1}, c0[3], c1[3], c2[3], t[3], r[3];
      hse3::rotate(inv_pose, ex, r); hse3::rotate(inv_e, r, c0);
      hse3::rotate(inv_pose, ey, r); hse3::rotate(inv_e, r, c1);
      hse3::rotate(inv_pose, ez, r); hse3::rotate(inv_e, r, c2);
      d.zrow[0] = c0[2]; d.zrow[1] = c1[2]; d.zrow[2] = c2[2];
      hse3::rotate(inv_e, inv_pose + 4, t);
      d.zoff = t[2] + inv_e[6];
      std::memcpy(d.R, &Rk[(size_t)9 * k], 72); std::memcpy(d.t, f.pose + 4, 24);
      d.w_tc = two_camera_weight(f.w_visual);
    }
  }
  // landmark table: all of it after a (re)allocation or a tick on the host path, otherwise only what add_landmark / slide / prune touched
  {
    auto fill = [&](LmDev& d, const lvf_window::Lm& l) {
      d.right_ob[0] = l.right_ob[0]; d.right_ob[1] = l.right_ob[1]; d.pw[0] = l.pw[0]; d.pw[1] = l.pw[1]; d.pw[2] = l.pw[2];
      d.inv_depth = l.inv_depth; d.birth_kf = l.alive ? (long long)l.birth_kf : -1; d.fixed = (l.alive && l.fixed) ? 1 : 0; d.alive = l.alive ? 1 : 0;
    };
    const bool resident = w->lmtab_valid;
    w->lmtab_valid = false;                    // (set again once this tick has gone through: an error below leaves the table to be re-sent)
    if (!resident) {
      LmDev* ld = reinterpret_cast<LmDev*>(seg(w->d_lmtab.p, nl * sizeof(LmDev)));
      for (size_t i = 0; i < nl; ++i) fill(ld[i], w->lms[i]);
    } else if (!w->lm_dirty.empty()) {
      const size_t nd = w->lm_dirty.size();
      ua.n_lm_dirty = (int)nd; ua.lmtab = w->d_lmtab.p;
      ua.lm_idx_off = cur;
      int32_t* di = reinterpret_cast<int32_t*>(hs + cur); cur += up16(nd * 4);
      ua.lm_rec_off = cur;
      LmDev* dr = reinterpret_cast<LmDev*>(hs + cur); cur += nd * sizeof(LmDev);
      for (size_t j = 0; j < nd; ++j) { di[j] = w->lm_dirty[j]; fill(dr[j], w->lms[w->lm_dirty[j]]); }
    }
    for (int i : w->lm_dirty) w->lm_dirty_flag[i] = 0;
    w->lm_dirty.clear();
  }
  // state (the inverse depths are filled on the device, by slot)
  st->n_kf = n_kf;
  LVF_TRY(st->poses.ensure((size_t)7 * n_kf + 2)); LVF_TRY(st->vel.ensure((size_t)3 * n_kf + 2)); LVF_TRY(st->ba.ensure((size_t)3 * n_kf + 2)); LVF_TRY(st->bg.ensure((size_t)3 * n_kf + 2));
  LVF_TRY(st->w_visual.ensure((size_t)n_kf + 2)); LVF_TRY(st->inv_depth.ensure(nl + 2));
  st->poses.n = (size_t)7 * n_kf; st->vel.n = st->ba.n = st->bg.n = (size_t)3 * n_kf; st->w_visual.n = n_kf;
  {
    double* sp = reinterpret_cast<double*>(seg(st->poses.p, (size_t)7 * n_kf * 8));
    double* sv = reinterpret_cast<double*>(seg(st->vel.p, (size_t)3 * n_kf * 8));
    double* sa = reinterpret_cast<double*>(seg(st->ba.p, (size_t)3 * n_kf * 8));
    double* sg = reinterpret_cast<double*>(seg(st->bg.p, (size_t)3 * n_kf * 8));
    double* sw = reinterpret_cast<double*>(seg(st->w_visual.p, (size_t)n_kf * 8));
    for (int k = 0; k < n_kf; ++k) {
      const lvf_window::Kf& f = w->kfs[k];
      std::memcpy(&sp[(size_t)7 * k], f.pose, 56); std::memcpy(&sv[(size_t)3 * k], f.vel, 24); std::memcpy(&sa[(size_t)3 * k], f.ba, 24);
      std::memcpy(&sg[(size_t)3 * k], f.bg, 24); sw[k] = f.w_visual;
    }
  }
  // ImuError blocks (backend.cpp:141-153: both frames good, a pre-integration on the later one) do not depend on the assembly: their
  // pre-integrations ride in this upload and their information matrices are factored while the host waits for the assembly's counters
  std::vector<int32_t> imu_i, imu_j;
  {
    for (int k = 1; k < n_kf; ++k)
      if (w->kfs[k].good_imu && w->kfs[k - 1].good_imu && w->kfs[k].has_pre) { imu_i.push_back(k - 1); imu_j.push_back(k); }
    lvf_batch* b = w->imu;
    const size_t ni = imu_i.size();
    LVF_TRY(b->pre.ensure(467 * ni + 2)); LVF_TRY(b->idx_a.ensure(ni + 4)); LVF_TRY(b->idx_b.ensure(ni + 4));
    b->pre.n = 467 * ni; b->idx_a.n = b->idx_b.n = ni;
    if (ni) {
      static_assert(sizeof(lvf_preint) == 467 * 8, "lvf_preint is 467 doubles");
      lvf_preint* dst = reinterpret_cast<lvf_preint*>(seg(b->pre.p, 467 * ni * 8));
      for (size_t f = 0; f < ni; ++f) dst[f] = w->kfs[imu_j[f]].pre;
      std::memcpy(seg(b->idx_a.p, ni * 4), imu_i.data(), ni * 4); std::memcpy(seg(b->idx_b.p, ni * 4), imu_j.data(), ni * 4);
      // where each factor's information matrix sits in what the previous tick left (-1: not there, factor it)
      LVF_TRY(w->d_sq_src.ensure(ni + 4));
      int32_t* src = reinterpret_cast<int32_t*>(seg(w->d_sq_src.p, ni * 4));
      std::vector<unsigned> ver(ni);
      if (!w->sq_valid) w->sq_ver.clear();            // (a tick that failed after the swap below, or the host-assembly path, left nothing to reuse)
      for (size_t f = 0; f < ni; ++f) {
        ver[f] = w->kfs[imu_j[f]].pre_ver;
        src[f] = -1;
        for (size_t g = 0; g < w->sq_ver.size(); ++g) if (w->sq_ver[g] == ver[f]) { src[f] = (int32_t)g; break; }
      }
      b->sqrt_info.swap(w->sq_prev);                 // last tick's matrices become the source; the batch gets the other buffer
      w->sq_ver = std::move(ver);
      w->sq_valid = false;                           // until this tick's matrices are on their way (below)
    } else w->sq_ver.clear();
    b->host_kf1 = imu_i; b->host_kf2 = imu_j;
    LVF_TRY(b->sqrt_info.ensure((size_t)225 * ni)); LVF_TRY(b->res.ensure((size_t)15 * ni));
    for (int q = 0; q < 8; ++q) LVF_TRY(b->jac[q].ensure((size_t)15 * b->block_size[q] * ni));
    b->n = (int)ni; b->min_n_kf = n_kf; b->min_n_lm = 0; b->evaluated = false;
    w->n_imu = (int)ni;
  }
  // dirty feature segments
  for (lvf_window::Kf& f : w->kfs) {
    if (!f.d_dirty) continue;
    const size_t c = f.obs.size();
    if (c) {
      int32_t* dl; double* dx;
      std::vector<int32_t> tl; std::vector<double> tx;
      if (direct_segments) { tl.resize(c); tx.resize(2 * c); dl = tl.data(); dx = tx.data(); }
      else { dl = reinterpret_cast<int32_t*>(seg(w->d_obs_lm.p + f.d_off, c * 4)); dx = reinterpret_cast<double*>(seg(w->d_obs_xy.p + 2 * f.d_off, c * 16)); }
      for (size_t j = 0; j < c; ++j) { dl[j] = f.obs[j].lm; dx[2 * j] = f.obs[j].ob[0]; dx[2 * j + 1] = f.obs[j].ob[1]; }
      if (direct_segments) {
        LVF_HIP(hipMemcpyAsync(w->d_obs_lm.p + f.d_off, dl, c * 4, hipMemcpyHostToDevice, s));
        LVF_HIP(hipMemcpyAsync(w->d_obs_xy.p + 2 * f.d_off, dx, c * 16, hipMemcpyHostToDevice, s));
        LVF_HIP(hipStreamSynchronize(s));                 // (pageable temporaries)
      }
    }
    f.d_dirty = false;
  }
  LVF_REQUIRE(cur <= w->h_stage.cap && ua.n_segs <= kMaxSegs, "lvf_window_solve: staging overflow");
  const auto t_staged = now();
  LVF_HIP(hipMemcpyAsync(w->d_stage.p, hs, cur, hipMemcpyHostToDevice, s));
  ua.stage = w->d_stage.p; ua.g_seg = 64;
  // (the assembly's `used` flags and counters are cleared by the unpack launch: two fill launches less)
  LVF_TRY(w->d_used.ensure(nl + 32)); LVF_TRY(w->d_counts.ensure(4 + 2 * kDaMaxKf));
  ua.zero_dst[0] = w->d_used.p; ua.zero_words[0] = (unsigned)((nl + 16 + 15) / 16);
  ua.zero_dst[1] = reinterpret_cast<unsigned char*>(w->d_counts.p); ua.zero_words[1] = (unsigned)(((4 + 2 * kDaMaxKf) * sizeof(int) + 15) / 16);
  hipLaunchKernelGGL(k_window_unpack, dim3(ua.g_seg), dim3(256), 0, s, ua);
  // ---- assembly kernels
  const int n_wg = (int)((n_obs + 255) / 256);
  LVF_TRY(w->d_cls.ensure(n_obs + 16)); LVF_TRY(w->d_wgcnt.ensure((size_t)3 * n_wg + 4)); LVF_TRY(w->d_wgbase.ensure((size_t)3 * n_wg + 4));
  LVF_TRY(w->d_slot.ensure(nl + 4)); LVF_TRY(w->d_slot_lm.ensure(nl + 4)); LVF_TRY(w->d_lm_invd_out.ensure(nl + 4));
  LVF_TRY(w->h_counts.reserve(4 + 2 * kDaMaxKf));
  LVF_TRY(w->tc->ob_a.ensure(2 * std::min(n_obs, nl) + 2)); LVF_TRY(w->tc->ob_b.ensure(2 * std::min(n_obs, nl) + 2)); LVF_TRY(w->tc->idx_a.ensure(std::min(n_obs, nl) + 2)); LVF_TRY(w->tc->idx_b.ensure(std::min(n_obs, nl) + 2)); LVF_TRY(w->tc->wblk.ensure(std::min(n_obs, nl) + 2));
  LVF_TRY(w->tf->ob_a.ensure(2 * n_obs + 2)); LVF_TRY(w->tf->ob_b.ensure(2 * n_obs + 2)); LVF_TRY(w->tf->idx_a.ensure(n_obs + 2)); LVF_TRY(w->tf->idx_b.ensure(n_obs + 2)); LVF_TRY(w->tf->idx_c.ensure(n_obs + 2));
  LVF_TRY(w->po->ob_a.ensure(2 * n_obs + 2)); LVF_TRY(w->po->idx_a.ensure(n_obs + 2)); LVF_TRY(w->po->idx_b.ensure(n_obs + 2)); LVF_TRY(w->po->table.ensure(3 * n_obs + 2));
  DaArgs da{};
  da.n_kf = n_kf; da.total = (int)n_obs; da.nl = (int)nl; da.n_wg = n_wg;
  da.frames = reinterpret_cast<const FrameDev*>(w->d_frames.p); da.obs_lm = w->d_obs_lm.p; da.obs_xy = reinterpret_cast<const double2*>(w->d_obs_xy.p);
  da.lm = reinterpret_cast<const LmDev*>(w->d_lmtab.p);
  std::memcpy(da.Re, &Rk[(size_t)9 * n_kf], 72); std::memcpy(da.te, w->right.extrinsic + 4, 24);
  da.fx = w->right.fx; da.fy = w->right.fy; da.cx = w->right.cx; da.cy = w->right.cy; da.far_z = w->opt.baseline * 50.0;
  da.cls = w->d_cls.p; da.used = w->d_used.p; da.wgcnt = w->d_wgcnt.p; da.wgbase = w->d_wgbase.p; da.slot = w->d_slot.p; da.slot_lm = w->d_slot_lm.p; da.counts = w->d_counts.p;
  da.state_invd = st->inv_depth.p; da.lm_invd_out = w->d_lm_invd_out.p;
  da.tc_l = reinterpret_cast<double2*>(w->tc->ob_a.p); da.tc_r = reinterpret_cast<double2*>(w->tc->ob_b.p); da.tc_lm = w->tc->idx_a.p; da.tc_kf = w->tc->idx_b.p; da.tc_w = w->tc->wblk.p;
  da.tf_f = reinterpret_cast<double2*>(w->tf->ob_a.p); da.tf_o = reinterpret_cast<double2*>(w->tf->ob_b.p); da.tf_lm = w->tf->idx_a.p; da.tf_k1 = w->tf->idx_b.p; da.tf_k2 = w->tf->idx_c.p;
  da.po_o = reinterpret_cast<double2*>(w->po->ob_a.p); da.po_pw = w->po->table.p; da.po_kf = w->po->idx_a.p; da.po_pi = w->po->idx_b.p;
  if (n_wg) hipLaunchKernelGGL(k_da_classify, dim3(n_wg), dim3(256), 0, s, da);
  hipLaunchKernelGGL(k_da_scan, dim3(1), dim3(1024), 0, s, da);
  if (n_wg) hipLaunchKernelGGL(k_da_emit, dim3(n_wg), dim3(256), 0, s, da);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(w->h_counts.p, w->d_counts.p, (4 + 2 * kDaMaxKf) * sizeof(int), hipMemcpyDeviceToHost, s));
  if (!w->ev_counts) LVF_HIP(hipEventCreateWithFlags(&w->ev_counts, hipEventDisableTiming));
  LVF_HIP(hipEventRecord(w->ev_counts, s));
  if (w->n_imu) { LVF_TRY(launch_imu_sqrt_info_cached(w->imu, w->d_sq_src.p, w->sq_prev.p)); w->sq_valid = true; }      // runs while the host reads the counters and configures the problem
  LVF_HIP(hipEventSynchronize(w->ev_counts));
  const int* hc = w->h_counts.p;
  const size_t ntc = (size_t)hc[0], ntf = (size_t)hc[1], npo = (size_t)hc[2];
  const int n_lm = hc[3];
  const auto t_assembled = now();
  // ---- weak-constraint priors (backend.cpp:164-178): frames without an ImuError block whose near-feature count is below the threshold
  std::vector<double> pr_t, pr_w, pr_v;
  std::vector<int32_t> pr_a, pr_b, kf2_counts(n_kf, 0);
  {
    size_t fi = 0;                                   // imu_j is ascending
    for (int k = 0; k < n_kf; ++k) {
      lvf_window::Kf& f = w->kfs[k];
      kf2_counts[k] = hc[4 + k];
      const bool imu_block = fi < imu_j.size() && imu_j[fi] == k;
      if (imu_block) ++fi;
      if (!imu_block && hc[4 + kDaMaxKf + k] < w->opt.weak_visual_threshold) {
        double t7[7] = {0, 0, 0, 0, 0, 0, 0};
        if (k > 0) { LVF_TRY(lvf_relative_rpyxyz(w->kfs[k - 1].pose, f.pose, t7)); pr_a.push_back(k - 1); }
        else { std::memcpy(t7, f.pose, 56); pr_a.push_back(-1); }
        pr_b.push_back(k); pr_t.insert(pr_t.end(), t7, t7 + 7); pr_w.push_back(w->opt.prior_weight); pr_v.push_back(w->opt.prior_v);
      }
    }
  }
  w->n_lm_problem = n_lm;
  w->n_tc = (int)ntc; w->n_tf = (int)ntf; w->n_po = (int)npo; w->n_prior = (int)pr_b.size();
  st->n_lm = n_lm; st->inv_depth.n = n_lm;
  auto idx_ok = [](lvf_batch* b, int n, int nkf, int nlm) { b->n = n; b->min_n_kf = nkf; b->min_n_lm = nlm; b->evaluated = false; };
  w->tc->ob_a.n = w->tc->ob_b.n = 2 * ntc; w->tc->idx_a.n = w->tc->idx_b.n = w->tc->wblk.n = ntc;
  idx_ok(w->tc, w->n_tc, n_kf, n_lm);
  w->tf->ob_a.n = w->tf->ob_b.n = 2 * ntf; w->tf->idx_a.n = w->tf->idx_b.n = w->tf->idx_c.n = ntf;
  idx_ok(w->tf, w->n_tf, n_kf, n_lm);
  w->tf->sorted_by_kf = true; w->tf->host_kf1.clear(); w->tf->host_kf2.clear(); w->tf->host_lm.clear(); w->tf->unique_lk2_known = true; w->tf->kf2_counts = std::move(kf2_counts);
  w->po->ob_a.n = 2 * npo; w->po->idx_a.n = w->po->idx_b.n = npo; w->po->table.n = 3 * npo;
  idx_ok(w->po, w->n_po, n_kf, 0); w->po->n_table = w->n_po; w->po->sorted_by_kf = true;
  // second (small) staged upload: the priors
  {
    UnpackArgs ub{};
    size_t at = 0;
    auto seg2 = [&](void* dst, size_t bytes) -> unsigned char* {
      unsigned char* pp = hs + at;
      if (bytes) { ub.seg_dst[ub.n_segs] = static_cast<unsigned char*>(dst); ub.seg_off[ub.n_segs] = at; ub.seg_words[ub.n_segs] = (unsigned)((bytes + 15) / 16); ++ub.n_segs; }
      at += up16(bytes);
      return pp;
    };
    LVF_TRY(w->h_stage.reserve(16 * 64 + (size_t)n_kf * 128 + 4096));      // (grow-only: the first copy above has completed)
    hs = w->h_stage.p;
    lvf_batch* b = w->prior;
    const size_t np_ = (size_t)w->n_prior;
    LVF_TRY(b->idx_a.ensure(np_ + 4)); LVF_TRY(b->idx_b.ensure(np_ + 4)); LVF_TRY(b->table.ensure(7 * np_ + 2)); LVF_TRY(b->ob_a.ensure(np_ + 2)); LVF_TRY(b->ob_b.ensure(np_ + 2));
    b->idx_a.n = b->idx_b.n = np_; b->table.n = 7 * np_; b->ob_a.n = b->ob_b.n = np_;
    if (np_) {
      std::memcpy(seg2(b->idx_a.p, np_ * 4), pr_a.data(), np_ * 4); std::memcpy(seg2(b->idx_b.p, np_ * 4), pr_b.data(), np_ * 4);
      std::memcpy(seg2(b->table.p, 7 * np_ * 8), pr_t.data(), 7 * np_ * 8);
      std::memcpy(seg2(b->ob_a.p, np_ * 8), pr_w.data(), np_ * 8); std::memcpy(seg2(b->ob_b.p, np_ * 8), pr_v.data(), np_ * 8);
    }
    LVF_TRY(b->res.ensure((size_t)6 * w->n_prior)); LVF_TRY(b->jac[0].ensure((size_t)42 * w->n_prior)); LVF_TRY(b->jac[1].ensure((size_t)42 * w->n_prior));
    idx_ok(b, w->n_prior, n_kf, 0);
    b->host_kf1 = pr_a; b->host_kf2 = pr_b;
    if (ub.n_segs) {
      LVF_TRY(w->d_stage.ensure(at + 16));
      LVF_HIP(hipMemcpyAsync(w->d_stage.p, hs, at, hipMemcpyHostToDevice, s));
      ub.stage = w->d_stage.p; ub.g_seg = 8;
      hipLaunchKernelGGL(k_window_unpack, dim3(ub.g_seg), dim3(256), 0, s, ub);
      LVF_HIP(hipGetLastError());
    }
  }
  const auto t_uploaded = now();
  if (!w->prob) {
    LVF_TRY(lvf_problem_create(ctx, st, w->tc, w->tf, w->po, w->imu, &w->prob));
    LVF_TRY(lvf_problem_set_pose_priors(w->prob, w->prior));
  } else {
    LVF_TRY(problem_configure(w->prob));
  }
  const auto t_configured = now();
  // ---- solve, with the read-back (poses, velocities, biases by keyframe position; inverse depths in landmark-index order) enqueued
  // behind the last iteration and ahead of the wait that ends the solve: ONE stream wait per tick for both
  size_t offs[5];
  struct PackCtx { lvf_window* w; lvf_state* st; hipStream_t s; int n_kf, n_lm; size_t nl; size_t* offs; };
  PackCtx pc{w, st, s, n_kf, n_lm, nl, offs};
  auto pack_tail = [](void* u) -> int {
    PackCtx& c = *static_cast<PackCtx*>(u);
    lvf_window* w = c.w; lvf_state* st = c.st; hipStream_t s = c.s;
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    if (c.n_lm) hipLaunchKernelGGL(k_da_scatter_invd, dim3((c.n_lm + 255) / 256), dim3(256), 0, s, c.n_lm, w->d_slot_lm.p, st->inv_depth.p, w->d_lm_invd_out.p, reinterpret_cast<LmDev*>(w->d_lmtab.p));
    PackArgs pa{};
    size_t at = 0;
    auto segp = [&](const void* src, size_t bytes, int slot) {
      c.offs[slot] = at;
      if (bytes) { pa.src[pa.n_segs] = static_cast<const unsigned char*>(src); pa.off[pa.n_segs] = at; pa.words[pa.n_segs] = (unsigned)((bytes + 15) / 16); ++pa.n_segs; }
      at += up16(bytes);
    };
    segp(st->poses.p, (size_t)7 * c.n_kf * 8, 0); segp(st->vel.p, (size_t)3 * c.n_kf * 8, 1); segp(st->ba.p, (size_t)3 * c.n_kf * 8, 2); segp(st->bg.p, (size_t)3 * c.n_kf * 8, 3);
    segp(w->d_lm_invd_out.p, c.nl * 8, 4);
    LVF_TRY(w->d_stage.ensure(at + 16)); LVF_TRY(w->h_state.reserve(at / 8 + 2));
    pa.stage = w->d_stage.p;
    hipLaunchKernelGGL(k_window_pack, dim3(32), dim3(256), 0, s, pa);
    LVF_HIP(hipGetLastError());
    LVF_HIP(hipMemcpyAsync(w->h_state.p, w->d_stage.p, at, hipMemcpyDeviceToHost, s));
    return LVF_OK;
  };
  LVF_TRY(lvf_problem_solve_then(w->prob, o, summary, pack_tail, &pc));      // (the wait for the control block also covers the copy above: same stream, enqueued first)
  if (o->max_num_iterations <= 0) LVF_HIP(hipStreamSynchronize(s));            // (no iteration, no control block to wait for)
  const auto t_solved = now();
  {
    const double *poses = w->h_state.p + offs[0] / 8, *vel = w->h_state.p + offs[1] / 8, *ba = w->h_state.p + offs[2] / 8, *bg = w->h_state.p + offs[3] / 8,
                 *invd = w->h_state.p + offs[4] / 8;
    for (int k = 0; k < n_kf; ++k) {
      lvf_window::Kf& f = w->kfs[k];
      std::memcpy(f.pose, &poses[(size_t)7 * k], 56);
      if (f.good_imu) { std::memcpy(f.vel, &vel[(size_t)3 * k], 24); std::memcpy(f.ba, &ba[(size_t)3 * k], 24); std::memcpy(f.bg, &bg[(size_t)3 * k], 24); }
    }
    for (size_t i = 0; i < nl; ++i) if (w->lms[i].alive) w->lms[i].inv_depth = invd[i];
  }
  w->slot_lm.clear();
  w->lmtab_valid = true;
  if (timing)
    std::fprintf(stderr, "lvf_window_solve (device assembly): host tables %.3f ms (%zu bytes staged), upload + assembly + counters %.3f ms, small uploads %.3f ms, configure %.3f ms, solve %.3f ms (%d its), read-back %.3f ms\n",
                 ms(t_begin, t_staged), cur, ms(t_staged, t_assembled), ms(t_assembled, t_uploaded), ms(t_uploaded, t_configured), ms(t_configured, t_solved), summary->num_iterations, ms(t_solved, now()));
  return LVF_OK;
}

extern "C" {
int lvf_window_solve(lvf_window* w, const lvf_solver_options* o, lvf_solver_summary* summary) {
  LVF_REQUIRE(w && o && summary, "lvf_window_solve: null argument");
  LVF_REQUIRE(!w->kfs.empty(), "lvf_window_solve: empty window");
  lvf_ctx* ctx = w->ctx;
  LVF_TRY(lvf::enter(ctx));
  static const bool host_asm = getenv("LVF_WINDOW_HOST_ASM") != nullptr;
  if (w->opt.device_assembly && !host_asm && (int)w->kfs.size() <= lvf::kDaMaxKf) return window_solve_device(w, o, summary);
  hipStream_t s = ctx->stream;
  const int n_kf = (int)w->kfs.size();
  w->lmtab_valid = false;                     // (host-side assembly: the device-resident landmark table, if any, goes stale)
  static const bool timing = getenv("LVF_WINDOW_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t_begin = now();
  // ---- assemble the block lists in BuildProblem's order, as 48-byte records in ONE pinned staging buffer (see k_window_unpack)
  std::vector<double> pr_t, pr_w, pr_v;
  std::vector<int32_t> imu_i, imu_j, pr_a, pr_b;
  std::vector<lvf_preint> imu_pre;
  w->slot_lm.clear();
  // landmark->ToWorld() once per live landmark per tick (the reference recomputes it per feature) with the keyframe rotations
  // expanded once, and per frame the one row of the world->cam0 transform that Camera::Far needs
  std::vector<LmHot>& hot = w->hot;
  landmarks_hot(w, hot);
  double inv_e[7];
  hse3::inv(w->left.extrinsic, inv_e);
  const double far_z = w->opt.baseline * 50.0;
  size_t n_obs = 0;
  for (const lvf_window::Kf& f : w->kfs) n_obs += f.obs.size();
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  // record regions sized by their upper bounds: one TwoCamera block per live landmark at most, every other feature one TwoFrame or
  // PoseOnly block; the PoseOnly region comes last so that the copy ends where its data ends
  const size_t cap_tc = std::min(n_obs, w->lms.size());
  const size_t off_tc = 0, off_tf = off_tc + cap_tc * 48, off_po = off_tf + n_obs * 48;
  const size_t tail_bound = up16((size_t)(17 * n_kf) * 8) + 8 * up16((size_t)n_kf * 8 + 64) + up16(w->lms.size() * 8 + 16) + up16((size_t)n_kf * 467 * 8) + 4096;
  LVF_TRY(w->h_stage.reserve(off_po + n_obs * 48 + tail_bound));
  unsigned char* hs = w->h_stage.p;
  TcRec* tcr = reinterpret_cast<TcRec*>(hs + off_tc);
  TfRec* tfr = reinterpret_cast<TfRec*>(hs + off_tf);
  PoRec* por = reinterpret_cast<PoRec*>(hs + off_po);
  std::vector<int32_t> kf2_counts(n_kf, 0);
  size_t ntc = 0, ntf = 0, npo = 0;
  // large windows: the records written so far go up while the rest is still being assembled (two extra submissions per third of the
  // frames; the 4 MB copy of a 50-keyframe window otherwise sits, ~0.1 ms, between the assembly and everything on the device)
  LVF_TRY(w->d_stage.ensure(off_po + n_obs * 48 + tail_bound));
  const bool chunked = n_obs >= 30000 && n_kf >= 6;
  size_t tc_sent = 0, tf_sent = 0;
  auto send_records = [&]() -> int {
    if (ntc > tc_sent) LVF_HIP(hipMemcpyAsync(w->d_stage.p + off_tc + tc_sent * 48, hs + off_tc + tc_sent * 48, (ntc - tc_sent) * 48, hipMemcpyHostToDevice, s));
    if (ntf > tf_sent) LVF_HIP(hipMemcpyAsync(w->d_stage.p + off_tf + tf_sent * 48, hs + off_tf + tf_sent * 48, (ntf - tf_sent) * 48, hipMemcpyHostToDevice, s));
    tc_sent = ntc; tf_sent = ntf;
    return LVF_OK;
  };
  for (int k = 0; k < n_kf; ++k) {
    if (chunked && (k == n_kf / 3 || k == (2 * n_kf) / 3)) LVF_TRY(send_records());
    lvf_window::Kf& f = w->kfs[k];
    f.sort_unique();
    // z_cam(pw) = zrow . pw + zoff  with  pc = R_e^-1 (R_wc^-1 pw + t_inv) + t_e_inv
    double inv_pose[7], zrow[3], zoff;
    hse3::inv(f.pose, inv_pose);
    {
      double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1}, c0[3], c1[3], c2[3], t[3], r[3];
      hse3::rotate(inv_pose, ex, r); hse3::rotate(inv_e, r, c0);
      hse3::rotate(inv_pose, ey, r); hse3::rotate(inv_e, r, c1);
      hse3::rotate(inv_pose, ez, r); hse3::rotate(inv_e, r, c2);
      zrow[0] = c0[2]; zrow[1] = c1[2]; zrow[2] = c2[2];
      hse3::rotate(inv_e, inv_pose + 4, t);
      zoff = t[2] + inv_e[6];
    }
    int near_visual = 0;
    for (const lvf_window::Obs& ob : f.obs) {
      LmHot& l = hot[ob.lm];
      if (l.birth_kf == f.id) {
        if (l.slot < 0) { l.slot = (int)w->slot_lm.size(); w->slot_lm.push_back(ob.lm); }
        TcRec& r = tcr[ntc++];
        r.l[0] = ob.ob[0]; r.l[1] = ob.ob[1]; r.r[0] = l.right_ob[0]; r.r[1] = l.right_ob[1]; r.lm = l.slot; r.kf = k; r.w = two_camera_weight(f.w_visual);
        continue;
      }
      const double* pw = l.pw;
      const int bpos = l.bpos;
      if (bpos < 0) {
        if (!l.fixed) continue;                        // birth frame unknown (never happens through this API)
        PoRec& r = por[npo];
        r.o[0] = ob.ob[0]; r.o[1] = ob.ob[1]; r.pw[0] = pw[0]; r.pw[1] = pw[1]; r.pw[2] = pw[2]; r.kf = k; r.pi = (int)npo;
        ++npo;
      } else {
        if (l.slot < 0) { l.slot = (int)w->slot_lm.size(); w->slot_lm.push_back(ob.lm); }
        TfRec& r = tfr[ntf];
        r.f[0] = l.right_ob[0]; r.f[1] = l.right_ob[1]; r.o[0] = ob.ob[0]; r.o[1] = ob.ob[1]; r.lm = l.slot; r.k1 = bpos; r.k2 = k;
        ++kf2_counts[k];
        ++ntf;
      }
      if (!(zrow[0] * pw[0] + zrow[1] * pw[1] + zrow[2] * pw[2] + zoff > far_z)) ++near_visual;   // !Camera::Far
    }
    bool imu_block = false;
    if (f.good_imu && k > 0 && w->kfs[k - 1].good_imu && f.has_pre) {
      imu_pre.push_back(f.pre); imu_i.push_back(k - 1); imu_j.push_back(k);
      imu_block = true;
    }
    if (!imu_block && near_visual < w->opt.weak_visual_threshold) {
      double t7[7] = {0, 0, 0, 0, 0, 0, 0};
      if (k > 0) { LVF_TRY(lvf_relative_rpyxyz(w->kfs[k - 1].pose, f.pose, t7)); pr_a.push_back(k - 1); }
      else { std::memcpy(t7, f.pose, 56); pr_a.push_back(-1); }
      pr_b.push_back(k); pr_t.insert(pr_t.end(), t7, t7 + 7); pr_w.push_back(w->opt.prior_weight); pr_v.push_back(w->opt.prior_v);
    }
  }
  for (size_t i = 0; i < hot.size(); ++i) w->lms[i].slot = hot[i].slot;
  LVF_REQUIRE(ntc <= cap_tc, "lvf_window_solve: more TwoCamera blocks than live landmarks (%zu > %zu)", ntc, cap_tc);
  const int n_lm = (int)w->slot_lm.size();
  w->n_lm_problem = n_lm;
  w->n_tc = (int)ntc; w->n_tf = (int)ntf; w->n_po = (int)npo; w->n_imu = (int)imu_i.size(); w->n_prior = (int)pr_b.size();

  const auto t_assembled = now();
  // ---- persistent device objects: created once (empty), re-filled every tick through grow-only buffers
  if (!w->st) {
    LVF_TRY(lvf_state_create(ctx, 0, 0, &w->st));
    const double z2[2] = {0, 0}; const int32_t z = 0; const double id7[7] = {0, 0, 0, 1, 0, 0, 0};
    LVF_TRY(lvf_two_camera_create(ctx, &w->left, &w->right, 0, z2, z2, &z, &z, &w->tc));
    LVF_TRY(lvf_two_frame_create(ctx, &w->left, &w->right, 0, z2, z2, &z, &z, &z, &w->tf));
    LVF_TRY(lvf_pose_only_create(ctx, &w->left, 0, z2, &z, &z, 0, id7, &w->po));
    LVF_TRY(lvf_imu_create(ctx, 0, nullptr, nullptr, nullptr, &w->imu));
    LVF_TRY(lvf_pose_prior_create(ctx, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &w->prior));
  }
  lvf_state* st = w->st;
  st->n_kf = n_kf; st->n_lm = n_lm;
  // the read-back staging of the state (filled after the solve)
  const size_t o_pose = 0, o_vel = (size_t)7 * n_kf, o_ba = o_vel + (size_t)3 * n_kf, o_bg = o_ba + (size_t)3 * n_kf, o_wv = o_bg + (size_t)3 * n_kf,
               o_invd = o_wv + n_kf;
  LVF_TRY(w->h_state.reserve(o_invd + n_lm));
  double *poses = w->h_state.p + o_pose, *vel = w->h_state.p + o_vel, *ba = w->h_state.p + o_ba, *bg = w->h_state.p + o_bg, *invd = w->h_state.p + o_invd;
  // the plain segments behind the PoseOnly records: state, IMU, priors.  Every destination gets two spare elements: segments are
  // copied in 16-byte words.
  UnpackArgs ua{};
  size_t cur = up16(off_po + npo * 48);
  auto seg = [&](void* dst, size_t bytes) -> unsigned char* {
    unsigned char* at = hs + cur;
    if (bytes) { ua.seg_dst[ua.n_segs] = static_cast<unsigned char*>(dst); ua.seg_off[ua.n_segs] = cur; ua.seg_words[ua.n_segs] = (unsigned)((bytes + 15) / 16); ++ua.n_segs; }
    cur += up16(bytes);
    return at;
  };
  auto idx_ok = [](lvf_batch* b, int n, int nkf, int nlm) { b->n = n; b->min_n_kf = nkf; b->min_n_lm = nlm; b->evaluated = false; };
  LVF_TRY(st->poses.ensure((size_t)7 * n_kf + 2)); LVF_TRY(st->vel.ensure((size_t)3 * n_kf + 2)); LVF_TRY(st->ba.ensure((size_t)3 * n_kf + 2)); LVF_TRY(st->bg.ensure((size_t)3 * n_kf + 2));
  LVF_TRY(st->w_visual.ensure((size_t)n_kf + 2)); LVF_TRY(st->inv_depth.ensure((size_t)n_lm + 2));
  st->poses.n = (size_t)7 * n_kf; st->vel.n = st->ba.n = st->bg.n = (size_t)3 * n_kf; st->w_visual.n = n_kf; st->inv_depth.n = n_lm;
  {
    double* sp = reinterpret_cast<double*>(seg(st->poses.p, (size_t)7 * n_kf * 8));
    double* sv = reinterpret_cast<double*>(seg(st->vel.p, (size_t)3 * n_kf * 8));
    double* sa = reinterpret_cast<double*>(seg(st->ba.p, (size_t)3 * n_kf * 8));
    double* sg = reinterpret_cast<double*>(seg(st->bg.p, (size_t)3 * n_kf * 8));
    double* sw = reinterpret_cast<double*>(seg(st->w_visual.p, (size_t)n_kf * 8));
    double* si = reinterpret_cast<double*>(seg(st->inv_depth.p, (size_t)n_lm * 8));
    for (int k = 0; k < n_kf; ++k) {
      const lvf_window::Kf& f = w->kfs[k];
      std::memcpy(&sp[(size_t)7 * k], f.pose, 56); std::memcpy(&sv[(size_t)3 * k], f.vel, 24); std::memcpy(&sa[(size_t)3 * k], f.ba, 24);
      std::memcpy(&sg[(size_t)3 * k], f.bg, 24); sw[k] = f.w_visual;
    }
    for (int l = 0; l < n_lm; ++l) si[l] = w->lms[w->slot_lm[l]].inv_depth;
  }
  // block arrays (filled by the unpack kernel from the records)
  LVF_TRY(w->tc->ob_a.ensure(2 * ntc)); LVF_TRY(w->tc->ob_b.ensure(2 * ntc)); LVF_TRY(w->tc->idx_a.ensure(ntc)); LVF_TRY(w->tc->idx_b.ensure(ntc)); LVF_TRY(w->tc->wblk.ensure(ntc));
  idx_ok(w->tc, w->n_tc, n_kf, n_lm);
  LVF_TRY(w->tf->ob_a.ensure(2 * ntf)); LVF_TRY(w->tf->ob_b.ensure(2 * ntf)); LVF_TRY(w->tf->idx_a.ensure(ntf)); LVF_TRY(w->tf->idx_b.ensure(ntf)); LVF_TRY(w->tf->idx_c.ensure(ntf));
  idx_ok(w->tf, w->n_tf, n_kf, n_lm);
  w->tf->sorted_by_kf = true; w->tf->host_kf1.clear(); w->tf->host_kf2.clear(); w->tf->host_lm.clear(); w->tf->unique_lk2_known = true; w->tf->kf2_counts = std::move(kf2_counts);   // Kf::sort_unique keeps one observation per (keyframe, landmark); assembled frame by frame: sorted by current keyframe
  LVF_TRY(w->po->ob_a.ensure(2 * npo)); LVF_TRY(w->po->idx_a.ensure(npo)); LVF_TRY(w->po->idx_b.ensure(npo)); LVF_TRY(w->po->table.ensure(3 * npo));
  idx_ok(w->po, w->n_po, n_kf, 0); w->po->n_table = w->n_po; w->po->sorted_by_kf = true;
  {
    lvf_batch* b = w->imu;
    const size_t ni = (size_t)w->n_imu;
    LVF_TRY(b->pre.ensure(467 * ni + 2)); LVF_TRY(b->idx_a.ensure(ni + 4)); LVF_TRY(b->idx_b.ensure(ni + 4));
    b->pre.n = 467 * ni; b->idx_a.n = b->idx_b.n = ni;
    if (ni) {
      std::memcpy(seg(b->pre.p, 467 * ni * 8), imu_pre.data(), 467 * ni * 8);
      std::memcpy(seg(b->idx_a.p, ni * 4), imu_i.data(), ni * 4); std::memcpy(seg(b->idx_b.p, ni * 4), imu_j.data(), ni * 4);
    }
    b->host_kf1 = imu_i; b->host_kf2 = imu_j;
    LVF_TRY(b->sqrt_info.ensure((size_t)225 * w->n_imu)); LVF_TRY(b->res.ensure((size_t)15 * w->n_imu));
    for (int q = 0; q < 8; ++q) LVF_TRY(b->jac[q].ensure((size_t)15 * b->block_size[q] * w->n_imu));
    idx_ok(b, w->n_imu, n_kf, 0);
  }
  {
    lvf_batch* b = w->prior;
    const size_t np_ = (size_t)w->n_prior;
    LVF_TRY(b->idx_a.ensure(np_ + 4)); LVF_TRY(b->idx_b.ensure(np_ + 4)); LVF_TRY(b->table.ensure(7 * np_ + 2)); LVF_TRY(b->ob_a.ensure(np_ + 2)); LVF_TRY(b->ob_b.ensure(np_ + 2));
    b->idx_a.n = b->idx_b.n = np_; b->table.n = 7 * np_; b->ob_a.n = b->ob_b.n = np_;
    if (np_) {
      std::memcpy(seg(b->idx_a.p, np_ * 4), pr_a.data(), np_ * 4); std::memcpy(seg(b->idx_b.p, np_ * 4), pr_b.data(), np_ * 4);
      std::memcpy(seg(b->table.p, 7 * np_ * 8), pr_t.data(), 7 * np_ * 8);
      std::memcpy(seg(b->ob_a.p, np_ * 8), pr_w.data(), np_ * 8); std::memcpy(seg(b->ob_b.p, np_ * 8), pr_v.data(), np_ * 8);
    }
    LVF_TRY(b->res.ensure((size_t)6 * w->n_prior)); LVF_TRY(b->jac[0].ensure((size_t)42 * w->n_prior)); LVF_TRY(b->jac[1].ensure((size_t)42 * w->n_prior));
    idx_ok(b, w->n_prior, n_kf, 0);
    b->host_kf1 = pr_a; b->host_kf2 = pr_b;
  }
  LVF_REQUIRE(cur <= w->h_stage.cap && ua.n_segs <= kMaxSegs, "lvf_window_solve: staging overflow");
  // the rest of the records, then the PoseOnly records and the plain segments in one copy; one unpack launch
  LVF_TRY(send_records());
  if (cur > off_po) LVF_HIP(hipMemcpyAsync(w->d_stage.p + off_po, hs + off_po, cur - off_po, hipMemcpyHostToDevice, s));
  ua.stage = w->d_stage.p; ua.ntc = (int)ntc; ua.ntf = (int)ntf; ua.npo = (int)npo; ua.off_tc = off_tc; ua.off_tf = off_tf; ua.off_po = off_po;
  ua.tc_l = reinterpret_cast<double2*>(w->tc->ob_a.p); ua.tc_r = reinterpret_cast<double2*>(w->tc->ob_b.p); ua.tc_lm = w->tc->idx_a.p; ua.tc_kf = w->tc->idx_b.p; ua.tc_w = w->tc->wblk.p;
  ua.tf_f = reinterpret_cast<double2*>(w->tf->ob_a.p); ua.tf_o = reinterpret_cast<double2*>(w->tf->ob_b.p); ua.tf_lm = w->tf->idx_a.p; ua.tf_k1 = w->tf->idx_b.p; ua.tf_k2 = w->tf->idx_c.p;
  ua.po_o = reinterpret_cast<double2*>(w->po->ob_a.p); ua.po_pw = w->po->table.p; ua.po_kf = w->po->idx_a.p; ua.po_pi = w->po->idx_b.p;
  ua.g_tc = (int)((ntc + 255) / 256); ua.g_tf = (int)((ntf + 255) / 256); ua.g_po = (int)((npo + 255) / 256); ua.g_seg = 64;
  hipLaunchKernelGGL(k_window_unpack, dim3(ua.g_tc + ua.g_tf + ua.g_po + ua.g_seg), dim3(256), 0, s, ua);
  LVF_HIP(hipGetLastError());
  if (w->n_imu) LVF_TRY(launch_imu_sqrt_info(w->imu));
  w->sq_valid = false;                               // (this path factors every pair: nothing is tracked for the cached form)
  const auto t_uploaded = now();
  if (!w->prob) {
    LVF_TRY(lvf_problem_create(ctx, st, w->tc, w->tf, w->po, w->imu, &w->prob));
    LVF_TRY(lvf_problem_set_pose_priors(w->prob, w->prior));
  } else {
    LVF_TRY(problem_configure(w->prob));
  }
  const auto t_configured = now();
  LVF_TRY(lvf_problem_solve(w->prob, o, summary));
  const auto t_solved = now();
  // ---- read the solution back into the host mirror (frame->pose, Vw, biases, landmark->inv_depth): gathered on the device, one copy
  {
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    PackArgs pa{};
    size_t at = 0;
    size_t offs[5];
    auto seg = [&](const void* src, size_t bytes, int slot) {
      offs[slot] = at;
      if (bytes) { pa.src[pa.n_segs] = static_cast<const unsigned char*>(src); pa.off[pa.n_segs] = at; pa.words[pa.n_segs] = (unsigned)((bytes + 15) / 16); ++pa.n_segs; }
      at += up16(bytes);
    };
    seg(st->poses.p, (size_t)7 * n_kf * 8, 0); seg(st->vel.p, (size_t)3 * n_kf * 8, 1); seg(st->ba.p, (size_t)3 * n_kf * 8, 2); seg(st->bg.p, (size_t)3 * n_kf * 8, 3);
    seg(st->inv_depth.p, (size_t)n_lm * 8, 4);
    LVF_TRY(w->d_stage.ensure(at + 16)); LVF_TRY(w->h_state.reserve(at / 8 + 2));
    pa.stage = w->d_stage.p;
    hipLaunchKernelGGL(k_window_pack, dim3(32), dim3(256), 0, s, pa);
    LVF_HIP(hipGetLastError());
    LVF_HIP(hipMemcpyAsync(w->h_state.p, w->d_stage.p, at, hipMemcpyDeviceToHost, s));
    LVF_HIP(hipStreamSynchronize(s));
    poses = w->h_state.p + offs[0] / 8; vel = w->h_state.p + offs[1] / 8; ba = w->h_state.p + offs[2] / 8; bg = w->h_state.p + offs[3] / 8; invd = w->h_state.p + offs[4] / 8;
  }
  for (int k = 0; k < n_kf; ++k) {
    lvf_window::Kf& f = w->kfs[k];
    std::memcpy(f.pose, &poses[(size_t)7 * k], 56);
    if (f.good_imu) { std::memcpy(f.vel, &vel[(size_t)3 * k], 24); std::memcpy(f.ba, &ba[(size_t)3 * k], 24); std::memcpy(f.bg, &bg[(size_t)3 * k], 24); }
  }
  for (int l = 0; l < n_lm; ++l) w->lms[w->slot_lm[l]].inv_depth = invd[l];
  if (timing)
    std::fprintf(stderr, "lvf_window_solve: assemble %.3f ms, upload %.3f ms, configure %.3f ms, solve %.3f ms (%d its), read-back %.3f ms\n", ms(t_begin, t_assembled),
                 ms(t_assembled, t_uploaded), ms(t_uploaded, t_configured), ms(t_configured, t_solved), summary->num_iterations, ms(t_solved, now()));
  return LVF_OK;
}

// Backend::Optimize's outlier gate (src/lvio_fusion/src/backend.cpp:185-190, :229-245): every feature that is not its landmark's first
// observation is re-projected with weight 1 — compute_reprojection_error = |PoseOnlyReprojectionError(ob, landmark->ToWorld(), camera, 1)
// (frame->pose)| — and removed when the error exceeds max_px (10 in the reference).  The residual pass is the PoseOnly hot-path kernel
// on device (one batch over all such features of the window); the host only applies the removals to its mirror, like
// landmark->RemoveObservation(feature) + frame->RemoveFeature(feature).  removed_lm / removed_kf (may be NULL) receive up to `capacity`
// (landmark id, keyframe id) pairs in frame order; *n_removed the total.
int lvf_window_reject_outliers(lvf_window* w, double max_px, int64_t* removed_lm, int64_t* removed_kf, int capacity, int* n_removed) {
  LVF_REQUIRE(w && n_removed, "lvf_window_reject_outliers: null argument");
  LVF_REQUIRE(capacity >= 0 && (capacity == 0 || (removed_lm && removed_kf)), "lvf_window_reject_outliers: bad output arrays");
  *n_removed = 0;
  if (w->kfs.empty()) return LVF_OK;
  lvf_ctx* ctx = w->ctx;
  LVF_TRY(lvf::enter(ctx));
  hipStream_t s = ctx->stream;
  const int n_kf = (int)w->kfs.size();
  std::vector<double> lm_pw;
  std::vector<int> lm_birth_pos;
  landmarks_to_world(w, lm_pw, lm_birth_pos);
  // features that are not their landmark's first observation, in frame order / ascending landmark id
  std::vector<double> ob, pw, poses((size_t)7 * n_kf), ones(n_kf, 1.0);
  std::vector<int32_t> kf, pi;
  std::vector<std::pair<int, int>> where;          // (keyframe position, index into its obs)
  for (int k = 0; k < n_kf; ++k) {
    lvf_window::Kf& f = w->kfs[k];
    f.sort_unique();
    std::memcpy(&poses[(size_t)7 * k], f.pose, 56);
    for (size_t j = 0; j < f.obs.size(); ++j) {
      const lvf_window::Obs& o = f.obs[j];
      const lvf_window::Lm& l = w->lms[o.lm];
      if (l.birth_kf == f.id) continue;
      if (!l.fixed && lm_birth_pos[o.lm] < 0) continue;   // birth pose unknown (never happens through this API)
      ob.push_back(o.ob[0]); ob.push_back(o.ob[1]);
      pw.insert(pw.end(), &lm_pw[(size_t)3 * o.lm], &lm_pw[(size_t)3 * o.lm] + 3);
      pi.push_back((int32_t)kf.size()); kf.push_back(k);
      where.emplace_back(k, (int)j);
    }
  }
  const int n = (int)kf.size();
  if (n == 0) return LVF_OK;
  if (!w->rej) {
    const double z2[2] = {0, 0}; const int32_t z = 0; const double id7[7] = {0, 0, 0, 1, 0, 0, 0};
    LVF_TRY(lvf_pose_only_create(ctx, &w->left, 0, z2, &z, &z, 0, id7, &w->rej));
    LVF_TRY(lvf_state_create(ctx, 0, 0, &w->rej_st));
  }
  lvf_batch* b = w->rej;
  lvf_state* st = w->rej_st;
  st->n_kf = n_kf; st->n_lm = 0;
  LVF_TRY(st->poses.assign(poses.data(), poses.size(), s)); LVF_TRY(st->w_visual.assign(ones.data(), ones.size(), s));
  LVF_TRY(b->ob_a.assign(ob.data(), ob.size(), s)); LVF_TRY(b->idx_a.assign(kf.data(), kf.size(), s)); LVF_TRY(b->idx_b.assign(pi.data(), pi.size(), s));
  LVF_TRY(b->table.assign(pw.data(), pw.size(), s)); LVF_TRY(b->res.ensure((size_t)2 * n));
  b->n = n; b->n_table = n; b->min_n_kf = n_kf; b->min_n_lm = 0; b->sorted_by_kf = true; b->evaluated = false;
  LVF_TRY(launch_pose_only(b, st, false));
  LVF_TRY(w->rej_flags.ensure(n));
  hipLaunchKernelGGL(k_flag_outliers, dim3((n + 255) / 256), dim3(256), 0, s, n, reinterpret_cast<const double2*>(b->res.p), max_px, w->rej_flags.p);
  LVF_HIP(hipGetLastError());
  std::vector<uint8_t> flags(n);
  LVF_HIP(hipMemcpyAsync(flags.data(), w->rej_flags.p, n, hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  // apply the removals (back to front inside each keyframe so the stored positions stay valid)
  int total = 0;
  for (int i = 0; i < n; ++i)
    if (flags[i]) {
      if (total < capacity) { removed_lm[total] = w->kfs[where[i].first].obs[where[i].second].lm_id; removed_kf[total] = w->kfs[where[i].first].id; }
      ++total;
    }
  for (int i = n - 1; i >= 0; --i)
    if (flags[i]) { auto& v = w->kfs[where[i].first].obs; v.erase(v.begin() + where[i].second); w->kfs[where[i].first].d_dirty = true; }
  *n_removed = total;
  if (total) prune(w);
  return LVF_OK;
}

// Test hook (tests/test_gpu_window.py): the block lists of the LAST lvf_window_solve as they sit in the device batches — in the batch's order,
// keyframe positions translated to keyframe ids and landmark slots to landmark ids — so that the assembly (host walk or device kernels) can be
// compared with the reference's Backend::BuildProblem block by block (backend.cpp:96-183; tests/golden/ref_v4.npz).
//   kind 0 TwoCamera : ids {landmark, -1, keyframe}            vals {5 w_visual[kf], left ob x, y, right ob x, y, 0, 0, 0}
//   kind 1 PoseOnly  : ids {-1, -1, keyframe}                  vals {w_visual[kf], ob x, y, 0, 0, pw x, y, z}
//   kind 2 TwoFrame  : ids {landmark, first keyframe, keyframe} vals {w_visual[kf], ob x, y, first (right) ob x, y, 0, 0, 0}
//   kind 3 ImuError  : ids {-1, keyframe i, keyframe j}
//   kind 4 priors    : ids {-1, previous keyframe or -1 (PoseError), keyframe}   vals {weight, v, 0, ...}
// ids [capacity][3], vals [capacity][8]; *n = blocks of that kind (may exceed capacity: only `capacity` are written).
int lvf_window_debug_blocks(lvf_window* w, int kind, int capacity, int64_t* ids, double* vals, int* n) {
  LVF_REQUIRE(w && n && kind >= 0 && kind <= 4 && capacity >= 0 && (capacity == 0 || (ids && vals)), "lvf_window_debug_blocks: bad arguments");
  LVF_TRY(lvf::enter(w->ctx));
  hipStream_t s = w->ctx->stream;
  lvf_batch* b = kind == 0 ? w->tc : kind == 1 ? w->po : kind == 2 ? w->tf : kind == 3 ? w->imu : w->prior;
  const int nb = kind == 0 ? w->n_tc : kind == 1 ? w->n_po : kind == 2 ? w->n_tf : kind == 3 ? w->n_imu : w->n_prior;
  *n = nb;
  if (!b || nb == 0 || capacity == 0) return LVF_OK;
  const int m = std::min(nb, capacity), n_kf = (int)w->kfs.size();
  auto get_i = [&](const lvf::DevBuf<int32_t>& d, std::vector<int32_t>& h, size_t cnt) -> int { h.resize(cnt); LVF_HIP(hipMemcpyAsync(h.data(), d.p, cnt * 4, hipMemcpyDeviceToHost, s)); return LVF_OK; };
  auto get_d = [&](const lvf::DevBuf<double>& d, std::vector<double>& h, size_t cnt) -> int { h.resize(cnt); LVF_HIP(hipMemcpyAsync(h.data(), d.p, cnt * 8, hipMemcpyDeviceToHost, s)); return LVF_OK; };
  std::vector<int32_t> ia, ib, ic, slot_lm;
  std::vector<double> oa, ob, tab;
  if (kind <= 2) { LVF_TRY(get_i(b->idx_a, ia, m)); LVF_TRY(get_d(b->ob_a, oa, (size_t)2 * m)); }
  if (kind == 0 || kind == 2) { LVF_TRY(get_i(b->idx_b, ib, m)); LVF_TRY(get_d(b->ob_b, ob, (size_t)2 * m)); }
  std::vector<double> wb;
  if (kind == 0 && b->wblk.n >= (size_t)m) LVF_TRY(get_d(b->wblk, wb, m));
  if (kind == 2) LVF_TRY(get_i(b->idx_c, ic, m));
  if (kind == 1) { LVF_TRY(get_i(b->idx_b, ib, m)); LVF_TRY(get_d(b->table, tab, (size_t)3 * b->n_table)); }
  if (kind == 3) { LVF_TRY(get_i(b->idx_a, ia, m)); LVF_TRY(get_i(b->idx_b, ib, m)); }
  if (kind == 4) { LVF_TRY(get_i(b->idx_a, ia, m)); LVF_TRY(get_i(b->idx_b, ib, m)); LVF_TRY(get_d(b->ob_a, oa, m)); LVF_TRY(get_d(b->ob_b, ob, m)); }
  const bool need_slots = kind == 0 || kind == 2;
  if (need_slots && (int)w->slot_lm.size() != w->n_lm_problem) {      // device assembly: the slot table lives on the device
    slot_lm.resize((size_t)w->n_lm_problem);
    if (w->n_lm_problem) LVF_HIP(hipMemcpyAsync(slot_lm.data(), w->d_slot_lm.p, (size_t)w->n_lm_problem * 4, hipMemcpyDeviceToHost, s));
  }
  LVF_HIP(hipStreamSynchronize(s));
  const std::vector<int>* sl = nullptr;
  std::vector<int> sl_copy;
  if (need_slots) { if (slot_lm.empty() && (int)w->slot_lm.size() == w->n_lm_problem) sl = &w->slot_lm; else { sl_copy.assign(slot_lm.begin(), slot_lm.end()); sl = &sl_copy; } }
  auto kf_id = [&](int pos) -> int64_t { return (pos >= 0 && pos < n_kf) ? w->kfs[pos].id : -1; };
  auto lm_id = [&](int slot) -> int64_t { return (sl && slot >= 0 && slot < (int)sl->size()) ? w->lms[(*sl)[slot]].id : -1; };
  auto wv = [&](int pos) { return (pos >= 0 && pos < n_kf) ? w->kfs[pos].w_visual : 0.0; };
  for (int i = 0; i < m; ++i) {
    int64_t* I = ids + 3 * (size_t)i; double* V = vals + 8 * (size_t)i;
    I[0] = I[1] = I[2] = -1;
    for (int q = 0; q < 8; ++q) V[q] = 0.0;
    if (kind == 0) { I[0] = lm_id(ia[i]); I[2] = kf_id(ib[i]); V[0] = wb.empty() ? 5 * wv(ib[i]) : wb[i]; V[1] = oa[2 * i]; V[2] = oa[2 * i + 1]; V[3] = ob[2 * i]; V[4] = ob[2 * i + 1]; }
    else if (kind == 1) { I[2] = kf_id(ia[i]); V[0] = wv(ia[i]); V[1] = oa[2 * i]; V[2] = oa[2 * i + 1]; const int t = ib[i]; if (t >= 0 && t < b->n_table) { V[5] = tab[3 * t]; V[6] = tab[3 * t + 1]; V[7] = tab[3 * t + 2]; } }
    else if (kind == 2) { I[0] = lm_id(ia[i]); I[1] = kf_id(ib[i]); I[2] = kf_id(ic[i]); V[0] = wv(ic[i]); V[1] = ob[2 * i]; V[2] = ob[2 * i + 1]; V[3] = oa[2 * i]; V[4] = oa[2 * i + 1]; }
    else if (kind == 3) { I[1] = kf_id(ia[i]); I[2] = kf_id(ib[i]); }
    else { I[1] = kf_id(ia[i]); I[2] = kf_id(ib[i]); V[0] = oa[i]; V[1] = ob[i]; }
  }
  return LVF_OK;
}

}  // extern "C"

