// lvf_internal.hpp — host-side objects behind the opaque C-ABI handles of include/lvf.h.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
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

// Per-context caching device allocator.  hipMalloc / hipFree cost 0.1-0.3 ms each (hipFree synchronises the device) and the
// create-heavy paths (map index build, cloud filters, feature extraction, per-tick batches) issue dozens of them per call;
// freed blocks are parked in size buckets and handed back to later requests of the SAME context, i.e. the same HIP stream,
// so reuse is stream-ordered and needs no extra synchronisation.  A closed pool (context destroyed) frees directly.
struct Pool {
  std::mutex mu;
  std::unordered_multimap<size_t, void*> parked;
  size_t held = 0;
  bool closed = false;
  // Beyond this much parked memory, blocks of OTHER sizes are evicted to make room for the one coming back: what is parked follows the work the
  // context is doing now.  (Round 4 refused the new block instead — 4 GB cap — so after the 64-window batches of bench.py had filled the pool
  // with sizes nothing else asks for, every later allocation of a run was a hipMalloc and every release a device-synchronising hipFree: the
  // map index built in 1.0 ms instead of 0.13, the feature extraction in 0.65 ms instead of 0.41, in that process only.)
  // The cap is a quarter of what the device had free when the context was created, at most 16 GB (lvf::enter sets it; a 16-GB constant on a
  // 16-64 GB part, or with several contexts on one GPU, let gigabytes sit parked while a new bucket size failed to allocate: ADVICE r05).
  // A failed hipMalloc additionally drains this pool and retries once (DevBuf::raw_alloc).
  static constexpr size_t kMaxHeldCeiling = (size_t)16 << 30;
  size_t kMaxHeld = (size_t)4 << 30;
  void set_cap_from_device() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) {
      std::lock_guard<std::mutex> g(mu);
      kMaxHeld = std::min(kMaxHeldCeiling, std::max((size_t)256 << 20, free_b / 4));
    }
  }
  // everything parked goes back to the device (the caller's hipMalloc failed): returns the bytes released
  size_t drain() {
    std::unordered_multimap<size_t, void*> all;
    size_t n = 0;
    { std::lock_guard<std::mutex> g(mu); all.swap(parked); n = held; held = 0; }
    for (auto& kv : all) (void)hipFree(kv.second);
    return n;
  }
  static size_t bucket(size_t bytes) {
    if (bytes <= 4096) return (bytes + 255) & ~(size_t)255;
    size_t b = 4096;
    while (b < bytes) b <<= 1;                           // next power of two ...
    const size_t step = b >> 4;                          // ... refined to 1/8-octave steps: <= 12.5 % slack
    return ((bytes + step - 1) / step) * step;
  }
  void* get(size_t bucket_bytes) {
    std::lock_guard<std::mutex> g(mu);
    auto it = parked.find(bucket_bytes);
    if (it == parked.end()) return nullptr;
    void* p = it->second;
    parked.erase(it);
    held -= bucket_bytes;
    return p;
  }
  bool put(void* p, size_t bucket_bytes) {               // false: caller must hipFree
    std::vector<void*> evicted;
    bool kept = false;
    {
      std::lock_guard<std::mutex> g(mu);
      if (closed || bucket_bytes > kMaxHeld) return false;
      for (auto it = parked.begin(); held + bucket_bytes > kMaxHeld && it != parked.end();) {
        if (it->first == bucket_bytes) { ++it; continue; }      // (blocks of the size in use stay)
        evicted.push_back(it->second); held -= it->first;
        it = parked.erase(it);
      }
      if (held + bucket_bytes <= kMaxHeld) { parked.emplace(bucket_bytes, p); held += bucket_bytes; kept = true; }
    }
    for (void* q : evicted) (void)hipFree(q);            // (outside the lock: hipFree waits for the device)
    return kept;
  }
  void close() {
    std::unordered_multimap<size_t, void*> all;
    { std::lock_guard<std::mutex> g(mu); closed = true; all.swap(parked); held = 0; }
    for (auto& kv : all) (void)hipFree(kv.second);
  }
  ~Pool() { close(); }
};
extern thread_local std::shared_ptr<Pool> g_pool;         // the calling entry point's context pool (set by lvf::enter)

// grow-only PINNED host staging array (hipHostMalloc): what the persistent window assembles its block lists into, so that the
// per-tick uploads are real asynchronous DMA instead of pageable copies through the runtime's bounce buffer, and nothing is
// re-allocated or zero-filled per tick.  Contents are not preserved across a growth.
// Pinned blocks are recycled process-wide in size buckets: hipHostMalloc / hipHostFree page-lock and unlock memory (hundreds of
// microseconds per megabyte), and a problem built per ceres::Solve call (the adapter's path) owns half a dozen of them.
struct HostPinPool {
  std::mutex mu;
  std::unordered_multimap<size_t, void*> parked;
  size_t held = 0;
  static constexpr size_t kMaxHeld = (size_t)256 << 20;
  static HostPinPool& get() { static HostPinPool* g = new HostPinPool(); return *g; }      // (never destroyed: the runtime may be gone at exit)
  void* take(size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    auto it = parked.find(bytes);
    if (it == parked.end()) return nullptr;
    void* p = it->second;
    parked.erase(it); held -= bytes;
    return p;
  }
  bool give(void* p, size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    if (held + bytes > kMaxHeld) return false;
    parked.emplace(bytes, p); held += bytes;
    return true;
  }
};
template <typename T>
struct HostPin {
  T* p = nullptr;
  size_t cap = 0;
  size_t bytes = 0;
  HostPin() = default;
  HostPin(const HostPin&) = delete;
  HostPin& operator=(const HostPin&) = delete;
  ~HostPin() { drop(); }
  void swap(HostPin& o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(bytes, o.bytes); }
  void drop() {
    if (p && !HostPinPool::get().give(p, bytes)) (void)hipHostFree(p);
    p = nullptr; cap = 0; bytes = 0;
  }
  int reserve(size_t count) {
    if (count <= cap) return LVF_OK;
    drop();
    const size_t want = Pool::bucket((count + count / 4 + 64) * sizeof(T));
    void* q = HostPinPool::get().take(want);
    if (!q) {
      hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
      if (e != hipSuccess) return ::lvf::hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    p = static_cast<T*>(q); cap = want / sizeof(T); bytes = want;
    return LVF_OK;
  }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

// uploads up to this size go through pooled pinned staging (LVF_STAGE_MAX_KB: A/B)
inline size_t stage_max_bytes() {
  static const size_t v = [] { const char* e = std::getenv("LVF_STAGE_MAX_KB"); return e ? (size_t)std::atol(e) << 10 : (size_t)1 << 20; }();
  return v;
}
template <typename T>
struct DevBuf {  // owning device buffer
  T* p = nullptr;
  size_t n = 0;
  size_t cap = 0;   // usable elements (>= n); ensure()/assign() only ever grow it
  size_t bytes = 0; // bucketed allocation size
  std::shared_ptr<Pool> pool;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(bytes, o.bytes); pool.swap(o.pool); }
  void release() {
    if (p) { if (!(pool && pool->put(p, bytes))) (void)hipFree(p); }
    p = nullptr; cap = 0; bytes = 0; pool.reset();
  }
  int raw_alloc(size_t count) {
    release();
    if (count == 0) return LVF_OK;
    pool = g_pool;
    bytes = Pool::bucket(count * sizeof(T));
    void* q = pool ? pool->get(bytes) : nullptr;
    if (!q) {
      hipError_t e = hipMalloc(&q, bytes);
      if (e == hipErrorOutOfMemory && pool && pool->drain() > 0) {      // parked blocks of other sizes were holding the memory: give them back, once
        (void)hipGetLastError();
        e = hipMalloc(&q, bytes);
      }
      if (e != hipSuccess) { bytes = 0; pool.reset(); return ::lvf::hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
    }
    p = static_cast<T*>(q);
    cap = count;
    return LVF_OK;
  }
  int alloc(size_t count) {
    n = count;
    return raw_alloc(count);
  }
  // grow-only resize (contents are NOT preserved across a growth): the persistent-window path re-uses buffers across ticks
  int ensure(size_t count) {
    if (count <= cap && (p || count == 0)) { n = count; return LVF_OK; }
    LVF_TRY(raw_alloc(count + count / 4 + 16));
    n = count;
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
  // the same through a pooled PINNED staging block (`stage`, which must stay alive until the stream has been waited for): a pageable
  // source makes the runtime pin and unpin the caller's pages around the copy, which serialises host threads that upload on different
  // streams (measured: eight loop-closure candidates on four streams gained nothing in a long-running process)
  template <typename Stage>
  int upload_staged(const T* host, size_t count, hipStream_t s, Stage& stage) {
    LVF_TRY(alloc(count));
    if (!count) return LVF_OK;
    if (count * sizeof(T) > stage_max_bytes()) {       // large: the copy through host memory costs more than the pinning (5.4 MB: +0.5 ms)
      LVF_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
      return LVF_OK;
    }
    LVF_TRY(stage.reserve(count));
    std::memcpy(stage.p, host, count * sizeof(T));
    LVF_HIP(hipMemcpyAsync(p, stage.p, count * sizeof(T), hipMemcpyHostToDevice, s));
    return LVF_OK;
  }
  // exchange the storage (not the logical size) of two buffers: the accepted-step pointer swap of the solver
  void swap_storage(DevBuf& o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(bytes, o.bytes); std::swap(pool, o.pool); }
};

}  // namespace lvf

struct lvf_ctx {
  std::shared_ptr<lvf::Pool> pool = std::make_shared<lvf::Pool>();
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cu = 256;
  lvf::HostPin<char> mailbox;       // pinned landing zone of the small read-backs (counters, bounds, moments): see lvf::read_back
  lvf::HostPin<char> stage;         // pinned staging of host arrays up to 1 MB that are copied and waited for in one call (lvf_state_set / _get)
  // status words + ticket of the one-launch scan (sort_util.hip device_scan1): self-cleaning through the epoch, so it lives with the context
  // (two lanes: launches on the context's stream use lane 0, launches on its side stream lane 1)
  unsigned long long* scan_status = nullptr;
  int scan_tiles = 0;
  unsigned scan_epoch = 0;
  // set by device_scan1_on (the one-launch scan's sticky error word, raised by a tile that gave up waiting for a predecessor): the next small
  // read_back() of this context fetches it along with the counts and fails with LVF_ERR_STATE instead of handing out a count that may be wrong.
  // A context is driven by ONE host thread at a time (tickets, status words and the epoch live on it, unguarded): INTEGRATION.md, threading.
  unsigned long long* scan_err = nullptr;
  // side stream for the second of two independent launch chains inside one call (lvf_lidar_extract's ground tail beside the surf tail), with
  // the events that fork it off the context's stream and join it back; created on first use (lvf::side_stream)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

namespace lvf {
// Small device -> host read-back + wait, through the context's pinned mailbox: a copy into a stack variable takes the runtime's pageable
// path (the page is pinned and unpinned around the copy, ~10-15 us each time; the map index alone reads back once per grid level).
// One use at a time per context — every caller waits for the stream before it returns, and a context belongs to one host thread.
inline int read_back(lvf_ctx* ctx, void* host_out, const void* dev, size_t bytes) {
  if (bytes == 0) return LVF_OK;
  if (bytes > 4096) {          // (not a scalar read-back: the plain path)
    hipError_t e = hipMemcpyAsync(host_out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return ::lvf::hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return ::lvf::hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    return LVF_OK;
  }
  LVF_TRY(ctx->mailbox.reserve(4096 + 64));
  hipError_t e = hipMemcpyAsync(ctx->mailbox.p, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e != hipSuccess) return ::lvf::hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
  unsigned long long* const scan_err = ctx->scan_err;
  if (scan_err) {                // a one-launch scan ran on this context since the last look: its error word rides along
    ctx->scan_err = nullptr;
    e = hipMemcpyAsync(reinterpret_cast<char*>(ctx->mailbox.p) + 4096, scan_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return ::lvf::hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
  }
  e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ::lvf::hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
  std::memcpy(host_out, ctx->mailbox.p, bytes);
  if (scan_err) {
    unsigned long long w = 0;
    std::memcpy(&w, reinterpret_cast<const char*>(ctx->mailbox.p) + 4096, sizeof(w));
    if (w != 0) {
      (void)hipMemsetAsync(scan_err, 0, sizeof(unsigned long long), ctx->stream);
      ::lvf::set_error("device scan: tile %llu of launch %llu never published its sum (status block corrupted, or one context driven by two host threads)", (w & 0xffffffffull) - 1, w >> 32);
      return LVF_ERR_STATE;
    }
  }
  return LVF_OK;
}
// host array -> device / device -> host array, waited for, through the context's pinned staging when it is at most 1 MB (the same
// reason: a pageable source / destination is pinned and unpinned around every copy)
inline int copy_up_wait(lvf_ctx* ctx, void* dev, const void* host, size_t bytes) {
  if (bytes == 0) return LVF_OK;
  const void* src = host;
  if (bytes <= ((size_t)1 << 20)) { LVF_TRY(ctx->stage.reserve(bytes)); std::memcpy(ctx->stage.p, host, bytes); src = ctx->stage.p; }
  hipError_t e = hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) return ::lvf::hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
  e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ::lvf::hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
  return LVF_OK;
}
inline int copy_down_wait(lvf_ctx* ctx, void* host, const void* dev, size_t bytes) {
  const bool staged = bytes > 0 && bytes <= ((size_t)1 << 20);
  if (staged) LVF_TRY(ctx->stage.reserve(bytes));
  hipError_t e = bytes ? hipMemcpyAsync(staged ? (void*)ctx->stage.p : host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
  if (e != hipSuccess) return ::lvf::hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
  e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ::lvf::hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
  if (staged) std::memcpy(host, ctx->stage.p, bytes);
  return LVF_OK;
}
}  // namespace lvf

namespace lvf {
// Declared right AFTER a pinned staging block that feeds an asynchronous copy (HostPin + DevBuf::upload_staged): destroyed BEFORE it, on every
// path out of the function — early error returns included — it waits for the stream, so the block never goes back to the process-wide pinned
// pool (where another thread could take and overwrite it) while the DMA may still be reading it.
struct StreamWaitGuard {
  hipStream_t s;
  bool armed = true;
  explicit StreamWaitGuard(hipStream_t q) : s(q) {}
  StreamWaitGuard(const StreamWaitGuard&) = delete;
  StreamWaitGuard& operator=(const StreamWaitGuard&) = delete;
  void dismiss() { armed = false; }        // the staging block has found an owner that outlives the copy
  ~StreamWaitGuard() { if (armed) (void)hipStreamSynchronize(s); }
};
}  // namespace lvf

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
  bool unique_lk2_known = false;  // two-frame: the creator guarantees no (landmark, current keyframe) pair repeats (skips the host check)
  std::vector<int32_t> kf2_counts; // two-frame, set by a creator that ALSO guarantees k1 < k2 for every block: blocks per current keyframe (ascending);
                                   // the solver then builds its work list from these counts without walking per-block host copies of the indices
  int min_n_kf = 0, min_n_lm = 0;  // smallest state the index arrays are valid for
  std::vector<int32_t> host_kf1, host_kf2, host_lm;  // two-frame batches keep their indices for the solver's work list / uniqueness check
  lvf::CamD cam_a{}, cam_b{};  // visual: cam_a = left / cam0, cam_b = right
  // inputs (device)
  lvf::DevBuf<double> ob_a, ob_b;        // [n][2] each
  lvf::DevBuf<int32_t> idx_a, idx_b, idx_c;
  lvf::DevBuf<double> table;             // pose-only: pw[n_pw][3]
  lvf::DevBuf<double> wblk;              // two-camera: optional per-block weight (the functor's ctor argument); empty = 5 * w_visual[kf]
  int n_table = 0;
  // lidar
  int lidar_mode = 0;
  double lidar_weight = 1.0;
  double Twc1[7] = {0, 0, 0, 1, 0, 0, 0};
  lvf::DevBuf<double> lp, lpa, lnrm;     // SoA [3][n]
  lvf::DevBuf<char> icp_dev;             // device LM state of lvf_lidar_solve
  lvf::HostPin<char> icp_host;           // its pinned host mirror (initial state up, result down: real asynchronous copies)
  // imu
  lvf::DevBuf<double> pre;               // [n][467] flattened lvf_preint
  lvf::DevBuf<double> sqrt_info;         // [n][225]
  // outputs (device)
  lvf::DevBuf<double> res;
  lvf::DevBuf<double> jac[8];
};

#define LVF_MAX_GRID_LEVELS 8
struct lvf_map {
  struct Level {
    lvf::DevBuf<float4> sorted;      // cell-sorted points, .w = bitcast original index
    lvf::DevBuf<int> cell_start;     // ncells + 1
    float ox = 0, oy = 0, oz = 0, cell = 1, inv_cell = 1;
    int nx = 1, ny = 1, nz = 1;
    Level() = default;
    Level(Level&& o) noexcept { *this = std::move(o); }
    Level& operator=(Level&& o) noexcept {
      sorted.swap_storage(o.sorted); std::swap(sorted.n, o.sorted.n);
      cell_start.swap_storage(o.cell_start); std::swap(cell_start.n, o.cell_start.n);
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
  lvf::HostPin<char> icp_host;       // its pinned host mirror (initial state up, result down: real asynchronous copies)
  // the upload of lvf_scan_create is not waited for: its staging (pinned block + raw device copy) belongs to the scan until the scan goes
  lvf::DevBuf<float> create_src; lvf::HostPin<float> create_stage; hipEvent_t create_done = nullptr;      // recorded behind the upload: what lvf_scan_destroy waits for (the context may be gone by then)
  ~lvf_scan() { if (create_done) { (void)hipEventSynchronize(create_done); (void)hipEventDestroy(create_done); } }      // (runs before the members above are released)
};

struct lvf_cloud {
  lvf_ctx* ctx = nullptr;
  int n = 0;
  lvf::DevBuf<float4> pts;           // x, y, z, intensity  (pcl::PointXYZI payload, 16 B on device)
};

struct lvf_problem;
namespace lvf {
// butterfly inside each group of four lanes on the DPP path (quad_perm, VALU latency): __shfl_xor compiles to ds_bpermute, an LDS
// round trip of ~120 cycles, and two of them per step WERE the panel solve's dependent chain (11 us per 64 columns)
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {     // any DPP control word, not only quad_perm ones
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += quad_perm<0xB1>(v);     // lanes [1,0,3,2]
  v += quad_perm<0x4E>(v);     // lanes [2,3,0,1]
  return v;
}

// broadcast of one lane's double through v_readlane (scalar result: no VGPR, no LDS round trip)
__device__ __forceinline__ double lane_bcast(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes on the DPP path: Hillis-Steele inside each 16-lane row (row_shr 1, 2, 4, 8; out-of-row sources
// read zero), then lane 15 of a row into the next row (row_bcast:15, rows 1 and 3) and lane 31 into rows 2 and 3 (row_bcast:31).  Six
// VALU instructions; through __shfl_up it is six dependent ds_bpermute round trips (~100 clocks each).
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return v;
}
// sum over each aligned group of 16 lanes (a DPP "row"), delivered to all 16: the first four steps of wave_sum
__device__ __forceinline__ double row16_sum(double v) {
  v = quad_sum(v);
  v += quad_perm<0x141>(v);    // row_half_mirror
  v += quad_perm<0x140>(v);    // row_mirror
  return v;
}
// wave-wide sum on the DPP path: 4 butterfly steps inside each 16-lane row (quad_perm x2, row_half_mirror, row_mirror), then
// the four row sums are read as scalars.  Every lane returns the total.  (The ds_bpermute-based __shfl_down ladder is 6 dependent
// LDS round trips per value, and the linearisation kernels reduce 27 values per wave.)  Call with all 64 lanes active.
__device__ __forceinline__ double wave_sum(double v) {
  v = quad_sum(v);
  v += quad_perm<0x141>(v);    // row_half_mirror
  v += quad_perm<0x140>(v);    // row_mirror
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
// Sums of up to 32 per-lane values over the wave at once ("transposed" reduction): instead of 32 independent wave_sum()s (23 instructions
// each), every butterfly stage halves the number of values a lane carries — the lane keeps the half selected by one bit of its lane id and
// adds the partner's copy of that half — so 16 + 8 + 4 + 2 + 1 pair-combinations plus one final add do all 32 sums (~125 instructions):
//   lane bits 5 and 4 (across 16-lane rows): v_permlane32_swap / v_permlane16_swap exchange the halves of a register pair in place,
//       two swaps + one add per pair, no select;
//   lane bits 3, 2, 1 (inside a row): a select of keep / send followed by a DPP move (row_ror:8, row_half_mirror, quad_perm [2,3,0,1]);
//   lane bit 0: a plain quad_perm add.
// On return EVERY lane l holds the wave-wide sum of v[l >> 1].  v is clobbered.
typedef unsigned lvf_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap_halves32(double& x, double& y) {        // lanes 32-63 of x <-> lanes 0-31 of y
  const lvf_v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const lvf_v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void swap_rows16(double& x, double& y) {          // odd rows of x <-> even rows of y
  const lvf_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const lvf_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double wave_sum32(double v[32]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 16; ++i) { swap_halves32(v[i], v[i + 16]); v[i] += v[i + 16]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { swap_rows16(v[i], v[i + 8]); v[i] += v[i + 8]; }
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double keep = b3 ? v[i + 4] : v[i], send = b3 ? v[i] : v[i + 4]; v[i] = keep + quad_perm<0x128>(send); }   // row_ror:8
#pragma unroll
  for (int i = 0; i < 2; ++i) { const double keep = b2 ? v[i + 2] : v[i], send = b2 ? v[i] : v[i + 2]; v[i] = keep + quad_perm<0x141>(send); }   // row_half_mirror
  { const double keep = b1 ? v[1] : v[0], send = b1 ? v[0] : v[1]; v[0] = keep + quad_perm<0x4E>(send); }                                         // lanes [2,3,0,1]
  return v[0] + quad_perm<0xB1>(v[0]);                                                                                                            // lanes [1,0,3,2]
}

// Same-address global atomics serialise in L2 at ~65 ns each (measured), so counters are touched once per RUN of consecutive lanes
// holding the same key (clouds arrive in scan order: a coarse cell, a segment or a voxel sees long runs): only the head lane of a
// run issues the atomic, with the run length.  `start` = head lane of this lane's run, `len` = run
// length (meaningful on head lanes).
__device__ __forceinline__ bool cell_runs(int c, int& start, int& len) {
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(c, 1);
  const bool head = lane == 0 || c != prev;
  const unsigned long long hm = __ballot(head);
  const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
  start = 63 - __clzll((long long)(hm & upto));
  const unsigned long long above = hm & ~upto;
  len = (above ? __ffsll((long long)above) - 1 : 64) - lane;
  return head;
}


// (re)derives a problem's dimensions from its state's CURRENT n_kf / n_lm, grows its work buffers if needed and rebuilds the
// TwoFrame work list; called by lvf_problem_create and, every tick, by the persistent window (window.hip)
int problem_configure(lvf_problem* p);
}  // namespace lvf
// lvf_problem_solve with a caller's launches enqueued behind the last iteration and ahead of the wait that ends the solve (solver_kernels.hip).
// CONTRACT: tail(user) MUST be idempotent — pure enqueues that can be repeated (pack + copy, as window.hip's): it runs once per pass of the
// hand-over retry loop, and when a chained hand-over times out in the LAST iteration enqueued (seen only after the wait) it has already run on a
// state that is not final and runs again behind the un-chained re-run.  A tail that accumulates, consumes a buffer or enqueues once-only work
// must not be passed here.
extern "C" __attribute__((visibility("hidden"))) int lvf_problem_solve_then(lvf_problem* p, const lvf_solver_options* o, lvf_solver_summary* summary, int (*tail)(void*), void* user);
namespace lvf {
// stable LSD radix sort of (key, value) pairs by the low `key_bits` bits of the key (sort_util.hip)
int device_sort_pairs_u32(lvf_ctx* ctx, const unsigned* keys_in, unsigned* keys_out, const int* vals_in, int* vals_out, int n, int key_bits);
// the same with the key count and width on the device (written by an earlier launch on the stream): n keys of `bits` bits, digits of db bits, nbins = 1 << db
struct SortP { int n, bits, db, nbins; };
constexpr int kSortMaxDigitBits = 11;
// (one job per independent sort: its stream, its arrays, and its scratch — which must outlive the launches when the stream is the side stream)
struct SortScratch { DevBuf<int> hist; DevBuf<unsigned> tkeys; DevBuf<int> tvals; };
struct SortJobDc {
  hipStream_t q; const unsigned* keys_in; unsigned* keys_out; const int* vals_in; int* vals_out; const SortP* sp; SortScratch* keep;
  const unsigned* ki; const int* vi; unsigned* ko; int* vo;      // (cursor of the passes)
};
int device_sort_pairs_u32_dc_multi(lvf_ctx* ctx, int n_jobs, SortJobDc* jobs, int cap, int passes);
// exclusive scan (and order-preserving compaction of float4 records) in one launch, element count optionally on the device (sort_util.hip)
int device_scan1(lvf_ctx* ctx, const int* in, int cap, const int* n_dev, int* pos, int* total_out, const float4* pts, float4* out);
// the same on stream q with status lane `lane` (0: the context's stream, 1: its side stream)
int device_scan1_on(lvf_ctx* ctx, hipStream_t q, int lane, const int* in, int cap, const int* n_dev, int* pos, int* total_out, const float4* pts, float4* out);
// the context's side stream and fork / join events (created on first use)
int side_stream(lvf_ctx* ctx, hipStream_t* out);
// every extern "C" entry point starts here: selects the context's device and allocator for the calling thread
int enter(lvf_ctx* ctx);
int device_exclusive_scan_i32(lvf_ctx* ctx, const int* in, int n, int* out);
int compact_points(lvf_ctx* ctx, const float4* pts, int n, const int* flags_dev, lvf_cloud** out);
// the PCL tail of the feature extraction with the counts on the device, and what reads its state back (cloud_kernels.hip)
struct DcTailPlan { bool supported, two_streams; int cap, passes, grid_cells, lane_ground, max_iterations, min_neighbors; float leaf, radius, thr; };
int dc_tail_begin(lvf_ctx* ctx, int cap, float resolution, float max_range, DevBuf<unsigned char>& state, std::shared_ptr<void>& keep, DcTailPlan* plan);
int dc_tail_fork(lvf_ctx* ctx, const DcTailPlan& plan, hipStream_t* q);
int dc_tail_run(lvf_ctx* ctx, const DcTailPlan& plan, hipStream_t q, const float4* surf_raw, const int* n_surf_raw, const float4* ground_raw, const int* n_ground_raw,
                unsigned long long seed, DevBuf<unsigned char>& state, std::shared_ptr<void>& keep, DevBuf<float4>& surf_out, DevBuf<float4>& ground_out);
int dc_state_bytes();
int dc_state_counts_offset();      // int cnt[4] (surf voxels, surf features, ground voxels, ground features), then int err
int transform_points(lvf_ctx* ctx, const float4* in, int n, const double* pose, lvf_cloud** out);
// kernels / launchers implemented in the .hip translation units
int launch_pose_only(lvf_batch* b, const lvf_state* st, bool want_j);
int launch_two_frame(lvf_batch* b, const lvf_state* st, bool want_j);
int launch_two_camera(lvf_batch* b, const lvf_state* st, bool want_j);
int launch_lidar_normals(lvf_batch* b, const double* d_pb, const double* d_pc);
int launch_lidar_plane(lvf_batch* b, const double* rpyxyz_host, bool want_j);
int launch_imu_sqrt_info(lvf_batch* b);
int launch_imu_sqrt_info_cached(lvf_batch* b, const int* src_dev, const double* prev_dev);   // src[f] >= 0: copy prev[src[f]] instead of factoring
// A pointer MEMBER of a kernel-argument struct.  The solver's kernels take their arguments by value for one window and read them from a device
// table (one entry per window, blockIdx = window) for a batch.  A plain `double*` loaded from such a table is a GENERIC pointer to the
// compiler — it cannot know the table only ever holds global-memory addresses — so every access through it becomes a FLAT instruction:
// counted by vmcnt AND lgkmcnt (a wait for an LDS result also waits for every load in flight, and the reverse), no scalar-base addressing
// (64-bit VALU address arithmetic per access).  All batched kernels were flat-only while their single-window twins used global_*
// (ISA, round 5).  GP<T> stores the address as a global-address-space pointer on the device side (same 8 bytes, same bits), converts to
// T* where it is used, and address-space inference does the rest.  On the host it is a plain pointer.
template <typename T>
struct GP {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(1))) T* ptr_t;
#else
  typedef T* ptr_t;
#endif
  ptr_t p;
  GP() = default;
  __host__ __device__ GP(T* q) : p((ptr_t)q) {}
  __host__ __device__ operator T*() const { return (T*)p; }
  __host__ __device__ T* operator->() const { return (T*)p; }
};
static_assert(sizeof(GP<double>) == sizeof(double*), "GP<T> is a pointer");
// arrays to clear before a linearisation (one launch, or extra workgroups of another launch)
constexpr int kZeroListMax = 8;
// tri[k] > 0: entry k is a square matrix of that leading dimension of which only the LOWER triangle (widened to the 64-column block of
// the diagonal) is ever written, so only that is cleared (B and S: half of the 9.4 MB per window and iteration)
struct ZeroList { GP<double> p[kZeroListMax]; unsigned long long n[kZeroListMax]; int tri[kZeroListMax]; int count; };
// cost_stripes (optional): 32 striped accumulators that receive 1/2 |r|^2 of every factor
// zero (optional): arrays cleared by extra workgroups of the same launch
int launch_imu(lvf_batch* b, const lvf_state* st, bool want_j, double* cost_stripes = nullptr, const ZeroList* zero = nullptr);
// the ImuError evaluation as an argument block: by value for one window, as a device table (one entry per window, blockIdx.y) for a
// batch of windows.  `done` (optional) points at the window's LM control flag: a finished window skips the launch.
struct ImuOut { double* j[8]; };
struct ImuArgs {
  int n; const double *pre, *sqrt_info; const int *kf_i, *kf_j; const double *poses, *vel, *ba, *bg; double* res; ImuOut out;
  double* cost_stripes; ZeroList zero; int zero_wgs; const int* done;
};
void fill_imu_args(const lvf_batch* b, const double* poses, const double* vel, const double* ba, const double* bg, double* cost_stripes,
                   const ZeroList* zero, const int* done, ImuArgs* out);
int launch_imu_args(hipStream_t s, const ImuArgs& a, bool want_j);
int launch_imu_table(hipStream_t s, const ImuArgs* table_dev, int n_windows, int max_blocks, bool want_j);
constexpr int kImuZeroWgs = 4096;
int launch_pose_prior(lvf_batch* b, const lvf_state* st, bool want_j);
void make_camd(const lvf_camera& c, CamD& d);
}  // namespace lvf
