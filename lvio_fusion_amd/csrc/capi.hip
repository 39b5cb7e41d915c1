// capi.hip — the extern "C" surface of include/lvf.h: context, parameter state, factor batches.
#include <chrono>
#include <algorithm>
#include <vector>
#include <cstdlib>
#include "lvf_internal.hpp"

namespace lvf {

static thread_local std::string g_err;
thread_local std::shared_ptr<Pool> g_pool;

int enter(lvf_ctx* ctx) {
  if (!ctx) { set_error("null context"); return LVF_ERR_INVALID; }
  LVF_HIP(hipSetDevice(ctx->device));
  g_pool = ctx->pool;
  return LVF_OK;
}

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
  return LVF_ERR_HIP;
}

// host-side rotation of the (constant) camera extrinsic; mirrors derive_pose
static void host_rot(const double q[4], double R[9]) {
  const double s = 1.0 / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double x = s * q[0], y = s * q[1], z = s * q[2], w = s * q[3];
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}
void make_camd(const lvf_camera& c, CamD& d) {
  d.fx = c.fx; d.fy = c.fy; d.cx = c.cx; d.cy = c.cy;
  host_rot(c.extrinsic, d.Re);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d.E[3 * i + j] = d.Re[3 * j + i];
  for (int i = 0; i < 3; ++i) {
    d.te[i] = c.extrinsic[4 + i];
    d.c0[i] = -(d.E[3 * i] * c.extrinsic[4] + d.E[3 * i + 1] * c.extrinsic[5] + d.E[3 * i + 2] * c.extrinsic[6]);
  }
}

static int check_idx(const int32_t* idx, int n, const char* name) {
  for (int i = 0; i < n; ++i)
    if (idx[i] < 0) { set_error("%s[%d] = %d is negative", name, i, idx[i]); return LVF_ERR_INVALID; }
  return LVF_OK;
}
static int32_t max_idx(const int32_t* idx, int n) { int32_t m = -1; for (int i = 0; i < n; ++i) if (idx[i] > m) m = idx[i]; return m; }
static bool is_sorted_i32(const int32_t* idx, int n) { for (int i = 1; i < n; ++i) if (idx[i] < idx[i - 1]) return false; return true; }

static int alloc_outputs(lvf_batch* b) {
  LVF_TRY(b->res.alloc((size_t)b->n * b->n_res));
  for (int k = 0; k < b->n_blocks; ++k) LVF_TRY(b->jac[k].alloc((size_t)b->n * b->n_res * b->block_size[k]));
  return LVF_OK;
}

// [n][3] AoS -> [3][n] SoA
static std::vector<double> to_soa3(const double* a, int n) {
  std::vector<double> o((size_t)3 * n);
  for (int i = 0; i < n; ++i) { o[i] = a[3 * i]; o[(size_t)n + i] = a[3 * i + 1]; o[(size_t)2 * n + i] = a[3 * i + 2]; }
  return o;
}

}  // namespace lvf

using namespace lvf;

extern "C" {

const char* lvf_last_error(void) { return g_err.c_str(); }
const char* lvf_version(void) { return "lvio_fusion_amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------ context
// Pinned host blocks for callers (lvf.h): a 64-byte header in front of the user pointer says whether the block is page-locked (then it
// is parked in the process-wide HostPinPool by bucket size) or ordinary memory (no usable device: host-only tools).
namespace {
struct HostHdr { unsigned long long magic; unsigned long long bucket; int pinned; int pad[11]; };
static_assert(sizeof(HostHdr) == 64, "the user pointer keeps 64-byte alignment");
constexpr unsigned long long kHostMagic = 0x4c56465f484f5354ull;     // "LVF_HOST"
}
void* lvf_host_alloc(size_t bytes) {
  const size_t bucket = lvf::Pool::bucket(bytes + sizeof(HostHdr));
  void* base = lvf::HostPinPool::get().take(bucket);
  int pinned = 1;
  if (!base) {
    if (hipHostMalloc(&base, bucket, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      base = std::aligned_alloc(64, (bucket + 63) & ~(size_t)63);      // (the user pointer keeps the 64-byte alignment of a page-locked block)
      pinned = 0;
      if (!base) return nullptr;
    }
  }
  HostHdr* h = static_cast<HostHdr*>(base);
  h->magic = kHostMagic; h->bucket = bucket; h->pinned = pinned;
  return static_cast<char*>(base) + sizeof(HostHdr);
}
void lvf_host_free(void* p, size_t /*bytes*/) {
  if (!p) return;
  HostHdr* h = reinterpret_cast<HostHdr*>(static_cast<char*>(p) - sizeof(HostHdr));
  if (h->magic != kHostMagic) return;                   // not ours: leave it alone rather than corrupt a heap
  h->magic = 0;
  if (!h->pinned) { std::free(h); return; }
  if (!lvf::HostPinPool::get().give(h, (size_t)h->bucket)) (void)hipHostFree(h);
}

int lvf_ctx_create(int device, void* hip_stream, lvf_ctx** out) {
  LVF_REQUIRE(out, "lvf_ctx_create: out is null");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_error("lvf_ctx_create: no usable HIP device (hipGetDeviceCount -> %d, count %d); there is no CPU fallback", (int)e, count);
    (void)hipGetLastError();
    return LVF_ERR_NO_DEVICE;
  }
  LVF_REQUIRE(device >= 0 && device < count, "lvf_ctx_create: device %d out of range [0,%d)", device, count);
  LVF_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  LVF_HIP(hipGetDeviceProperties(&prop, device));
  auto* c = new lvf_ctx();
  c->device = device;
  g_pool = c->pool;
  c->pool->set_cap_from_device();
  c->num_cu = prop.multiProcessorCount;
  if (hip_stream) { c->stream = static_cast<hipStream_t>(hip_stream); c->own_stream = false; }
  else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__); }
    c->own_stream = true;
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c; set_error("lvf_ctx_create: hipEventCreate failed"); return LVF_ERR_HIP;
  }
  *out = c;
  return LVF_OK;
}
int lvf_ctx_destroy(lvf_ctx* c) {
  if (!c) return LVF_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  if (c->scan_status) (void)hipFree(c->scan_status);
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  c->pool->close();                 // objects that outlive their context free straight to the driver from now on
  if (g_pool == c->pool) g_pool.reset();
  delete c;
  return LVF_OK;
}
int lvf_ctx_synchronize(lvf_ctx* c) { LVF_REQUIRE(c, "null ctx"); LVF_HIP(hipStreamSynchronize(c->stream)); return LVF_OK; }
void* lvf_ctx_stream(lvf_ctx* c) { return c ? c->stream : nullptr; }
int lvf_timer_begin(lvf_ctx* c) { LVF_REQUIRE(c, "null ctx"); LVF_HIP(hipEventRecord(c->ev0, c->stream)); return LVF_OK; }
int lvf_timer_end(lvf_ctx* c) { LVF_REQUIRE(c, "null ctx"); LVF_HIP(hipEventRecord(c->ev1, c->stream)); return LVF_OK; }
int lvf_timer_elapsed_ms(lvf_ctx* c, float* ms) {
  LVF_REQUIRE(c && ms, "null argument");
  LVF_HIP(hipEventSynchronize(c->ev1));
  LVF_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return LVF_OK;
}

// ------------------------------------------------------------------------------------------------ box calibration
}  // extern "C"
namespace lvf {
// one wave, kCalN dependent v_fma_f64: what every latency-bound chain of the solver (pivot sweeps, back substitution) is made of
constexpr int kCalN = 4096;
__global__ __launch_bounds__(64) void k_cal_fma_chain(double* out, unsigned long long* t, double y) {
  double x = 1.0 + 1e-9 * threadIdx.x;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
#pragma unroll 64
  for (int i = 0; i < kCalN; ++i) x = fma(x, y, 1e-3);
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
  out[threadIdx.x] = x;
}
__global__ void k_cal_empty() {}
}  // namespace lvf
extern "C" {
// What the box this process landed on is worth for the latency-bound legs, so that a headline measured on two boxes can be compared
// (the driver's 3 945 it/s against the builder's 4 873 on the same commit).  out[8]:
//   [0] ns per dependent v_fma_f64 of ONE wave (100 MHz wall clock around 4 096 of them, best of 5)
//   [1] shader clocks per dependent v_fma_f64 (clock64 around the same chain)
//   [2] effective shader clock in MHz while that chain ran ([1] / [0] * 1000)
//   [3] us per EMPTY kernel launch, back to back on the context's stream (64 launches between two events, median of 5)
//   [4] us from enqueueing one empty kernel to the host seeing it finished (launch + stream wait, median of 21)
//   [5] hipDeviceAttributeClockRate in MHz (the part's rated shader clock), [6] hipDeviceAttributeMemoryClockRate in MHz, [7] compute units
int lvf_box_calibration(lvf_ctx* ctx, double* out8) {
  LVF_REQUIRE(ctx && out8, "lvf_box_calibration: null argument");
  LVF_TRY(lvf::enter(ctx));
  hipStream_t q = ctx->stream;
  for (int k = 0; k < 8; ++k) out8[k] = 0.0;
  lvf::DevBuf<double> sink; lvf::DevBuf<unsigned long long> t;
  LVF_TRY(sink.alloc(64)); LVF_TRY(t.alloc(2));
  double best_ns = 1e300, best_clk = 0.0;
  for (int rep = 0; rep < 6; ++rep) {
    hipLaunchKernelGGL(lvf::k_cal_fma_chain, dim3(1), dim3(64), 0, q, sink.p, t.p, 0.999);
    unsigned long long h[2] = {0, 0};
    LVF_HIP(hipMemcpyAsync(h, t.p, sizeof(h), hipMemcpyDeviceToHost, q));
    LVF_HIP(hipStreamSynchronize(q));
    if (rep == 0) continue;                        // (first launch: code load)
    const double ns = 10.0 * (double)h[0] / lvf::kCalN;      // wall_clock64: 100 MHz
    if (ns < best_ns) { best_ns = ns; best_clk = (double)h[1] / lvf::kCalN; }
  }
  out8[0] = best_ns; out8[1] = best_clk; out8[2] = best_ns > 0.0 ? 1e3 * best_clk / best_ns : 0.0;
  {
    std::vector<float> ms(5);
    for (float& m : ms) {
      LVF_HIP(hipEventRecord(ctx->ev0, q));
      for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(lvf::k_cal_empty, dim3(1), dim3(64), 0, q);
      LVF_HIP(hipEventRecord(ctx->ev1, q));
      LVF_HIP(hipEventSynchronize(ctx->ev1));
      LVF_HIP(hipEventElapsedTime(&m, ctx->ev0, ctx->ev1));
    }
    std::sort(ms.begin(), ms.end());
    out8[3] = 1e3 * ms[2] / 64.0;
  }
  {
    std::vector<double> us(21);
    for (double& u : us) {
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(lvf::k_cal_empty, dim3(1), dim3(64), 0, q);
      LVF_HIP(hipStreamSynchronize(q));
      u = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    std::sort(us.begin(), us.end());
    out8[4] = us[10];
  }
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, ctx->device) == hipSuccess) out8[5] = v / 1e3;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, ctx->device) == hipSuccess) out8[6] = v / 1e3;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess) out8[7] = v;
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

// ------------------------------------------------------------------------------------------------ state
// A state with its fields set in the same breath: ONE staged upload and one wait.  A NULL field takes its default (identity poses, zero
// velocities / biases / inverse depths, unit visual weights).
int lvf_state_create_from(lvf_ctx* ctx, int n_kf, int n_lm, const double* poses, const double* vel, const double* ba, const double* bg, const double* inv_depth,
                          const double* w_visual, lvf_state** out) {
  LVF_REQUIRE(ctx && out, "lvf_state_create: null ctx/out");
  LVF_REQUIRE(n_kf >= 0 && n_lm >= 0, "lvf_state_create: negative size");
  LVF_TRY(lvf::enter(ctx));
  auto* st = new lvf_state();
  st->ctx = ctx; st->n_kf = n_kf; st->n_lm = n_lm;
  int rc;
  if ((rc = st->poses.alloc((size_t)7 * n_kf)) || (rc = st->vel.alloc((size_t)3 * n_kf)) || (rc = st->ba.alloc((size_t)3 * n_kf)) ||
      (rc = st->bg.alloc((size_t)3 * n_kf)) || (rc = st->inv_depth.alloc(n_lm)) || (rc = st->w_visual.alloc(n_kf))) {
    delete st; return rc;
  }
  hipStream_t s = ctx->stream;
  const size_t k = (size_t)n_kf, l = (size_t)n_lm, total = 17 * k + l;
  if (total) {
    lvf::HostPin<double> stage;
    lvf::StreamWaitGuard guard(s);             // (the pinned block goes back to the pool only after the copies have been waited for)
    if ((rc = stage.reserve(total))) { guard.dismiss(); delete st; return rc; }
    double* h = stage.p;
    double *hp = h, *hv = h + 7 * k, *ha = hv + 3 * k, *hg = ha + 3 * k, *hw = hg + 3 * k, *hd = hw + k;
    if (poses) std::memcpy(hp, poses, 7 * k * 8); else for (size_t i = 0; i < k; ++i) { double* q = hp + 7 * i; q[0] = q[1] = q[2] = 0.0; q[3] = 1.0; q[4] = q[5] = q[6] = 0.0; }
    if (vel) std::memcpy(hv, vel, 3 * k * 8); else std::memset(hv, 0, 3 * k * 8);
    if (ba) std::memcpy(ha, ba, 3 * k * 8); else std::memset(ha, 0, 3 * k * 8);
    if (bg) std::memcpy(hg, bg, 3 * k * 8); else std::memset(hg, 0, 3 * k * 8);
    if (w_visual) std::memcpy(hw, w_visual, k * 8); else for (size_t i = 0; i < k; ++i) hw[i] = 1.0;
    if (inv_depth) std::memcpy(hd, inv_depth, l * 8); else std::memset(hd, 0, l * 8);
    hipError_t e = hipSuccess;
    auto up = [&](double* dst, const double* src, size_t n) { if (n && e == hipSuccess) e = hipMemcpyAsync(dst, src, n * 8, hipMemcpyHostToDevice, s); };
    up(st->poses.p, hp, 7 * k); up(st->vel.p, hv, 3 * k); up(st->ba.p, ha, 3 * k); up(st->bg.p, hg, 3 * k); up(st->w_visual.p, hw, k); up(st->inv_depth.p, hd, l);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { delete st; return lvf::hip_fail(e, "lvf_state_create: upload", __FILE__, __LINE__); }
    guard.dismiss();
  }
  *out = st;
  return LVF_OK;
}
int lvf_state_create(lvf_ctx* ctx, int n_kf, int n_lm, lvf_state** out) {
  return lvf_state_create_from(ctx, n_kf, n_lm, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out);
}
int lvf_state_destroy(lvf_state* st) { delete st; return LVF_OK; }

static int state_field(lvf_state* st, int field, double** p, size_t* n) {
  switch (field) {
    case LVF_POSES: *p = st->poses.p; *n = (size_t)7 * st->n_kf; return LVF_OK;
    case LVF_VEL: *p = st->vel.p; *n = (size_t)3 * st->n_kf; return LVF_OK;
    case LVF_BA: *p = st->ba.p; *n = (size_t)3 * st->n_kf; return LVF_OK;
    case LVF_BG: *p = st->bg.p; *n = (size_t)3 * st->n_kf; return LVF_OK;
    case LVF_INV_DEPTH: *p = st->inv_depth.p; *n = (size_t)st->n_lm; return LVF_OK;
    case LVF_W_VISUAL: *p = st->w_visual.p; *n = (size_t)st->n_kf; return LVF_OK;
  }
  set_error("unknown state field %d", field);
  return LVF_ERR_INVALID;
}
int lvf_state_set(lvf_state* st, int field, const double* host) {
  LVF_REQUIRE(st && host, "lvf_state_set: null argument");
  double* p; size_t n;
  LVF_TRY(state_field(st, field, &p, &n));
  if (n) LVF_TRY(lvf::copy_up_wait(st->ctx, p, host, n * 8));
  return LVF_OK;
}
// every field in ONE staged upload and one wait (lvf_state_set is a copy and a wait per field: six of them were 0.12 ms of adapt::Solve);
// any pointer may be NULL (that field is left as it is)
int lvf_state_set_all(lvf_state* st, const double* poses, const double* vel, const double* ba, const double* bg, const double* inv_depth, const double* w_visual) {
  LVF_REQUIRE(st, "lvf_state_set_all: null state");
  LVF_TRY(lvf::enter(st->ctx));
  hipStream_t q = st->ctx->stream;
  const size_t k = (size_t)st->n_kf, l = (size_t)st->n_lm;
  struct Seg { const double* src; double* dst; size_t n; } seg[6] = {{poses, st->poses.p, 7 * k}, {vel, st->vel.p, 3 * k}, {ba, st->ba.p, 3 * k}, {bg, st->bg.p, 3 * k}, {inv_depth, st->inv_depth.p, l}, {w_visual, st->w_visual.p, k}};
  size_t total = 0;
  for (const Seg& g : seg) if (g.src) total += g.n;
  if (total == 0) return LVF_OK;
  lvf::HostPin<double> stage;
  lvf::StreamWaitGuard guard(q);             // (the pinned block returns to the pool only after the copies have been waited for, on every path out)
  LVF_TRY(stage.reserve(total));
  size_t at = 0;
  for (const Seg& g : seg) if (g.src && g.n) { std::memcpy(stage.p + at, g.src, g.n * 8); LVF_HIP(hipMemcpyAsync(g.dst, stage.p + at, g.n * 8, hipMemcpyHostToDevice, q)); at += g.n; }
  LVF_HIP(hipStreamSynchronize(q));
  guard.dismiss();
  return LVF_OK;
}
// dst <- src, every field, device to device on dst's stream; nothing is waited for (a later call on the same context is ordered after it)
int lvf_state_copy(lvf_state* dst, const lvf_state* src) {
  LVF_REQUIRE(dst && src, "lvf_state_copy: null argument");
  LVF_REQUIRE(dst->n_kf == src->n_kf && dst->n_lm == src->n_lm, "lvf_state_copy: shapes differ (%d/%d keyframes, %d/%d landmarks)", dst->n_kf, src->n_kf, dst->n_lm, src->n_lm);
  LVF_TRY(lvf::enter(dst->ctx));
  hipStream_t q = dst->ctx->stream;
  const size_t k = (size_t)src->n_kf, l = (size_t)src->n_lm;
  if (k) {
    LVF_HIP(hipMemcpyAsync(dst->poses.p, src->poses.p, 7 * k * 8, hipMemcpyDeviceToDevice, q)); LVF_HIP(hipMemcpyAsync(dst->vel.p, src->vel.p, 3 * k * 8, hipMemcpyDeviceToDevice, q));
    LVF_HIP(hipMemcpyAsync(dst->ba.p, src->ba.p, 3 * k * 8, hipMemcpyDeviceToDevice, q)); LVF_HIP(hipMemcpyAsync(dst->bg.p, src->bg.p, 3 * k * 8, hipMemcpyDeviceToDevice, q));
    LVF_HIP(hipMemcpyAsync(dst->w_visual.p, src->w_visual.p, k * 8, hipMemcpyDeviceToDevice, q));
  }
  if (l) LVF_HIP(hipMemcpyAsync(dst->inv_depth.p, src->inv_depth.p, l * 8, hipMemcpyDeviceToDevice, q));
  return LVF_OK;
}
int lvf_state_get(lvf_state* st, int field, double* host) {
  LVF_REQUIRE(st && host, "lvf_state_get: null argument");
  double* p; size_t n;
  LVF_TRY(state_field(st, field, &p, &n));
  LVF_TRY(lvf::copy_down_wait(st->ctx, host, p, n * 8));
  return LVF_OK;
}

// ------------------------------------------------------------------------------------------------ batches
static lvf_batch* new_batch(lvf_ctx* ctx, int kind, int n, int n_res, std::initializer_list<int> sizes) {
  auto* b = new lvf_batch();
  b->ctx = ctx; b->kind = kind; b->n = n; b->n_res = n_res;
  for (int s : sizes) b->block_size[b->n_blocks++] = s;
  return b;
}

int lvf_pose_only_create(lvf_ctx* ctx, const lvf_camera* cam0, int n, const double* ob, const int32_t* kf_idx,
                         const int32_t* pw_idx, int n_pw, const double* pw, lvf_batch** out) {
  LVF_REQUIRE(ctx && cam0 && out, "lvf_pose_only_create: null argument");
  LVF_REQUIRE(n >= 0 && n_pw >= 0, "lvf_pose_only_create: negative size");
  LVF_REQUIRE(n == 0 || (ob && kf_idx && pw_idx && pw), "lvf_pose_only_create: null input array");
  LVF_TRY(check_idx(kf_idx, n, "kf_idx")); LVF_TRY(check_idx(pw_idx, n, "pw_idx"));
  LVF_REQUIRE(max_idx(pw_idx, n) < n_pw, "lvf_pose_only_create: pw_idx out of range (n_pw=%d)", n_pw);
  LVF_TRY(lvf::enter(ctx));
  lvf_batch* b = new_batch(ctx, LVF_K_POSE_ONLY, n, 2, {7});
  make_camd(*cam0, b->cam_a);
  b->n_table = n_pw;
  b->sorted_by_kf = is_sorted_i32(kf_idx, n);
  b->min_n_kf = max_idx(kf_idx, n) + 1;
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = b->ob_a.upload(ob, (size_t)2 * n, s)) || (rc = b->idx_a.upload(kf_idx, n, s)) || (rc = b->idx_b.upload(pw_idx, n, s)) ||
      (rc = b->table.upload(pw, (size_t)3 * n_pw, s)) || (rc = alloc_outputs(b))) { delete b; return rc; }
  LVF_HIP(hipStreamSynchronize(s));
  *out = b;
  return LVF_OK;
}

int lvf_two_frame_create(lvf_ctx* ctx, const lvf_camera* left, const lvf_camera* right, int n, const double* first_ob,
                         const double* ob, const int32_t* lm_idx, const int32_t* kf1_idx, const int32_t* kf2_idx,
                         lvf_batch** out) {
  LVF_REQUIRE(ctx && left && right && out, "lvf_two_frame_create: null argument");
  LVF_REQUIRE(n >= 0, "lvf_two_frame_create: negative size");
  LVF_REQUIRE(n == 0 || (first_ob && ob && lm_idx && kf1_idx && kf2_idx), "lvf_two_frame_create: null input array");
  LVF_TRY(check_idx(lm_idx, n, "lm_idx")); LVF_TRY(check_idx(kf1_idx, n, "kf1_idx")); LVF_TRY(check_idx(kf2_idx, n, "kf2_idx"));
  LVF_TRY(lvf::enter(ctx));
  lvf_batch* b = new_batch(ctx, LVF_K_TWO_FRAME, n, 2, {1, 7, 7});
  make_camd(*left, b->cam_a); make_camd(*right, b->cam_b);
  b->sorted_by_kf = is_sorted_i32(kf2_idx, n);
  b->host_kf1.assign(kf1_idx, kf1_idx + n); b->host_kf2.assign(kf2_idx, kf2_idx + n); b->host_lm.assign(lm_idx, lm_idx + n);
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = b->ob_a.upload(first_ob, (size_t)2 * n, s)) || (rc = b->ob_b.upload(ob, (size_t)2 * n, s)) ||
      (rc = b->idx_a.upload(lm_idx, n, s)) || (rc = b->idx_b.upload(kf1_idx, n, s)) || (rc = b->idx_c.upload(kf2_idx, n, s)) ||
      (rc = alloc_outputs(b))) { delete b; return rc; }
  b->min_n_kf = std::max(max_idx(kf1_idx, n), max_idx(kf2_idx, n)) + 1;
  b->min_n_lm = max_idx(lm_idx, n) + 1;
  LVF_HIP(hipStreamSynchronize(s));
  *out = b;
  return LVF_OK;
}

int lvf_two_camera_create(lvf_ctx* ctx, const lvf_camera* left, const lvf_camera* right, int n, const double* left_ob,
                          const double* right_ob, const int32_t* lm_idx, const int32_t* kf_idx, lvf_batch** out) {
  LVF_REQUIRE(ctx && left && right && out, "lvf_two_camera_create: null argument");
  LVF_REQUIRE(n >= 0, "lvf_two_camera_create: negative size");
  LVF_REQUIRE(n == 0 || (left_ob && right_ob && lm_idx && kf_idx), "lvf_two_camera_create: null input array");
  LVF_TRY(check_idx(lm_idx, n, "lm_idx")); LVF_TRY(check_idx(kf_idx, n, "kf_idx"));
  LVF_TRY(lvf::enter(ctx));
  lvf_batch* b = new_batch(ctx, LVF_K_TWO_CAMERA, n, 2, {1});
  make_camd(*left, b->cam_a); make_camd(*right, b->cam_b);
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = b->ob_a.upload(left_ob, (size_t)2 * n, s)) || (rc = b->ob_b.upload(right_ob, (size_t)2 * n, s)) ||
      (rc = b->idx_a.upload(lm_idx, n, s)) || (rc = b->idx_b.upload(kf_idx, n, s)) || (rc = alloc_outputs(b))) { delete b; return rc; }
  b->min_n_kf = max_idx(kf_idx, n) + 1;
  b->min_n_lm = max_idx(lm_idx, n) + 1;
  LVF_HIP(hipStreamSynchronize(s));
  *out = b;
  return LVF_OK;
}

// The caller VOUCHES for the shape Backend::BuildProblem's TwoFrame blocks always have (backend.cpp:105-140: frames in time order, a frame's
// features in one pass, landmarks keyed by id) and hands the blocks-per-current-keyframe counts:
//   * the blocks are sorted by current keyframe (kf2 ascending);            * first keyframe < current keyframe in every block;
//   * at most one block per (landmark, current keyframe);                   * every landmark has ONE first keyframe.
// lvf_problem_create then skips its own host pass over the blocks (0.3 ms of adapt::Solve at 72 k blocks).  A false claim gives wrong results
// (plain stores into a landmark's E row meeting adds): callers that cannot prove it — include/lvf_ceres_adapter.hpp proves it block by block
// while the problem is being built — do not call this.
int lvf_two_frame_set_shape(lvf_batch* b, int n_kf, const int32_t* blocks_per_kf2) {
  LVF_REQUIRE(b && b->kind == LVF_K_TWO_FRAME, "lvf_two_frame_set_shape: not a two-frame batch");
  LVF_REQUIRE(n_kf >= b->min_n_kf && (n_kf == 0 || blocks_per_kf2), "lvf_two_frame_set_shape: n_kf %d below the batch's keyframe indices (%d)", n_kf, b->min_n_kf);
  long long total = 0;
  for (int k = 0; k < n_kf; ++k) { LVF_REQUIRE(blocks_per_kf2[k] >= 0, "lvf_two_frame_set_shape: negative count"); total += blocks_per_kf2[k]; }
  LVF_REQUIRE(total == (long long)b->n, "lvf_two_frame_set_shape: the counts add up to %lld, the batch has %d blocks", total, b->n);
  // (cheap necessary conditions on the host copies the batch keeps: a caller's slip should fail here, not as a wrong solve)
  if (!b->host_kf2.empty()) {
    int at = 0;
    for (int k = 0; k < n_kf; ++k) {
      const int c = blocks_per_kf2[k];
      if (c > 0) LVF_REQUIRE(b->host_kf2[at] == k && b->host_kf2[at + c - 1] == k, "lvf_two_frame_set_shape: run %d is not the blocks of current keyframe %d", k, k);
      at += c;
    }
  }
  b->kf2_counts.assign(blocks_per_kf2, blocks_per_kf2 + n_kf);
  b->sorted_by_kf = true; b->unique_lk2_known = true;
  return LVF_OK;
}

// per-block weights of a TwoCamera batch = the `weight` constructor argument of each TwoCameraReprojectionError (visual_error.hpp:112;
// backend.cpp:123 passes 5 * frame->weights.visual).  Without this call the kernels use 5 * w_visual[kf_idx[i]].
int lvf_two_camera_set_block_weights(lvf_batch* b, const double* weight) {
  LVF_REQUIRE(b && b->kind == LVF_K_TWO_CAMERA, "lvf_two_camera_set_block_weights: not a two-camera batch");
  LVF_TRY(lvf::enter(b->ctx));
  if (!weight || b->n == 0) { b->wblk.n = 0; return LVF_OK; }
  LVF_TRY(b->wblk.assign(weight, (size_t)b->n, b->ctx->stream));
  LVF_HIP(hipStreamSynchronize(b->ctx->stream));
  b->evaluated = false;
  return LVF_OK;
}

int lvf_imu_create(lvf_ctx* ctx, int n, const lvf_preint* pre, const int32_t* kf_i, const int32_t* kf_j, lvf_batch** out) {
  LVF_REQUIRE(ctx && out, "lvf_imu_create: null argument");
  LVF_REQUIRE(n >= 0, "lvf_imu_create: negative size");
  LVF_REQUIRE(n == 0 || (pre && kf_i && kf_j), "lvf_imu_create: null input array");
  static_assert(sizeof(lvf_preint) == 467 * sizeof(double), "lvf_preint must be 467 packed doubles");
  LVF_TRY(check_idx(kf_i, n, "kf_i")); LVF_TRY(check_idx(kf_j, n, "kf_j"));
  for (int f = 0; f < n; ++f) LVF_REQUIRE(kf_i[f] != kf_j[f], "lvf_imu_create: factor %d links keyframe %d to itself", f, kf_i[f]);
  LVF_TRY(lvf::enter(ctx));
  lvf_batch* b = new_batch(ctx, LVF_K_IMU, n, 15, {7, 3, 3, 3, 7, 3, 3, 3});
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = b->pre.upload(reinterpret_cast<const double*>(pre), (size_t)467 * n, s)) || (rc = b->idx_a.upload(kf_i, n, s)) ||
      (rc = b->idx_b.upload(kf_j, n, s)) || (rc = b->sqrt_info.alloc((size_t)225 * n)) || (rc = alloc_outputs(b)) ||
      (rc = launch_imu_sqrt_info(b))) { delete b; return rc; }
  b->min_n_kf = std::max(max_idx(kf_i, n), max_idx(kf_j, n)) + 1;
  if (n > 0) { b->host_kf1.assign(kf_i, kf_i + n); b->host_kf2.assign(kf_j, kf_j + n); }   // the solver's elimination order follows the IMU chain
  LVF_HIP(hipStreamSynchronize(s));
  *out = b;
  return LVF_OK;
}

int lvf_lidar_plane_create(lvf_ctx* ctx, int mode, int n, const double* p, const double* pa, const double* pb,
                           const double* pc, const double* Twc1, double weight, lvf_batch** out) {
  LVF_REQUIRE(ctx && out && Twc1, "lvf_lidar_plane_create: null argument");
  LVF_REQUIRE(mode == 0 || mode == 1, "lvf_lidar_plane_create: mode must be 0 (RPZ) or 1 (YXY)");
  LVF_REQUIRE(n >= 0, "lvf_lidar_plane_create: negative size");
  LVF_REQUIRE(n == 0 || (p && pa && pb && pc), "lvf_lidar_plane_create: null input array");
  LVF_TRY(lvf::enter(ctx));
  lvf_batch* b = new_batch(ctx, LVF_K_LIDAR, n, 1, {1, 1, 1});
  b->lidar_mode = mode; b->lidar_weight = weight;
  std::memcpy(b->Twc1, Twc1, sizeof(b->Twc1));
  hipStream_t s = ctx->stream;
  DevBuf<double> dpb, dpc;
  const auto sp = to_soa3(p, n), spa = to_soa3(pa, n), spb = to_soa3(pb, n), spc = to_soa3(pc, n);
  int rc;
  if ((rc = b->lp.upload(sp.data(), sp.size(), s)) || (rc = b->lpa.upload(spa.data(), spa.size(), s)) ||
      (rc = dpb.upload(spb.data(), spb.size(), s)) || (rc = dpc.upload(spc.data(), spc.size(), s)) ||
      (rc = b->lnrm.alloc((size_t)3 * n)) || (rc = alloc_outputs(b)) || (rc = launch_lidar_normals(b, dpb.p, dpc.p))) { delete b; return rc; }
  LVF_HIP(hipStreamSynchronize(s));
  *out = b;
  return LVF_OK;
}

int lvf_batch_destroy(lvf_batch* b) { delete b; return LVF_OK; }
int lvf_batch_size(const lvf_batch* b) { return b ? b->n : -1; }
int lvf_batch_num_param_blocks(const lvf_batch* b) { return b ? b->n_blocks : -1; }

int lvf_batch_evaluate(lvf_batch* b, const lvf_state* st, const double* rpyxyz, int want_jacobians) {
  LVF_REQUIRE(b, "lvf_batch_evaluate: null batch");
  LVF_TRY(lvf::enter(b->ctx));
  const bool wj = want_jacobians != 0;
  int rc = LVF_OK;
  if (b->kind == LVF_K_LIDAR) {
    LVF_REQUIRE(rpyxyz, "lvf_batch_evaluate: lidar batches need the live rpyxyz[6]");
    rc = launch_lidar_plane(b, rpyxyz, wj);
  } else {
    LVF_REQUIRE(st, "lvf_batch_evaluate: state is null");
    LVF_REQUIRE(st->ctx == b->ctx, "lvf_batch_evaluate: state and batch belong to different contexts");
    switch (b->kind) {
      case LVF_K_POSE_ONLY:
        LVF_REQUIRE(st->n_kf >= b->min_n_kf, "pose-only batch references keyframe %d but the state has %d", b->min_n_kf - 1, st->n_kf);
        rc = launch_pose_only(b, st, wj); break;
      case LVF_K_TWO_FRAME:
        LVF_REQUIRE(st->n_kf >= b->min_n_kf && st->n_lm >= b->min_n_lm, "two-frame batch indices exceed the state (n_kf=%d n_lm=%d)", st->n_kf, st->n_lm);
        rc = launch_two_frame(b, st, wj); break;
      case LVF_K_TWO_CAMERA:
        LVF_REQUIRE(st->n_kf >= b->min_n_kf && st->n_lm >= b->min_n_lm, "two-camera batch indices exceed the state (n_kf=%d n_lm=%d)", st->n_kf, st->n_lm);
        rc = launch_two_camera(b, st, wj); break;
      case LVF_K_IMU:
        LVF_REQUIRE(st->n_kf >= b->min_n_kf, "imu batch indices exceed the state (n_kf=%d)", st->n_kf);
        rc = launch_imu(b, st, wj); break;
      case LVF_K_POSE_PRIOR:
        LVF_REQUIRE(st->n_kf >= b->min_n_kf, "pose-prior batch indices exceed the state (n_kf=%d)", st->n_kf);
        rc = launch_pose_prior(b, st, wj); break;
      default: set_error("unknown batch kind %d", b->kind); return LVF_ERR_INVALID;
    }
  }
  if (rc == LVF_OK) { b->evaluated = true; b->have_jac = wj; }
  return rc;
}

static int download(lvf_batch* b, const double* dev, double* host, size_t count) {
  if (count) LVF_HIP(hipMemcpyAsync(host, dev, count * 8, hipMemcpyDeviceToHost, b->ctx->stream));
  LVF_HIP(hipStreamSynchronize(b->ctx->stream));
  return LVF_OK;
}
int lvf_batch_download_residuals(lvf_batch* b, double* host) {
  LVF_REQUIRE(b && host, "lvf_batch_download_residuals: null argument");
  if (!b->evaluated) { set_error("download before evaluate"); return LVF_ERR_STATE; }
  return download(b, b->res.p, host, (size_t)b->n * b->n_res);
}
int lvf_batch_download_jacobian(lvf_batch* b, int block, double* host) {
  LVF_REQUIRE(b && host, "lvf_batch_download_jacobian: null argument");
  LVF_REQUIRE(block >= 0 && block < b->n_blocks, "jacobian block %d out of range [0,%d)", block, b->n_blocks);
  if (!b->evaluated || !b->have_jac) { set_error("jacobians were not evaluated"); return LVF_ERR_STATE; }
  return download(b, b->jac[block].p, host, (size_t)b->n * b->n_res * b->block_size[block]);
}
int lvf_batch_download_normals(lvf_batch* b, double* host) {
  LVF_REQUIRE(b && host, "lvf_batch_download_normals: null argument");
  LVF_REQUIRE(b->kind == LVF_K_LIDAR, "normals exist only for lidar batches");
  std::vector<double> soa((size_t)3 * b->n);
  LVF_TRY(download(b, b->lnrm.p, soa.data(), soa.size()));
  for (int i = 0; i < b->n; ++i) { host[3 * i] = soa[i]; host[3 * i + 1] = soa[(size_t)b->n + i]; host[3 * i + 2] = soa[(size_t)2 * b->n + i]; }
  return LVF_OK;
}
void* lvf_batch_residuals_dev(lvf_batch* b) { return b ? b->res.p : nullptr; }
void* lvf_batch_jacobian_dev(lvf_batch* b, int block) { return (b && block >= 0 && block < b->n_blocks) ? b->jac[block].p : nullptr; }

}  // extern "C"
