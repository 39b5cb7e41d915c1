// extract_kernels.hip — LiDAR feature extraction of one keyframe scan on device.
//
// Replaces FeatureAssociation::Process (src/lvio_fusion/src/association.cpp:86-235) with everything it calls:
//   Preprocess (NaN + range gate)                                  association.cpp:97-102, include/lvio_fusion/utility.h:70-90
//   ImageProjection::Process: range-image projection, ground marking, BFS segmentation, segmented-cloud extraction
//                                                                  src/projection.cpp:26-320
//   AdjustDistortion (relative time into the intensity channel), CalculateSmoothness, ExtractFeatures' picks
//                                                                  association.cpp:104-208
//   VoxelGrid / RadiusOutlierRemoval / SegmentGround / Sensor2Robot tail   association.cpp:210-247  (cloud_kernels.hip)
// The reference's sequential constructs map to data-parallel ones with identical results:
//   * "later point overwrites the pixel" (projection.cpp:91-96)            -> atomicMax of the point index per pixel
//   * per-column ground sweep with overwriting -1/1 writes (:110-135)      -> closed form on the pair (i-1,i) / (i,i+1) states
//   * BFS LabelComponents (:200-320): the link test is symmetric, so segments are the connected components of an undirected
//     graph -> lock-free union-find (hook larger root under smaller: the root IS the BFS seed, the first pixel in row-major
//     order); "rows touched by pushed neighbours" = rows of all pixels but the seed -> 64-bit row mask by atomicOr
//   * running counters of Segment() (:163-197)                              -> exclusive prefix sum in row-major order
//   * AdjustDistortion's `half_passed` latch (association.cpp:125-131)     -> prefix-OR of the latch condition
// Float arithmetic follows the reference's float/double promotions; sin/cos of the two constant angles and the start/end
// orientations are taken on the HOST; every atan2 of floats — host or device — is cr_atan2f (cr_math.hpp: correctly rounded, one
// operation sequence for both sides), sqrtf is IEEE on both: the per-pixel decisions are reproducible bit for bit.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#include "cr_math.hpp"
#include "lvf_internal.hpp"

#pragma clang fp contract(off)

namespace lvf {

constexpr int kE = 256;
static inline int gride(long long n) { return (int)((std::max(n, 1ll) + kE - 1) / kE); }
constexpr double kPi = 3.14159265358979323846;

struct ExP {
  int R, Cn, ground_rows;
  float ang_res_x, ang_res_y, ang_bottom;
  float sin_ax, cos_ax, sin_ay, cos_ay, theta;
  float min2, max2;
};

__global__ __launch_bounds__(kE) void k_ex_preflag(int n, const float* __restrict__ src, int stride, float min2, float max2, float4* __restrict__ packed,
                                                   int* __restrict__ flags) {
  const int i = blockIdx.x * kE + threadIdx.x;
  if (i >= n) return;
  const float* s = src + (size_t)i * stride;
  const float x = s[0], y = s[1], z = s[2];
  packed[i] = make_float4(x, y, z, 0.0f);
  const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
  const float d = (x * x + y * y) + z * z;
  flags[i] = (fin && d > min2 && d < max2) ? 1 : 0;
}

__device__ __forceinline__ float deg_of(float rad) { return (float)((double)(rad * 180) / kPi); }   // `x * 180 / M_PI` with x float

__device__ __forceinline__ bool pixel_of(const float4 p, const ExP P, int& row, int& col) {
  const float va = deg_of(cr_atan2f(p.z, sqrtf(p.x * p.x + p.y * p.y)));
  row = (int)((va + P.ang_bottom) / P.ang_res_y);
  if (row < 0 || row >= P.R) return false;
  const float ha = deg_of(cr_atan2f(p.x, p.y));
  col = (int)(-round(((double)ha - 90.0) / (double)P.ang_res_x) + (double)(P.Cn / 2));
  if (col >= P.Cn) col -= P.Cn;
  return !(col < 0 || col >= P.Cn);
}
// pixel_src = -1, size = 0, rows = 0 in one launch (three fills otherwise)
__global__ __launch_bounds__(kE) void k_ex_clear(int npix, int* __restrict__ pixel_src, int* __restrict__ size, unsigned long long* __restrict__ rows) {
  const int i = blockIdx.x * kE + threadIdx.x;
  if (i >= npix) return;
  pixel_src[i] = -1; size[i] = 0; rows[i] = 0ull;
}
// (m_dev, here and below: the point count as an earlier launch left it on the device — the device-counted path — clips the by-value one)
__global__ __launch_bounds__(kE) void k_ex_project(int m, const int* __restrict__ m_dev, const float4* __restrict__ pts, ExP P, int* __restrict__ pixel_src) {
  const int i = blockIdx.x * kE + threadIdx.x;
  if (m_dev) m = min(m, *m_dev);
  if (i >= m) return;
  int row, col;
  if (pixel_of(pts[i], P, row, col)) atomicMax(pixel_src + (size_t)row * P.Cn + col, i);   // the last point in scan order wins
}
__global__ __launch_bounds__(kE) void k_ex_fill(int npix, const float4* __restrict__ pts, ExP P, const int* __restrict__ pixel_src, float4* __restrict__ full,
                                                float* __restrict__ range) {
  const int idx = blockIdx.x * kE + threadIdx.x;
  if (idx >= npix) return;
  const int src = pixel_src[idx];
  if (src < 0) { full[idx] = make_float4(NAN, NAN, NAN, -1.0f); range[idx] = FLT_MAX; return; }
  const float4 p = pts[src];
  const int row = idx / P.Cn, col = idx - row * P.Cn;
  range[idx] = sqrtf((p.x * p.x + p.y * p.y) + p.z * p.z);
  full[idx] = make_float4(p.x, p.y, p.z, (float)((double)(float)row + (double)(float)col / 10000.0));
}

// pair (i, i+1) of column j: 0 = a point is missing, 1 = |slope| <= 10 deg, 2 = steeper
__device__ __forceinline__ int pair_state(const float4* __restrict__ full, int Cn, int i, int j) {
  const float4 lo = full[(size_t)i * Cn + j], up = full[(size_t)(i + 1) * Cn + j];
  if (lo.w == -1.0f || up.w == -1.0f) return 0;
  const float dx = up.x - lo.x, dy = up.y - lo.y, dz = up.z - lo.z;
  const float angle = deg_of(cr_atan2f(dz, sqrtf(dx * dx + dy * dy)));
  return fabsf(angle) <= 10.0f ? 1 : 2;
}
// ground_mat == 1 after the reference's sequential sweep, and the initial label (-1 = ground or empty, 0 = to be segmented)
__global__ __launch_bounds__(kE) void k_ex_ground(int npix, const float4* __restrict__ full, const float* __restrict__ range, ExP P,
                                                  signed char* __restrict__ ground, int* __restrict__ parent) {
  const int idx = blockIdx.x * kE + threadIdx.x;
  if (idx >= npix) return;
  const int i = idx / P.Cn, j = idx - i * P.Cn;
  bool g = false;
  if (i < P.ground_rows) {
    const int s = pair_state(full, P.Cn, i, j);
    if (s == 1) g = true;
    else if (s == 2) g = i > 0 && pair_state(full, P.Cn, i - 1, j) == 1;
  } else if (i == P.ground_rows) {
    g = pair_state(full, P.Cn, i - 1, j) == 1;
  }
  ground[idx] = g ? 1 : 0;
  parent[idx] = (g || range[idx] == FLT_MAX) ? -1 : idx;
}

__device__ __forceinline__ int uf_find(const int* parent, int a) {
  int p = parent[a];
  while (p != a) { a = p; p = parent[a]; }
  return a;
}
__device__ __forceinline__ void uf_unite(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(parent + b, a);       // hook the larger root under the smaller one
    if (old == b) return;
    b = old;
  }
}
__device__ __forceinline__ bool linked(float ra, float rb, float s, float c, float theta) {
  const float d1 = fmaxf(ra, rb), d2 = fminf(ra, rb);
  return cr_atan2f(d2 * s, d1 - d2 * c) > theta;
}
__global__ __launch_bounds__(kE) void k_ex_union(int npix, const float* __restrict__ range, ExP P, int* __restrict__ parent) {
  const int idx = blockIdx.x * kE + threadIdx.x;
  if (idx >= npix || parent[idx] < 0) return;
  const int i = idx / P.Cn, j = idx - i * P.Cn;
  const int jr = j + 1 == P.Cn ? 0 : j + 1;                               // the range image wraps in azimuth
  const int right = i * P.Cn + jr;
  const bool has_down = i + 1 < P.R;
  const int down = has_down ? idx + P.Cn : idx;
  // (both neighbours' state and range are requested before either link is tested: behind the short-circuits they were four dependent round trips)
  const float ra = range[idx], rr = range[right], rd = range[down];
  const int pr = parent[right], pd = parent[down];
  const bool link_r = (right != idx) & (pr >= 0), link_d = has_down & (pd >= 0);
  if (link_r && linked(ra, rr, P.sin_ax, P.cos_ax, P.theta)) uf_unite(parent, idx, right);
  if (link_d && linked(ra, rd, P.sin_ay, P.cos_ay, P.theta)) uf_unite(parent, idx, down);
}
// segment size and the 64-bit mask of rows its pixels (other than the seed) lie in.  A wave holds 64 consecutive pixels of one
// or two image rows and neighbouring pixels mostly share their segment, so the root's counters are touched once per run of
// same-root lanes (cell_runs): per-pixel atomics on the few large segments' roots cost 138 us per scan.
__global__ __launch_bounds__(kE) void k_ex_stats(int npix, int Cn, int* __restrict__ parent, int* __restrict__ size, unsigned long long* __restrict__ rows) {
  const int idx = blockIdx.x * kE + threadIdx.x;
  const bool live = idx < npix && parent[idx] >= 0;
  const int root = live ? uf_find(parent, idx) : -1;
  if (live) parent[idx] = root;        // (path compression for k_ex_segflag's look-up: roots do not change any more, a concurrent walk only gets shorter)
  if (Cn < 64) {                     // a wave could span more than two rows: plain per-pixel counters
    if (live) { atomicAdd(size + root, 1); if (idx != root) atomicOr(rows + root, 1ull << (idx / Cn)); }
    return;
  }
  int start, len;
  const bool head = cell_runs(root, start, len);
  // row bits of the run: every lane contributes its own row unless it is the seed pixel itself
  const unsigned long long mine = (live && idx != root) ? (1ull << (idx / Cn)) : 0ull;
  // a run spans at most two rows: OR of the first and last contributing lanes' bits is not enough when the seed sits at either
  // end, so take the OR over the whole run with two ballots (one per possible row)
  const int row0 = (blockIdx.x * kE + (threadIdx.x & ~63)) / Cn;           // row of the wave's first pixel
  const unsigned long long in_r0 = __ballot(mine != 0ull && idx / Cn == row0);
  const unsigned long long in_r1 = __ballot(mine != 0ull && idx / Cn == row0 + 1);
  if (head && root >= 0) {
    const int lane = threadIdx.x & 63;
    const unsigned long long run = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
    atomicAdd(size + root, len);
    unsigned long long bits = 0ull;
    if (in_r0 & run) bits |= 1ull << row0;
    if (in_r1 & run) bits |= 1ull << (row0 + 1);
    if (bits) atomicOr(rows + root, bits);
  }
}
__global__ __launch_bounds__(kE) void k_ex_segflag(int npix, const int* __restrict__ parent, const int* __restrict__ size, const unsigned long long* __restrict__ rows,
                                                   const signed char* __restrict__ ground, int* __restrict__ flags, int* __restrict__ label) {
  const int idx = blockIdx.x * kE + threadIdx.x;
  if (idx >= npix) return;
  int lab = -1;
  if (parent[idx] >= 0) {
    const int root = uf_find(parent, idx);
    const int sz = size[root];
    const bool feasible = sz >= 30 || (sz >= 5 && __popcll(rows[root]) >= 3);
    lab = feasible ? root + 1 : 999999;
  }
  label[idx] = lab;
  flags[idx] = (ground[idx] == 1 || (lab > 0 && lab != 999999)) ? 1 : 0;
}
__global__ __launch_bounds__(kE) void k_ex_segemit(int npix, int Cn, const int* __restrict__ flags, const int* __restrict__ pos, const float4* __restrict__ full,
                                                   const float* __restrict__ range, const signed char* __restrict__ ground, float4* __restrict__ seg,
                                                   float* __restrict__ seg_range, int* __restrict__ seg_ground, int* __restrict__ seg_row) {
  const int idx = blockIdx.x * kE + threadIdx.x;
  if (idx >= npix || !flags[idx]) return;
  const int k = pos[idx];
  seg[k] = full[idx]; seg_range[k] = range[idx]; seg_ground[k] = ground[idx] == 1 ? 1 : 0; seg_row[k] = idx / Cn;
}

// AdjustDistortion: the latch condition evaluated under "!half_passed" (valid up to and including the first true)
struct OriP { float start, end, diff; double cycle; };
__device__ __forceinline__ float ori_first_half(float x, float y, OriP o, bool& latch) {
  float ori = -cr_atan2f(y, x);
  if ((double)ori < (double)o.start - kPi / 2) ori = (float)((double)ori + 2 * kPi);
  else if ((double)ori > (double)o.start + kPi * 3 / 2) ori = (float)((double)ori - 2 * kPi);
  latch = (double)(ori - o.start) > kPi;
  return ori;
}
// FindStartEndAngle (projection.cpp:42-56): the reference's float / double promotions, for the host-counted path on the host and for the
// device-counted one in a one-thread launch
__host__ __device__ inline OriP find_start_end(const float4 first, const float4 last, double cycle) {
  OriP o;
  float start = -cr_atan2f(first.y, first.x);
  float end = -cr_atan2f(last.y, last.x) + 2 * kPi;
  if (end - start > 3 * kPi) end -= 2 * kPi;
  else if (end - start < kPi) end += 2 * kPi;
  o.start = start; o.end = end; o.diff = end - start; o.cycle = cycle;
  return o;
}
// what the device-counted path keeps on the device between launches
struct ExDev { int cnt[4]; /* filtered, segmented, ground picks, surf picks */ };
// the device-counted path's source of the orientations: the filtered cloud's first and last point (count on the device)
struct OriSrc { const float4* filtered; const int* m_dev; double cycle; };
__device__ __forceinline__ OriP ori_of(const OriSrc& src) {
  const int m = *src.m_dev;
  const float4 unit = make_float4(1, 0, 0, 0);
  return find_start_end(m ? src.filtered[0] : unit, m ? src.filtered[m - 1] : unit, src.cycle);
}
__global__ __launch_bounds__(kE) void k_ex_latch(int m, const int* __restrict__ m_dev, const float4* __restrict__ seg, OriP o, const OriSrc op,
                                                 int* __restrict__ latch) {
  const int i = blockIdx.x * kE + threadIdx.x;
  if (m_dev) m = min(m, *m_dev);
  if (i >= m) return;
  if (op.m_dev) o = ori_of(op);
  bool l;
  (void)ori_first_half(seg[i].x, seg[i].y, o, l);
  latch[i] = l ? 1 : 0;
}
__device__ __forceinline__ void reltime_body(const int i, float4* __restrict__ seg, const OriP o, const int* __restrict__ latch_before);
__global__ __launch_bounds__(kE) void k_ex_reltime(int m, const int* __restrict__ m_dev, float4* __restrict__ seg, OriP o, const OriSrc op,
                                                   const int* __restrict__ latch_before) {
  const int i = blockIdx.x * kE + threadIdx.x;
  if (m_dev) m = min(m, *m_dev);
  if (i >= m) return;
  if (op.m_dev) o = ori_of(op);
  reltime_body(i, seg, o, latch_before);
}
__device__ __forceinline__ void reltime_body(const int i, float4* __restrict__ seg, const OriP o, const int* __restrict__ latch_before) {
  float4 p = seg[i];
  float ori;
  if (latch_before[i] == 0) { bool l; ori = ori_first_half(p.x, p.y, o, l); }
  else {
    ori = -cr_atan2f(p.y, p.x);
    ori = (float)((double)ori + 2 * kPi);
    if ((double)ori < (double)o.end - kPi * 3 / 2) ori = (float)((double)ori + 2 * kPi);
    else if ((double)ori > (double)o.end + kPi / 2) ori = (float)((double)ori - 2 * kPi);
  }
  const float rel = (ori - o.start) / o.diff;
  p.w = (float)((double)(int)p.w + o.cycle * (double)rel);   // int(intensity) + cycle_time_ (double) * rel_time
  seg[i] = p;
}

// CalculateSmoothness + ExtractFeatures' sector test for segmented point k
__device__ __forceinline__ void pick_body(const int k, const int m, int Cn, const float* __restrict__ rg, const int* __restrict__ seg_ground, const int* __restrict__ seg_row,
                                          const int* __restrict__ pixpos, int* __restrict__ pick_ground, int* __restrict__ pick_surf, float* __restrict__ curv_out);
__global__ __launch_bounds__(kE) void k_ex_pick(int m, const int* __restrict__ m_dev, int Cn, const float4* __restrict__ seg, const float* __restrict__ rg, const int* __restrict__ seg_ground,
                                                const int* __restrict__ seg_row, const int* __restrict__ pixpos /* exclusive prefix over pixels */,
                                                int* __restrict__ pick_ground, int* __restrict__ pick_surf, float* __restrict__ curv_out) {
  const int k = blockIdx.x * kE + threadIdx.x;
  if (m_dev) m = min(m, *m_dev);
  if (k >= m) return;
  pick_body(k, m, Cn, rg, seg_ground, seg_row, pixpos, pick_ground, pick_surf, curv_out);
}
// the picks of point k and its relative time in one launch (device-counted path: the two touch different arrays)
__global__ __launch_bounds__(kE) void k_ex_pick_reltime(int m, const int* __restrict__ m_dev, int Cn, float4* __restrict__ seg, const float* __restrict__ rg,
                                                        const int* __restrict__ seg_ground, const int* __restrict__ seg_row, const int* __restrict__ pixpos,
                                                        int* __restrict__ pick_ground, int* __restrict__ pick_surf, float* __restrict__ curv_out, const OriSrc op,
                                                        const int* __restrict__ latch_before) {
  const int k = blockIdx.x * kE + threadIdx.x;
  m = min(m, *m_dev);
  if (k >= m) return;
  pick_body(k, m, Cn, rg, seg_ground, seg_row, pixpos, pick_ground, pick_surf, curv_out);
  reltime_body(k, seg, ori_of(op), latch_before);
}
// the segmented cloud's records and, for the device-counted path, the latch condition of each (k_ex_latch's value: the same point)
__global__ __launch_bounds__(kE) void k_ex_segemit_latch(int npix, int Cn, const int* __restrict__ flags, const int* __restrict__ pos, const float4* __restrict__ full,
                                                         const float* __restrict__ range, const signed char* __restrict__ ground, float4* __restrict__ seg,
                                                         float* __restrict__ seg_range, int* __restrict__ seg_ground, int* __restrict__ seg_row, const OriSrc op,
                                                         int* __restrict__ latch) {
  const int idx = blockIdx.x * kE + threadIdx.x;
  if (idx >= npix || !flags[idx]) return;
  const int k = pos[idx];
  const float4 p = full[idx];
  seg[k] = p; seg_range[k] = range[idx]; seg_ground[k] = ground[idx] == 1 ? 1 : 0; seg_row[k] = idx / Cn;
  bool l;
  (void)ori_first_half(p.x, p.y, ori_of(op), l);
  latch[k] = l ? 1 : 0;
}
__device__ __forceinline__ void pick_body(const int k, const int m, int Cn, const float* __restrict__ rg, const int* __restrict__ seg_ground, const int* __restrict__ seg_row,
                                          const int* __restrict__ pixpos, int* __restrict__ pick_ground, int* __restrict__ pick_surf, float* __restrict__ curv_out) {
  // entries the reference leaves stale (k < 5, k >= size - 5 of its never-cleared array) read as 0
  float curv = 0.0f;
  if (k >= 5 && k < m - 5) {
    const float b = rg[k - 5];
    const float dr = (rg[k + 5] - b) / 10;
    const float r1 = rg[k + 4] - b - 9 * dr, r2 = rg[k + 3] - b - 8 * dr, r3 = rg[k + 2] - b - 7 * dr, r4 = rg[k + 1] - b - 6 * dr, r5 = rg[k] - b - 5 * dr;
    const float r6 = rg[k - 1] - b - 4 * dr, r7 = rg[k - 2] - b - 3 * dr, r8 = rg[k - 3] - b - 2 * dr, r9 = rg[k - 4] - b - 1 * dr;
    const float cov = ((((((((r1 * r1 + r2 * r2) + r3 * r3) + r4 * r4) + r5 * r5) + r6 * r6) + r7 * r7) + r8 * r8) + r9 * r9) / 9;
    curv = cov * 10 / rg[k];
  }
  if (curv_out) curv_out[k] = curv;
  const int row = seg_row[k];
  const int start = pixpos[(size_t)row * Cn] - 1 + 5, end = pixpos[(size_t)(row + 1) * Cn] - 1 - 5;   // start/end_ring_index
  bool in = false;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int sp = (start * (6 - j) + end * j) / 6, ep = (start * (5 - j) + end * (j + 1)) / 6 - 1;
    in |= (sp < ep && k >= sp && k <= ep);
  }
  const bool g = seg_ground[k] != 0;
  pick_ground[k] = (in && g) ? 1 : 0;
  pick_surf[k] = (in && !g && curv < 1.0f) ? 1 : 0;
}

}  // namespace lvf

using namespace lvf;

static std::atomic<int> g_extract_host_counts{[] { const char* e = std::getenv("LVF_EXTRACT_HOST_COUNTS"); return (e && e[0] == '1') ? 1 : 0; }()};

// The host-counted path (rounds 2-4; kept as the fallback of the device-counted one and as its reference in the tests): every stage reads its
// output count back before the next one is sized — 13 stream waits per scan.
static int extract_host_counted(lvf_ctx* ctx, const float* points, int n, int stride_floats, const lvf_lidar_params* prm, const double* extrinsic7, const ExP& P,
                                lvf_cloud** ground_out, lvf_cloud** surf_out, lvf_lidar_extract_debug* dbg) {
  hipStream_t s = ctx->stream;
  const int npix = P.R * P.Cn;
  // ---- Preprocess
  lvf_cloud* filtered = nullptr;
  {
    DevBuf<float> src; DevBuf<float4> packed; DevBuf<int> flags;
    HostPin<float> stage;                // (pooled pinned staging)
    StreamWaitGuard stage_guard(s);      // every path out of this scope — the error returns too — waits for the copy before the block returns to the pool
    LVF_TRY(src.upload_staged(points, (size_t)n * stride_floats, s, stage)); LVF_TRY(packed.alloc(std::max(n, 1))); LVF_TRY(flags.alloc(std::max(n, 1)));
    if (n) hipLaunchKernelGGL(k_ex_preflag, dim3(gride(n)), dim3(kE), 0, s, n, src.p, stride_floats, P.min2, P.max2, packed.p, flags.p);
    LVF_HIP(hipGetLastError());
    LVF_TRY(compact_points(ctx, packed.p, n, flags.p, &filtered));
  }
  struct Guard { lvf_cloud* c[8]; int k = 0; ~Guard() { for (int i = 0; i < k; ++i) if (c[i]) lvf_cloud_destroy(c[i]); } void add(lvf_cloud* x) { c[k++] = x; } } guard;
  guard.add(filtered);
  const int m = filtered->n;
  // ---- projection, ground, segmentation
  DevBuf<int> pixel_src, parent, size, flags, pos, label; DevBuf<unsigned long long> rows; DevBuf<float4> full; DevBuf<float> range; DevBuf<signed char> ground;
  LVF_TRY(pixel_src.alloc(npix)); LVF_TRY(parent.alloc(npix)); LVF_TRY(size.alloc(npix)); LVF_TRY(flags.alloc(npix)); LVF_TRY(pos.alloc((size_t)npix + 1));
  LVF_TRY(label.alloc(npix)); LVF_TRY(rows.alloc(npix)); LVF_TRY(full.alloc(npix)); LVF_TRY(range.alloc(npix)); LVF_TRY(ground.alloc(npix));
  LVF_HIP(hipMemsetAsync(pixel_src.p, 0xff, (size_t)4 * npix, s));      // -1
  LVF_HIP(hipMemsetAsync(size.p, 0, (size_t)4 * npix, s));
  LVF_HIP(hipMemsetAsync(rows.p, 0, (size_t)8 * npix, s));
  if (m) hipLaunchKernelGGL(k_ex_project, dim3(gride(m)), dim3(kE), 0, s, m, (const int*)nullptr, filtered->pts.p, P, pixel_src.p);
  hipLaunchKernelGGL(k_ex_fill, dim3(gride(npix)), dim3(kE), 0, s, npix, filtered->pts.p, P, pixel_src.p, full.p, range.p);
  hipLaunchKernelGGL(k_ex_ground, dim3(gride(npix)), dim3(kE), 0, s, npix, full.p, range.p, P, ground.p, parent.p);
  hipLaunchKernelGGL(k_ex_union, dim3(gride(npix)), dim3(kE), 0, s, npix, range.p, P, parent.p);
  hipLaunchKernelGGL(k_ex_stats, dim3(gride(npix)), dim3(kE), 0, s, npix, P.Cn, parent.p, size.p, rows.p);
  hipLaunchKernelGGL(k_ex_segflag, dim3(gride(npix)), dim3(kE), 0, s, npix, parent.p, size.p, rows.p, ground.p, flags.p, label.p);
  LVF_HIP(hipGetLastError());
  LVF_TRY(device_exclusive_scan_i32(ctx, flags.p, npix, pos.p));
  int num = 0;
  float4 ends[2] = {make_float4(1, 0, 0, 0), make_float4(1, 0, 0, 0)};
  {
    // three small read-backs, one wait, through the context's pinned mailbox (lvf::read_back's reason)
    LVF_TRY(ctx->mailbox.reserve(4096));
    char* mb = ctx->mailbox.p;
    LVF_HIP(hipMemcpyAsync(mb, pos.p + npix, sizeof(int), hipMemcpyDeviceToHost, s));
    if (m) {
      LVF_HIP(hipMemcpyAsync(mb + 16, filtered->pts.p, sizeof(float4), hipMemcpyDeviceToHost, s));
      LVF_HIP(hipMemcpyAsync(mb + 32, filtered->pts.p + (m - 1), sizeof(float4), hipMemcpyDeviceToHost, s));
    }
    LVF_HIP(hipStreamSynchronize(s));
    std::memcpy(&num, mb, sizeof(int));
    if (m) { std::memcpy(&ends[0], mb + 16, sizeof(float4)); std::memcpy(&ends[1], mb + 32, sizeof(float4)); }
  }
  const OriP o = find_start_end(ends[0], ends[1], prm->cycle_time);      // FindStartEndAngle (projection.cpp:42-56)
  DevBuf<float4> seg; DevBuf<float> seg_range, curv; DevBuf<int> seg_ground, seg_row, latch, latch_pos, pick_g, pick_s;
  const int mm = std::max(num, 1);
  LVF_TRY(seg.alloc(mm)); LVF_TRY(seg_range.alloc(mm)); LVF_TRY(curv.alloc(mm)); LVF_TRY(seg_ground.alloc(mm)); LVF_TRY(seg_row.alloc(mm));
  LVF_TRY(latch.alloc(mm)); LVF_TRY(latch_pos.alloc((size_t)mm + 1)); LVF_TRY(pick_g.alloc(mm)); LVF_TRY(pick_s.alloc(mm));
  lvf_cloud *g_raw = nullptr, *s_raw = nullptr;
  if (num) {
    hipLaunchKernelGGL(k_ex_segemit, dim3(gride(npix)), dim3(kE), 0, s, npix, P.Cn, flags.p, pos.p, full.p, range.p, ground.p, seg.p, seg_range.p, seg_ground.p, seg_row.p);
    hipLaunchKernelGGL(k_ex_latch, dim3(gride(num)), dim3(kE), 0, s, num, (const int*)nullptr, seg.p, o, OriSrc{nullptr, nullptr, 0.0}, latch.p);
    LVF_HIP(hipGetLastError());
    LVF_TRY(device_exclusive_scan_i32(ctx, latch.p, num, latch_pos.p));
    hipLaunchKernelGGL(k_ex_pick, dim3(gride(num)), dim3(kE), 0, s, num, (const int*)nullptr, P.Cn, seg.p, seg_range.p, seg_ground.p, seg_row.p, pos.p, pick_g.p, pick_s.p, curv.p);
    hipLaunchKernelGGL(k_ex_reltime, dim3(gride(num)), dim3(kE), 0, s, num, (const int*)nullptr, seg.p, o, OriSrc{nullptr, nullptr, 0.0}, latch_pos.p);
    LVF_HIP(hipGetLastError());
  }
  LVF_TRY(compact_points(ctx, seg.p, num, pick_g.p, &g_raw)); guard.add(g_raw);
  LVF_TRY(compact_points(ctx, seg.p, num, pick_s.p, &s_raw)); guard.add(s_raw);
  if (dbg) {
    dbg->n_filtered = m; dbg->n_segmented = num; dbg->n_ground_raw = g_raw->n; dbg->n_surf_raw = s_raw->n;
    if (dbg->label_mat) LVF_HIP(hipMemcpyAsync(dbg->label_mat, label.p, (size_t)4 * npix, hipMemcpyDeviceToHost, s));
    if (dbg->ground_mat) LVF_HIP(hipMemcpyAsync(dbg->ground_mat, ground.p, (size_t)npix, hipMemcpyDeviceToHost, s));
    if (dbg->range_mat) LVF_HIP(hipMemcpyAsync(dbg->range_mat, range.p, (size_t)4 * npix, hipMemcpyDeviceToHost, s));
    if (dbg->ground_raw && g_raw->n) LVF_HIP(hipMemcpyAsync(dbg->ground_raw, g_raw->pts.p, (size_t)16 * g_raw->n, hipMemcpyDeviceToHost, s));
    if (dbg->surf_raw && s_raw->n) LVF_HIP(hipMemcpyAsync(dbg->surf_raw, s_raw->pts.p, (size_t)16 * s_raw->n, hipMemcpyDeviceToHost, s));
    LVF_HIP(hipStreamSynchronize(s));
  }
  // ---- the PCL tail (association.cpp:210-234) on device clouds
  lvf_cloud *sv = nullptr, *sr = nullptr, *gv = nullptr, *gp = nullptr;
  LVF_TRY(lvf_cloud_voxel_filter(s_raw, 2 * prm->resolution, &sv)); guard.add(sv);
  LVF_TRY(lvf_cloud_radius_outlier_filter(sv, 4 * prm->resolution, 4, &sr)); guard.add(sr);
  LVF_TRY(lvf_cloud_voxel_filter(g_raw, 2 * prm->resolution, &gv)); guard.add(gv);
  LVF_TRY(lvf_cloud_segment_plane(gv, 0.1f * prm->resolution, 100, prm->ransac_seed, &gp, nullptr, nullptr)); guard.add(gp);
  LVF_TRY(lvf_cloud_transform(gp, extrinsic7, ground_out));
  const int rc = lvf_cloud_transform(sr, extrinsic7, surf_out);
  if (rc != LVF_OK) { lvf_cloud_destroy(*ground_out); *ground_out = nullptr; return rc; }
  LVF_HIP(hipStreamSynchronize(s));
  return LVF_OK;
}

// The device-counted path: ONE stream wait per scan.  Every intermediate cloud is a buffer of the raw scan's (or the range image's) capacity
// with its count in device memory; launches are sized by the capacity and read the count; scans / compactions are one launch each
// (device_scan1); the start / end orientations are taken by a one-thread launch with the host's arithmetic; the PCL tail is dc_pcl_tail.
// The counts (and the tail's verdict) come back in one read before the two extrinsic transforms, which are then sized exactly.
// *done = false: a scan outside what the path was sized for (the tail's a-priori bounds) — the caller takes the host-counted path.
static int extract_device_counted(lvf_ctx* ctx, const float* points, int n, int stride_floats, const lvf_lidar_params* prm, const double* extrinsic7, const ExP& P,
                                  lvf_cloud** ground_out, lvf_cloud** surf_out, lvf_lidar_extract_debug* dbg, bool* done) {
  hipStream_t s = ctx->stream;
  *done = false;
  static const bool timing = std::getenv("LVF_EXTRACT_TIMING") != nullptr;
  const auto t_in = std::chrono::steady_clock::now();
  auto us_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
  const int npix = P.R * P.Cn;
  const int capseg = std::min(npix, n);               // a segmented point is a pixel AND a filtered point
  DevBuf<ExDev> exd;
  LVF_TRY(exd.alloc(1));
  ExDev* ex = exd.p;
  // ---- what does not depend on the scan goes first: it runs while the host is still preparing the upload
  DevBuf<int> pixel_src, size; DevBuf<unsigned long long> rows;
  LVF_TRY(pixel_src.alloc(npix)); LVF_TRY(size.alloc(npix)); LVF_TRY(rows.alloc(npix));
  hipLaunchKernelGGL(k_ex_clear, dim3(gride(npix)), dim3(kE), 0, s, npix, pixel_src.p, size.p, rows.p);
  DevBuf<unsigned char> tail_state;
  std::shared_ptr<void> tail_keep;      // the tail's scratch: alive until this function has waited for the streams (declared before the guard below)
  struct SideGuard { lvf_ctx* c; ~SideGuard() { if (c->stream2) (void)hipStreamSynchronize(c->stream2); } } side_guard{ctx};      // an error return must not release scratch the side stream still uses
  DcTailPlan plan;
  LVF_TRY(dc_tail_begin(ctx, capseg, prm->resolution, prm->max_range, tail_state, tail_keep, &plan));
  if (!plan.supported) return LVF_OK;
  // ---- Preprocess
  DevBuf<float> src; DevBuf<float4> packed, filtered; DevBuf<int> pflags;
  HostPin<float> stage;                // (pooled pinned staging)
  StreamWaitGuard stage_guard(s);      // every path out — the error returns too — waits for the copy before the block returns to the pool
  LVF_TRY(src.upload_staged(points, (size_t)n * stride_floats, s, stage)); LVF_TRY(packed.alloc(n)); LVF_TRY(pflags.alloc(n)); LVF_TRY(filtered.alloc(n));
  const double us_upload = us_since(t_in);
  hipLaunchKernelGGL(k_ex_preflag, dim3(gride(n)), dim3(kE), 0, s, n, src.p, stride_floats, P.min2, P.max2, packed.p, pflags.p);
  LVF_HIP(hipGetLastError());
  LVF_TRY(device_scan1(ctx, pflags.p, n, nullptr, nullptr, &ex->cnt[0], packed.p, filtered.p));
  const int* m_dev = &ex->cnt[0];
  // ---- projection, ground, segmentation
  DevBuf<int> parent, flags, pos, label; DevBuf<float4> full; DevBuf<float> range; DevBuf<signed char> ground;
  LVF_TRY(parent.alloc(npix)); LVF_TRY(flags.alloc(npix)); LVF_TRY(pos.alloc((size_t)npix + 1));
  LVF_TRY(label.alloc(npix)); LVF_TRY(full.alloc(npix)); LVF_TRY(range.alloc(npix)); LVF_TRY(ground.alloc(npix));
  hipLaunchKernelGGL(k_ex_project, dim3(gride(n)), dim3(kE), 0, s, n, m_dev, filtered.p, P, pixel_src.p);
  hipLaunchKernelGGL(k_ex_fill, dim3(gride(npix)), dim3(kE), 0, s, npix, filtered.p, P, pixel_src.p, full.p, range.p);
  hipLaunchKernelGGL(k_ex_ground, dim3(gride(npix)), dim3(kE), 0, s, npix, full.p, range.p, P, ground.p, parent.p);
  hipLaunchKernelGGL(k_ex_union, dim3(gride(npix)), dim3(kE), 0, s, npix, range.p, P, parent.p);
  hipLaunchKernelGGL(k_ex_stats, dim3(gride(npix)), dim3(kE), 0, s, npix, P.Cn, parent.p, size.p, rows.p);
  hipLaunchKernelGGL(k_ex_segflag, dim3(gride(npix)), dim3(kE), 0, s, npix, parent.p, size.p, rows.p, ground.p, flags.p, label.p);
  LVF_HIP(hipGetLastError());
  LVF_TRY(device_scan1(ctx, flags.p, npix, nullptr, pos.p, &ex->cnt[1], nullptr, nullptr));
  const int* num_dev = &ex->cnt[1];
  // ---- AdjustDistortion, CalculateSmoothness, ExtractFeatures' picks
  DevBuf<float4> seg, g_raw, s_raw; DevBuf<float> seg_range, curv; DevBuf<int> seg_ground, seg_row, latch, latch_pos, pick_g, pick_s;
  const int mm = std::max(capseg, 1);
  LVF_TRY(seg.alloc(mm)); LVF_TRY(seg_range.alloc(mm)); LVF_TRY(curv.alloc(mm)); LVF_TRY(seg_ground.alloc(mm)); LVF_TRY(seg_row.alloc(mm));
  LVF_TRY(latch.alloc(mm)); LVF_TRY(latch_pos.alloc((size_t)mm + 1)); LVF_TRY(pick_g.alloc(mm)); LVF_TRY(pick_s.alloc(mm));
  LVF_TRY(g_raw.alloc(mm)); LVF_TRY(s_raw.alloc(mm));
  const OriSrc op{filtered.p, m_dev, prm->cycle_time};      // (start / end orientations: every thread of the two launches below takes them itself)
  hipLaunchKernelGGL(k_ex_segemit_latch, dim3(gride(npix)), dim3(kE), 0, s, npix, P.Cn, flags.p, pos.p, full.p, range.p, ground.p, seg.p, seg_range.p, seg_ground.p, seg_row.p, op,
                     latch.p);
  LVF_HIP(hipGetLastError());
  LVF_TRY(device_scan1(ctx, latch.p, capseg, num_dev, latch_pos.p, nullptr, nullptr, nullptr));
  hipLaunchKernelGGL(k_ex_pick_reltime, dim3(gride(capseg)), dim3(kE), 0, s, capseg, num_dev, P.Cn, seg.p, seg_range.p, seg_ground.p, seg_row.p, pos.p, pick_g.p, pick_s.p, curv.p, op,
                     latch_pos.p);
  LVF_HIP(hipGetLastError());
  // ---- the picks' compactions and the PCL tail (association.cpp:210-234): the ground half on the side stream, the surf half on this one
  hipStream_t q = s;
  DevBuf<float4> surf_pre, ground_pre;
  SideGuard side_fork_guard{ctx};      // declared AFTER every buffer the side stream touches (seg, pick_g, g_raw, ground_pre ...): on an error return it waits for stream2 before they go back to the pool
  LVF_TRY(dc_tail_fork(ctx, plan, &q));
  LVF_TRY(device_scan1_on(ctx, q, plan.lane_ground, pick_g.p, capseg, num_dev, nullptr, &ex->cnt[2], seg.p, g_raw.p));
  LVF_TRY(device_scan1_on(ctx, s, 0, pick_s.p, capseg, num_dev, nullptr, &ex->cnt[3], seg.p, s_raw.p));
  LVF_TRY(dc_tail_run(ctx, plan, q, s_raw.p, &ex->cnt[3], g_raw.p, &ex->cnt[2], (unsigned long long)prm->ransac_seed, tail_state, tail_keep, surf_pre, ground_pre));
  // ---- the one read-back: the four counts here, the tail's four and its verdict
  int hc[4] = {0, 0, 0, 0}, ht[5] = {0, 0, 0, 0, 0};
  {
    LVF_TRY(ctx->mailbox.reserve(4096));
    char* mb = ctx->mailbox.p;
    LVF_HIP(hipMemcpyAsync(mb, ex->cnt, sizeof(hc), hipMemcpyDeviceToHost, s));
    LVF_HIP(hipMemcpyAsync(mb + 64, tail_state.p + dc_state_counts_offset(), sizeof(ht), hipMemcpyDeviceToHost, s));
    const double us_enq = us_since(t_in);
    LVF_HIP(hipStreamSynchronize(s));
    stage_guard.dismiss();             // the upload has been waited for: no second full stream wait when this scope ends (the transforms below are consumed on the same stream)
    if (timing) std::fprintf(stderr, "extract: upload enqueued at %.1f us, all launches enqueued at %.1f us, stream drained at %.1f us\n", us_upload, us_enq, us_since(t_in));
    std::memcpy(hc, mb, sizeof(hc)); std::memcpy(ht, mb + 64, sizeof(ht));
  }
  if (ht[4] != 0) return LVF_OK;                        // outside the tail's a-priori bounds: the host-counted path decides (and reports)
  if (dbg) {
    dbg->n_filtered = hc[0]; dbg->n_segmented = hc[1]; dbg->n_ground_raw = hc[2]; dbg->n_surf_raw = hc[3];
    if (dbg->label_mat) LVF_HIP(hipMemcpyAsync(dbg->label_mat, label.p, (size_t)4 * npix, hipMemcpyDeviceToHost, s));
    if (dbg->ground_mat) LVF_HIP(hipMemcpyAsync(dbg->ground_mat, ground.p, (size_t)npix, hipMemcpyDeviceToHost, s));
    if (dbg->range_mat) LVF_HIP(hipMemcpyAsync(dbg->range_mat, range.p, (size_t)4 * npix, hipMemcpyDeviceToHost, s));
    if (dbg->ground_raw && hc[2]) LVF_HIP(hipMemcpyAsync(dbg->ground_raw, g_raw.p, (size_t)16 * hc[2], hipMemcpyDeviceToHost, s));
    if (dbg->surf_raw && hc[3]) LVF_HIP(hipMemcpyAsync(dbg->surf_raw, s_raw.p, (size_t)16 * hc[3], hipMemcpyDeviceToHost, s));
    LVF_HIP(hipStreamSynchronize(s));
  }
  // ---- Sensor2Robot (association.cpp:236-247), sized exactly; consumed on the same stream, nothing to wait for
  LVF_TRY(transform_points(ctx, ground_pre.p, ht[3], extrinsic7, ground_out));
  const int rc = transform_points(ctx, surf_pre.p, ht[1], extrinsic7, surf_out);
  if (rc != LVF_OK) { lvf_cloud_destroy(*ground_out); *ground_out = nullptr; return rc; }
  *done = true;
  return LVF_OK;
}

extern "C" {

void lvf_lidar_params_default(lvf_lidar_params* p) {
  if (!p) return;
  p->num_scans = 64; p->horizon_scan = 1800; p->ang_res_y = 0.427f; p->ang_bottom = 24.9f; p->ground_rows = 60;   // config/kitti.yaml:35-39
  p->cycle_time = 0.1036; p->min_range = 5.0f; p->max_range = 30.0f; p->resolution = 0.2f;                        // :40-45
  p->ransac_seed = 12345;
}

int lvf_lidar_extract(lvf_ctx* ctx, const float* points, int n, int stride_floats, const lvf_lidar_params* prm, const double* extrinsic7,
                      lvf_cloud** ground_out, lvf_cloud** surf_out, lvf_lidar_extract_debug* dbg) {
  LVF_REQUIRE(ctx && prm && extrinsic7 && ground_out && surf_out, "lvf_lidar_extract: null argument");
  LVF_REQUIRE(n >= 0 && (n == 0 || points) && stride_floats >= 3, "lvf_lidar_extract: bad scan (n=%d stride=%d)", n, stride_floats);
  LVF_REQUIRE(prm->num_scans > 1 && prm->num_scans <= 64 && prm->horizon_scan > 0 && prm->ground_rows >= 0 && prm->ground_rows < prm->num_scans,
              "lvf_lidar_extract: unsupported geometry (num_scans <= 64, 0 <= ground_rows < num_scans)");
  LVF_TRY(lvf::enter(ctx));
  ExP P;
  P.R = prm->num_scans; P.Cn = prm->horizon_scan; P.ground_rows = prm->ground_rows;
  P.ang_res_x = (float)(360.0 / (float)prm->horizon_scan); P.ang_res_y = prm->ang_res_y; P.ang_bottom = prm->ang_bottom;
  const float alpha_x = (float)((double)P.ang_res_x / 180.0 * M_PI), alpha_y = (float)((double)P.ang_res_y / 180.0 * M_PI);
  P.sin_ax = std::sin(alpha_x); P.cos_ax = std::cos(alpha_x); P.sin_ay = std::sin(alpha_y); P.cos_ay = std::cos(alpha_y);
  P.theta = (float)(60.0 / 180.0 * M_PI);
  P.min2 = prm->min_range * prm->min_range; P.max2 = prm->max_range * prm->max_range;
  if (dbg) dbg->n_filtered = dbg->n_segmented = dbg->n_ground_raw = dbg->n_surf_raw = 0;
  *ground_out = nullptr; *surf_out = nullptr;
  if (n > 0 && !g_extract_host_counts.load()) {
    bool done = false;
    LVF_TRY(extract_device_counted(ctx, points, n, stride_floats, prm, extrinsic7, P, ground_out, surf_out, dbg, &done));
    if (done) return LVF_OK;
  }
  return extract_host_counted(ctx, points, n, stride_floats, prm, extrinsic7, P, ground_out, surf_out, dbg);
}

// Test / A-B hook: 1 = lvf_lidar_extract takes the host-counted path (process-wide).  Not part of the reference surface.
int lvf_debug_extract_host_counts(int on) { const int was = g_extract_host_counts.exchange(on ? 1 : 0); return was; }

}  // extern "C"
