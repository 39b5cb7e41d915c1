// cloud_kernels.hip — map-cloud maintenance on device: the steps immediately before the kNN association.
//
// Replaces, with the clouds resident in HBM between scan-to-map solves,
//   Mapping::MergeScan / ToWorld, FeatureAssociation::Sensor2Robot   float SE3 transform of a whole cloud
//        src/lvio_fusion/src/mapping.cpp:193-220, src/association.cpp:236-247                     -> lvf_cloud_transform
//   Mapping::BuildMapFrame / BuildOldMapFrame  `points_merged += pointclouds[t]`  mapping.cpp:78-137 -> lvf_cloud_concat
//   pcl::VoxelGrid<PointI> (leaf 2 x resolution)              association.cpp:210-215,222-224       -> lvf_cloud_voxel_filter
//   pcl::RadiusOutlierRemoval<PointI> (r = 4 x resolution, min 4)  association.cpp:217-221          -> lvf_cloud_radius_outlier_filter
//   FeatureAssociation::SegmentGround: pcl::SACSegmentation(SACMODEL_PLANE, SAC_RANSAC, 100 its, thr 0.1 x resolution,
//        optimize coefficients) + ExtractIndices               association.cpp:249-268               -> lvf_cloud_segment_plane
// All HBM-streaming or small-grid work: one thread per point, counting sorts, prefix-sum compaction.
//
// DECLARED upstream semantics (PCL is un-vendored; restated in oracle/cloud.h):
//   VoxelGrid: inverse_leaf = 1/leaf (float); min_b = floor(min * inverse_leaf), div_b = max_b - min_b + 1; a point's voxel
//     is ijk = floor(p * inverse_leaf) - min_b, idx = i + j div_x + k div_x div_y; output = per-voxel centroid of ALL fields
//     (x, y, z, intensity), voxels in ascending idx.  PCL accumulates the centroid in FLOAT over the voxel's points in the order its
//     (unstable) std::sort leaves them; the declared order is ascending input index.  The device does exactly that: a stable sort of
//     (voxel, point index) pairs, then one thread per voxel adds its points in that order in float and divides by the count — the
//     result does not depend on scheduling and equals the sequential restatement bit for bit (round 2 used double atomics: 2e-6 off,
//     enough to flip later radius-outlier / plane decisions).
//   RadiusOutlierRemoval: keep a point iff (number of points with squared distance < r^2, itself included) > min_neighbors;
//     input order preserved.
//   SACSegmentation/RANSAC: hypothesis = plane through 3 distinct sampled points; inliers |n.p + d| < thr; best = most
//     inliers (first wins ties); PCL's adaptive stop k = log(1 - 0.99) / log(1 - w^3) is applied in hypothesis order; then
//     the plane is re-fitted to the inliers (centroid + smallest eigenvector of the covariance) and the inliers re-selected.
//     PCL draws samples from boost::mt19937(12345) through its own shuffling; that stream cannot be reproduced without PCL,
//     so sampling uses a documented counter-based generator (splitmix64 of (seed, hypothesis, draw)).
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <memory>
#include <vector>
#include "lvf_internal.hpp"

#pragma clang fp contract(off)   // the float SE3 transform must round after every multiply and add (bit-exact with the oracle)

namespace lvf {

constexpr int kC = 256;
static inline int gridc(int n) { return (std::max(n, 1) + kC - 1) / kC; }

// ---------------------------------------------------------------------------------------------- pack / transform / concat
__global__ __launch_bounds__(kC) void k_cloud_pack(int n, const float* __restrict__ src, int stride, int ioff, float4* __restrict__ dst) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (i >= n) return;
  const float* s = src + (size_t)i * stride;
  dst[i] = make_float4(s[0], s[1], s[2], ioff >= 0 ? s[ioff] : 0.0f);
}

// FeatureAssociation::AlignScan's slice: elements [first, first + count) of the virtual concatenation pc1 + pc2, copied as PointXYZI with
// intensity 0 (pcl::copyPointCloud from PointXYZ leaves the extra field at its default)
__global__ __launch_bounds__(kC) void k_align_slice(int first, int count, const float4* __restrict__ a, int na, const float4* __restrict__ b, float4* __restrict__ out) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (i >= count) return;
  const int j = first + i;
  const float4 p = j < na ? a[j] : b[j - na];
  out[i] = make_float4(p.x, p.y, p.z, 0.0f);
}

struct Tf32c { float a[9]; float t[3]; };
// float instantiation of ceres::QuaternionRotatePoint (1.x): normalise, then the expanded polynomial — same sequence of
// rounded operations as knn_kernels.hip::make_tf32 and oracle/knn.h::transform_query_f32
__device__ __forceinline__ Tf32c make_tf(const float tf[7]) {
  const float q0 = tf[3], q1 = tf[0], q2 = tf[1], q3 = tf[2];
  const float ss = ((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3;
  const float scale = 1.0f / __builtin_sqrtf(ss);
  const float u0 = scale * q0, u1 = scale * q1, u2 = scale * q2, u3 = scale * q3;
  const float t2 = u0 * u1, t3 = u0 * u2, t4 = u0 * u3, t5 = -u1 * u1, t6 = u1 * u2, t7 = u1 * u3, t8 = -u2 * u2, t9 = u2 * u3, t1 = -u3 * u3;
  Tf32c T;
  T.a[0] = t8 + t1; T.a[1] = t6 - t4; T.a[2] = t3 + t7;
  T.a[3] = t4 + t6; T.a[4] = t5 + t1; T.a[5] = t9 - t2;
  T.a[6] = t7 - t3; T.a[7] = t2 + t9; T.a[8] = t5 + t8;
  T.t[0] = tf[4]; T.t[1] = tf[5]; T.t[2] = tf[6];
  return T;
}
struct TfArgC { float v[7]; };
__global__ __launch_bounds__(kC) void k_cloud_transform(int n, const float4* __restrict__ in, TfArgC tfa, float4* __restrict__ out) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (i >= n) return;
  const Tf32c T = make_tf(tfa.v);
  const float4 p = in[i];
  float4 o;
  o.x = (2.0f * ((T.a[0] * p.x + T.a[1] * p.y) + T.a[2] * p.z) + p.x) + T.t[0];
  o.y = (2.0f * ((T.a[3] * p.x + T.a[4] * p.y) + T.a[5] * p.z) + p.y) + T.t[1];
  o.z = (2.0f * ((T.a[6] * p.x + T.a[7] * p.y) + T.a[8] * p.z) + p.z) + T.t[2];
  o.w = p.w;                                   // intensity rides along (mapping.cpp:201)
  out[i] = o;
}

// ---------------------------------------------------------------------------------------------- bounds
__device__ __forceinline__ unsigned f2ord_c(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float ord2f_c(unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &u, 4); return f; }
// capped grid + workgroup-level reduction: 6 atomics per WORKGROUP (one per wave cost 50-110 us on 100 k points, see cell_runs)
constexpr int kCapBlocks = 256;
__global__ __launch_bounds__(kC) void k_cloud_bounds(int n, const float4* __restrict__ pts, unsigned* __restrict__ bounds) {
  float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
  for (int i = blockIdx.x * kC + threadIdx.x; i < n; i += gridDim.x * kC) {
    const float4 p = pts[i];
    x = fminf(x, p.x); y = fminf(y, p.y); z = fminf(z, p.z); X = fmaxf(X, p.x); Y = fmaxf(Y, p.y); Z = fmaxf(Z, p.z);
  }
  for (int o = 32; o > 0; o >>= 1) {
    x = fminf(x, __shfl_down(x, o)); y = fminf(y, __shfl_down(y, o)); z = fminf(z, __shfl_down(z, o));
    X = fmaxf(X, __shfl_down(X, o)); Y = fmaxf(Y, __shfl_down(Y, o)); Z = fmaxf(Z, __shfl_down(Z, o));
  }
  __shared__ float red[kC / 64][6];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = x; red[w][1] = y; red[w][2] = z; red[w][3] = X; red[w][4] = Y; red[w][5] = Z; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int k = 1; k < kC / 64; ++k) v = threadIdx.x < 3 ? fminf(v, red[k][threadIdx.x]) : fmaxf(v, red[k][threadIdx.x]);
    if (threadIdx.x < 3) atomicMin(bounds + threadIdx.x, f2ord_c(v)); else atomicMax(bounds + threadIdx.x, f2ord_c(v));
  }
}
static int cloud_bounds(const lvf_cloud* c, float lo[3], float hi[3]) {
  DevBuf<unsigned> b;
  LVF_TRY(b.alloc(6));
  hipStream_t s = c->ctx->stream;
  // {+max, +max, +max, 0, 0, 0} as ordered-uint bounds: two memsets instead of a copy from a stack array (a pageable copy)
  LVF_HIP(hipMemsetAsync(b.p, 0xff, 3 * sizeof(unsigned), s));
  LVF_HIP(hipMemsetAsync(b.p + 3, 0, 3 * sizeof(unsigned), s));
  hipLaunchKernelGGL(k_cloud_bounds, dim3(std::min(kCapBlocks, gridc(c->n))), dim3(kC), 0, s, c->n, c->pts.p, b.p);
  unsigned h[6];
  LVF_TRY(read_back(c->ctx, h, b.p, sizeof(h)));
  for (int k = 0; k < 3; ++k) { lo[k] = ord2f_c(h[k]); hi[k] = ord2f_c(h[3 + k]); }
  for (int k = 0; k < 3; ++k)
    if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) { set_error("cloud has non-finite coordinates"); return LVF_ERR_INVALID; }
  return LVF_OK;
}

// ---------------------------------------------------------------------------------------------- voxel grid
struct VoxP { float inv_leaf; int minb[3]; int div[3]; };
__device__ __forceinline__ int voxel_of(const float4 p, const VoxP v) {
  const int i = (int)(floorf(p.x * v.inv_leaf) - (float)v.minb[0]);
  const int j = (int)(floorf(p.y * v.inv_leaf) - (float)v.minb[1]);
  const int k = (int)(floorf(p.z * v.inv_leaf) - (float)v.minb[2]);
  return i + j * v.div[0] + k * v.div[0] * v.div[1];
}
// (n_dev / vp, here and below: the count / the parameters as an earlier launch left them on the device — the device-counted pipeline of
// lvf_lidar_extract — instead of the by-value arguments)
__global__ __launch_bounds__(kC) void k_voxel_key(int n, const int* __restrict__ n_dev, const float4* __restrict__ pts, VoxP v, const VoxP* __restrict__ vp,
                                                  unsigned* __restrict__ key, int* __restrict__ val) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  if (i >= n) return;
  if (vp) v = *vp;
  key[i] = (unsigned)voxel_of(pts[i], v); val[i] = i;
}
// a sorted position is the HEAD of its voxel's run when its key differs from its predecessor's
__global__ __launch_bounds__(kC) void k_voxel_heads(int n, const int* __restrict__ n_dev, const unsigned* __restrict__ key_sorted, int* __restrict__ flags) {
  const int j = blockIdx.x * kC + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  if (j < n) flags[j] = (j == 0 || key_sorted[j] != key_sorted[j - 1]) ? 1 : 0;
}
// One thread per run head: the centroid of ALL four fields, accumulated in float over the voxel's points in ascending input index (`order` =
// point indices sorted stably by voxel, so a voxel is a run of equal keys), then divided by the count; pos = exclusive scan of the head
// flags = the voxel's output slot (runs come in ascending voxel index).  Nothing here is sized by the voxel GRID (562 k cells for 3.4 k
// occupied ones at configs[2]): round 3 counted per cell with atomics and scanned two grid-sized arrays.
// The workgroup first stages its 256 sorted positions (key, point) in LDS — one coalesced pass, every gather in flight at once — and a run's
// head then walks the run there; only the part of a run that leaves the workgroup's tile is gathered from memory.  (One thread walking its run
// through dependent gathers made the launch as long as the LONGEST run: 18-28 us for the 100-point voxels next to the sensor.)
__global__ __launch_bounds__(kC) void k_voxel_emit(int n, const int* __restrict__ n_dev, const float4* __restrict__ pts, const unsigned* __restrict__ key_sorted,
                                                   const int* __restrict__ order, const int* __restrict__ flags, const int* __restrict__ pos, float4* __restrict__ out) {
  __shared__ unsigned s_key[kC];
  __shared__ float4 s_pt[kC];
  const int base = blockIdx.x * kC;
  const int j0 = base + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  if (j0 < n) { s_key[threadIdx.x] = key_sorted[j0]; s_pt[threadIdx.x] = pts[order[j0]]; }
  __syncthreads();
  if (j0 >= n || !flags[j0]) return;
  const unsigned k0 = s_key[threadIdx.x];
  // the ADDS are sequential by definition (float accumulation in input order); what feeds them is not: eight points (and the keys that say
  // whether they still belong to the run) are requested before the first of them is added
  float4 s = s_pt[threadIdx.x];
  int cnt = 1;
  for (int q = j0 + 1; q < n; q += 8) {
    bool in[8]; float4 p[8]; unsigned kk[8];
    // (keys and points are read UNCONDITIONALLY and compared afterwards: written as `q + u < n && key == k0` the reads ended up behind eight
    // branches, each waiting for its own — 290 clocks per point)
    if (q + 7 - base < kC) {                         // the whole batch lies in the staged tile
      int t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = min(q + u, n - 1) - base;
#pragma unroll
      for (int u = 0; u < 8; ++u) kk[u] = s_key[t[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = s_pt[t[u]];
    } else {
      int o[8], t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = min(q + u, n - 1);
#pragma unroll
      for (int u = 0; u < 8; ++u) { kk[u] = key_sorted[t[u]]; o[u] = order[t[u]]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = pts[o[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) in[u] = (q + u < n) & (kk[u] == k0);
    bool run = true;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      run = run & in[u];
      // (selects, not branches: x + 0 would not do — it turns -0 into +0)
      s.x = run ? s.x + p[u].x : s.x; s.y = run ? s.y + p[u].y : s.y; s.z = run ? s.z + p[u].z : s.z; s.w = run ? s.w + p[u].w : s.w;
      cnt += run ? 1 : 0;
    }
    if (!run) break;
  }
  const float k = (float)cnt;
  out[pos[j0]] = make_float4(s.x / k, s.y / k, s.z / k, s.w / k);
}

// ---------------------------------------------------------------------------------------------- uniform grid (cell-sorted copy)
struct GridC { float ox, oy, oz, inv_cell; int nx, ny, nz; };
__device__ __forceinline__ int gcoord(float v, float o, float inv_cell, int n) {
  int c = (int)floorf((v - o) * inv_cell);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
__global__ __launch_bounds__(kC) void k_grid_count(int n, const int* __restrict__ n_dev, const float4* __restrict__ pts, GridC g, const GridC* __restrict__ gp,
                                                   int* __restrict__ cell_of, int* __restrict__ counts) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  if (gp) g = *gp;
  int c = -1;
  if (i < n) {
    const float4 p = pts[i];
    c = (gcoord(p.z, g.oz, g.inv_cell, g.nz) * g.ny + gcoord(p.y, g.oy, g.inv_cell, g.ny)) * g.nx + gcoord(p.x, g.ox, g.inv_cell, g.nx);
    cell_of[i] = c;
  }
  int start, len;
  const bool head = cell_runs(c, start, len);
  if (head && c >= 0) atomicAdd(counts + c, len);
}
__global__ __launch_bounds__(kC) void k_grid_scatter(int n, const int* __restrict__ n_dev, const float4* __restrict__ pts, const int* __restrict__ cell_of,
                                                     const int* __restrict__ cell_start, int* __restrict__ cursor, float4* __restrict__ sorted) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  const int c = (i < n) ? cell_of[i] : -1;
  int start, len;
  const bool head = cell_runs(c, start, len);
  int base = 0;
  if (head && c >= 0) base = atomicAdd(cursor + c, len);
  base = __shfl(base, start);
  if (c >= 0) sorted[cell_start[c] + base + ((int)(threadIdx.x & 63) - start)] = pts[i];
}
// neighbours within radius (squared distance < r2, the point itself included); cell size >= radius => 27 cells.
// The count does not depend on the order points sit inside a cell, so the atomic scatter above is harmless.
__global__ __launch_bounds__(kC) void k_radius_count(int n, const int* __restrict__ n_dev, const float4* __restrict__ pts, GridC g, const GridC* __restrict__ gp,
                                                     const int* __restrict__ cell_start, const float4* __restrict__ sorted, float r2, int min_neighbors,
                                                     int* __restrict__ flags) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  if (i >= n) return;
  if (gp) g = *gp;
  const float4 p = pts[i];
  const int cx = gcoord(p.x, g.ox, g.inv_cell, g.nx), cy = gcoord(p.y, g.oy, g.inv_cell, g.ny), cz = gcoord(p.z, g.oz, g.inv_cell, g.nz);
  // only "more than min_neighbors" matters: stop at the first row that settles it (dense ground cells hold thousands of points;
  // scanning all 27 cells to the end was 0.3-0.6 ms of the 0.95 ms filter)
  int cnt = 0;
  // rows nearest first: the point's own row of cells almost always settles the count
  const int order[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
  for (int k = 0; k < 9 && cnt <= min_neighbors; ++k) {
    const int z = cz + order[k][0], y = cy + order[k][1];
    if (z < 0 || z >= g.nz || y < 0 || y >= g.ny) continue;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    const int row = (z * g.ny + y) * g.nx;
    const int lo = cell_start[row + x0], hi = cell_start[row + x1 + 1];   // x-adjacent cells are contiguous
    // eight candidates per round trip, requested unconditionally and counted afterwards (one at a time behind the `cnt` test every candidate was
    // its own dependent load: a point of a sparse region walked its 27 cells at one L2 latency per point — 141 us for the launch on 100 k points)
    for (int j = lo; j < hi && cnt <= min_neighbors; j += 8) {
      float4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = sorted[min(j + u, hi - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float dx = p.x - q[u].x, dyy = p.y - q[u].y, dzz = p.z - q[u].z;
        const float d = (dx * dx + dyy * dyy) + dzz * dzz;
        cnt += ((j + u < hi) & (d < r2)) ? 1 : 0;
      }
    }
  }
  flags[i] = cnt > min_neighbors ? 1 : 0;
}
__global__ __launch_bounds__(kC) void k_compact(int n, const float4* __restrict__ pts, const int* __restrict__ flags, const int* __restrict__ pos,
                                                float4* __restrict__ out) {
  const int i = blockIdx.x * kC + threadIdx.x;
  if (i < n && flags[i]) out[pos[i]] = pts[i];
}

// ---------------------------------------------------------------------------------------------- RANSAC plane
__host__ __device__ inline unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// three distinct indices for hypothesis h (identical on host, device and in the oracle)
__host__ __device__ inline void sample3(unsigned long long seed, int h, int n, int idx[3]) {
  int got = 0;
  for (unsigned draw = 0; got < 3; ++draw) {
    const int v = (int)(splitmix64(seed ^ ((unsigned long long)h << 32) ^ draw) % (unsigned long long)n);
    bool dup = false;
    for (int k = 0; k < got; ++k) dup |= idx[k] == v;
    if (!dup) idx[got++] = v;
  }
}
// SampleConsensusModelPlane::computeModelCoefficients: n = (p1 - p0) x (p2 - p0) normalised, d = -n.p0 (float)
__device__ __forceinline__ bool plane_from3(const float4 a, const float4 b, const float4 c, float co[4]) {
  const float ux = b.x - a.x, uy = b.y - a.y, uz = b.z - a.z, vx = c.x - a.x, vy = c.y - a.y, vz = c.z - a.z;
  float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
  const float nn = (nx * nx + ny * ny) + nz * nz;
  if (!(nn > 0.0f)) return false;              // collinear sample
  const float s = 1.0f / __builtin_sqrtf(nn);
  nx *= s; ny *= s; nz *= s;
  co[0] = nx; co[1] = ny; co[2] = nz; co[3] = -((nx * a.x + ny * a.y) + nz * a.z);
  return true;
}
__global__ __launch_bounds__(kC) void k_ransac_count(int n, const int* __restrict__ n_dev, const float4* __restrict__ pts, unsigned long long seed, float thr,
                                                     int* __restrict__ counts) {
  __shared__ float co[4];
  __shared__ int ok;
  const int h = blockIdx.y;
  if (n_dev) { n = min(n, *n_dev); if (n < 3) return; }            // (SACSegmentation cannot fit a model: no hypothesis counts)
  if (threadIdx.x == 0) {
    int id[3];
    sample3(seed, h, n, id);
    float c4[4] = {0, 0, 0, 0};
    ok = plane_from3(pts[id[0]], pts[id[1]], pts[id[2]], c4) ? 1 : 0;
    co[0] = c4[0]; co[1] = c4[1]; co[2] = c4[2]; co[3] = c4[3];
  }
  __syncthreads();
  if (!ok) return;                              // counts[h] stays 0: a degenerate sample never wins
  int cnt = 0;
  for (int i = blockIdx.x * kC + threadIdx.x; i < n; i += gridDim.x * kC) {
    const float4 p = pts[i];
    const float d = ((co[0] * p.x + co[1] * p.y) + co[2] * p.z) + co[3];
    cnt += fabsf(d) < thr ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  __shared__ int red[kC / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < kC / 64; ++k) t += red[k];
    if (t) atomicAdd(counts + h, t);          // one atomic per workgroup and hypothesis
  }
}
// inliers of plane `co`: flags, and (optionally) the count and the first / second moments of the inliers for the least-squares refit.
// The moments are accumulated EXACTLY: a coordinate is a float, so x and x y are exact doubles; each term is scaled by 2^shift
// (shift chosen from the cloud's bounds so that |term| < 2^60), rounded ONCE to an integer and added as two integer parts
// (q >> 24 and q & 0xffffff: neither sum can overflow 64 bits below 2^27 points).  Integer sums do not depend on the order of the
// additions, so the refitted plane — and every inlier decision that follows — is reproducible and equals the sequential restatement
// bit for bit (double atomics left the last bits to scheduling: one flipped inlier per ~10 scans).
struct MomI { unsigned long long v[19]; };      // [0] count, then (hi, lo) of sx sy sz xx xy xz yy yz zz
__device__ __forceinline__ void mom_add(long long& hi, unsigned long long& lo, double t, double scale) {
  const long long q = __double2ll_rn(t * scale);          // (t * 2^shift is exact; one rounding to the integer grid)
  hi += q >> 24; lo += (unsigned long long)(q & 0xffffff);
}
// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi; the reference calls pcl::eigen33 here).  IEEE +, -, *, /, sqrt only and no
// contraction (the pragma at the top of the file): the host and the device produce the same bits.
__host__ __device__ inline void smallest_eigvec3(const double A_in[9], double v[3]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) A[k] = A_in[k];
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double akp = A[3 * k + p], akq = A[3 * k + q]; A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = A[3 * p + k], aqk = A[3 * q + k]; A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk; }
        for (int k = 0; k < 3; ++k) { const double vkp = V[3 * k + p], vkq = V[3 * k + q]; V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq; }
      }
  }
  int m = 0;
  for (int k = 1; k < 3; ++k) if (A[4 * k] < A[4 * m]) m = k;
  for (int k = 0; k < 3; ++k) v[k] = V[3 * k + m];
}

// the plane and the moment scale as launches hand them to each other on the device (device-counted pipeline)
struct PlaneP { float co[4]; int valid; int pad; double mom_scale; int mom_shift; int pad2; };
// least-squares plane through the inliers from their exact moments (SACSegmentation's optimizeModelCoefficients); false: fewer than 3 inliers
__host__ __device__ inline bool refit_plane(const MomI& hm, int shift, float co[4]) {
  double m[10];
  m[0] = (double)hm.v[0];
  for (int k = 0; k < 9; ++k) m[1 + k] = ldexp((double)(long long)hm.v[1 + 2 * k] * 16777216.0 + (double)hm.v[2 + 2 * k], -shift);
  if (!(m[0] >= 3.0)) return false;
  const double inv = 1.0 / m[0], cx = m[1] * inv, cy = m[2] * inv, cz = m[3] * inv;
  const double C[9] = {m[4] * inv - cx * cx, m[5] * inv - cx * cy, m[6] * inv - cx * cz, m[5] * inv - cx * cy, m[7] * inv - cy * cy, m[8] * inv - cy * cz,
                       m[6] * inv - cx * cz, m[8] * inv - cy * cz, m[9] * inv - cz * cz};
  double nv[3];
  smallest_eigvec3(C, nv);
  if (nv[2] < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }       // fixed orientation (inlier selection is sign-free)
  co[0] = (float)nv[0]; co[1] = (float)nv[1]; co[2] = (float)nv[2]; co[3] = (float)(-(nv[0] * cx + nv[1] * cy + nv[2] * cz));
  return true;
}
// state of the device-counted PCL tail (dc_pcl_tail below): everything a launch leaves for the next one
constexpr int kDcErrBounds = 1, kDcErrVoxel = 2, kDcErrBits = 4;
struct DcState {
  int cnt[4];                 // points after: surf voxel grid, surf radius filter (= the surf features), ground voxel grid, ground plane (= the ground features)
  int err;                    // kDcErr* bits: the host re-runs the scan through the host-counted path
  int ransac_used;
  int tick[6];                // arrival counters of the launches whose LAST workgroup finishes the job (4 bounds passes, the moment pass)
  unsigned bounds[4][6];      // ordered-uint min xyz | max xyz of: surf picks, surf voxels, ground picks, ground voxels
  VoxP vox[2];
  SortP sortp[2];
  GridC grid;
  int grid_ncell;
  PlaneP plane[2];            // [0] the winning hypothesis, [1] the plane re-fitted to its inliers
  MomI mom;
};
__device__ __forceinline__ unsigned long long dc_load_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// pp (device-counted): the plane comes from *pp (valid == 0: no model, no inliers), the moment scale from it too, and — with `st` — the LAST
// workgroup to arrive re-fits the plane to the moments and leaves it in st->plane[1] (the first plane again when there are fewer than 3 inliers)
__global__ __launch_bounds__(kC) void k_plane_inliers(int n, const int* __restrict__ n_dev, const float4* __restrict__ pts, float c0, float c1, float c2, float c3,
                                                      const PlaneP* __restrict__ pp, float thr, int* __restrict__ flags, MomI* __restrict__ mom, double scale,
                                                      DcState* __restrict__ st) {
  if (n_dev) n = min(n, *n_dev);
  bool model = true;
  if (pp) { c0 = pp->co[0]; c1 = pp->co[1]; c2 = pp->co[2]; c3 = pp->co[3]; scale = pp->mom_scale; model = pp->valid != 0; }
  long long hi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long cnt = 0;
  for (int i = blockIdx.x * kC + threadIdx.x; i < n; i += gridDim.x * kC) {
    const float4 p = pts[i];
    const float d = ((c0 * p.x + c1 * p.y) + c2 * p.z) + c3;
    const int in = (model && fabsf(d) < thr) ? 1 : 0;
    flags[i] = in;
    if (in && mom) {
      const double x = p.x, y = p.y, z = p.z;
      ++cnt;
      mom_add(hi[0], lo[0], x, scale); mom_add(hi[1], lo[1], y, scale); mom_add(hi[2], lo[2], z, scale);
      mom_add(hi[3], lo[3], x * x, scale); mom_add(hi[4], lo[4], x * y, scale); mom_add(hi[5], lo[5], x * z, scale);
      mom_add(hi[6], lo[6], y * y, scale); mom_add(hi[7], lo[7], y * z, scale); mom_add(hi[8], lo[8], z * z, scale);
    }
  }
  if (mom) {
    __shared__ unsigned long long red[kC / 64][19];
    unsigned long long v[19];
    v[0] = cnt;
#pragma unroll
    for (int k = 0; k < 9; ++k) { v[1 + 2 * k] = (unsigned long long)hi[k]; v[2 + 2 * k] = lo[k]; }      // (two's complement: the signed parts add modulo 2^64)
#pragma unroll
    for (int k = 0; k < 19; ++k) {
      unsigned long long t = v[k];
      for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 19) {
      unsigned long long t = 0;
      for (int w = 0; w < kC / 64; ++w) t += red[w][threadIdx.x];
      if (t) red[0][threadIdx.x] = atomicAdd(&mom->v[threadIdx.x], t);      // (returning, result consumed: performed before the arrival below)
    }
  }
  if (st) {
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&st->tick[4], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
      MomI hm;
      for (int k = 0; k < 19; ++k) hm.v[k] = dc_load_u64(&mom->v[k]);
      PlaneP out = *pp;
      float co[4];
      if (model && refit_plane(hm, pp->mom_shift, co)) { out.co[0] = co[0]; out.co[1] = co[1]; out.co[2] = co[2]; out.co[3] = co[3]; }
      st->plane[1] = out;
    }
  }
}

static int new_cloud(lvf_ctx* ctx, int n, lvf_cloud** out) {
  auto* c = new lvf_cloud();
  c->ctx = ctx; c->n = n;
  const int rc = c->pts.alloc(std::max(n, 0));
  if (rc != LVF_OK) { delete c; return rc; }
  *out = c;
  return LVF_OK;
}
// out (a NEW cloud) = the flagged points of pts[0..n), input order preserved — shared with extract_kernels.hip
// (device_scan1's tiles read every earlier tile's sum: fine for thousands of tiles, not for a 30 M-cell grid — beyond this the three-launch scan)
constexpr int kScan1Max = 1 << 22;
static int scan_i32(lvf_ctx* ctx, const int* in, int n, int* pos) {
  if (n <= kScan1Max) return device_scan1(ctx, in, n, nullptr, pos, nullptr, nullptr, nullptr);
  return device_exclusive_scan_i32(ctx, in, n, pos);
}
int compact_points(lvf_ctx* ctx, const float4* pts, int n, const int* flags_dev, lvf_cloud** out) {
  if (n > kScan1Max) {
    hipStream_t s = ctx->stream;
    DevBuf<int> pos;
    LVF_TRY(pos.alloc((size_t)n + 1));
    LVF_TRY(device_exclusive_scan_i32(ctx, flags_dev, n, pos.p));
    int total = 0;
    LVF_TRY(read_back(ctx, &total, pos.p + n, sizeof(int)));
    LVF_TRY(new_cloud(ctx, total, out));
    if (total) hipLaunchKernelGGL(k_compact, dim3(gridc(n)), dim3(kC), 0, s, n, pts, flags_dev, pos.p, (*out)->pts.p);
    LVF_HIP(hipGetLastError());
    LVF_HIP(hipStreamSynchronize(s));
    return LVF_OK;
  }
  // ONE launch (device_scan1: scan + compaction) into a block of the input's capacity, then the count comes back (round 4: three scan launches,
  // the count, a block of exactly that size, the compaction launch, a second wait)
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(ctx, n, &c));
  DevBuf<int> total_dev;
  int rc = total_dev.alloc(1);
  if (rc == LVF_OK) rc = device_scan1(ctx, flags_dev, n, nullptr, nullptr, total_dev.p, pts, c->pts.p);
  int total = 0;
  if (rc == LVF_OK) rc = read_back(ctx, &total, total_dev.p, sizeof(int));
  if (rc != LVF_OK) { delete c; return rc; }
  c->n = total; c->pts.n = (size_t)total;
  *out = c;
  return LVF_OK;
}
// out = the flagged points of `in`, input order preserved
static int compact_cloud(const lvf_cloud* in, const int* flags_dev, lvf_cloud** out) {
  return compact_points(in->ctx, in->pts.p, in->n, flags_dev, out);
}

// ============================================================================================== the PCL tail with the counts on the device
// FeatureAssociation::Process' tail (association.cpp:210-234: VoxelGrid + RadiusOutlierRemoval on the surf picks, VoxelGrid + SegmentGround on
// the ground picks) as ONE launch chain without a host wait: round 4 ran the four lvf_cloud_* entry points above one after the other, each of
// them reading a bounding box and a point count back (9 stream waits of ~22 us, 118 device events per scan).  Here a cloud is (buffer of the
// INPUT's capacity, count in device memory); every launch is sized by the capacity and reads the count; what the host computed between
// launches — the voxel grid's origin and dimensions, the sort's digit split, the radius grid, RANSAC's bookkeeping, the least-squares refit —
// is computed by the LAST workgroup of the launch that produces its inputs (arrival counter behind a fence) or by a one-thread launch, with
// the same arithmetic.  Results are the host-counted path's bit for bit (tests/test_gpu_extract.py runs both).
__global__ void k_dc_init(DcState* __restrict__ st, int* __restrict__ ransac_counts, int n_ransac, int cnt2) {
  const int t = threadIdx.x;
  for (int i = t; i < n_ransac; i += blockDim.x) ransac_counts[i] = 0;
  if (t < 4) st->cnt[t] = t == 2 ? cnt2 : 0;          // (cnt2: lvf_cloud_segment_plane enters the chain at the plane fit with a host-known count)
  if (t < 6) st->tick[t] = 0;
  if (t < 24) st->bounds[t / 6][t % 6] = (t % 6) < 3 ? 0xffffffffu : 0u;
  if (t < 19) st->mom.v[t] = 0ull;
  if (t == 0) { st->err = 0; st->ransac_used = 0; st->grid_ncell = 1; st->plane[0].valid = 0; st->plane[1].valid = 0; }
}
__device__ __forceinline__ float ord2f_dev(unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; return __uint_as_float(u); }
// what the last workgroup of a bounds pass derives from the box
constexpr int kDcBoundsBlocks = 64;      // (arrival atomics on one address: a few dozen workgroups, not hundreds)
struct DcSetup {
  int mode;          // 0: voxel grid `which` (VoxP + the sort's SortP), 1: the radius filter's grid, 2: the plane fit's moment scale
  int slot;          // bounds[slot], tick[slot]
  int which;
  float f0;          // 0: 1 / leaf;  1: the radius
  int i0;            // 0: digit passes the host enqueued;  1: cells the grid arrays hold
};
__global__ __launch_bounds__(kC) void k_dc_bounds(int cap, const int* __restrict__ n_dev, const float4* __restrict__ pts, DcState* __restrict__ st, DcSetup su) {
  const int n = min(cap, *n_dev);
  unsigned* bounds = st->bounds[su.slot];
  float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
  for (int i = blockIdx.x * kC + threadIdx.x; i < n; i += gridDim.x * kC) {
    const float4 p = pts[i];
    x = fminf(x, p.x); y = fminf(y, p.y); z = fminf(z, p.z); X = fmaxf(X, p.x); Y = fmaxf(Y, p.y); Z = fmaxf(Z, p.z);
  }
  for (int o = 32; o > 0; o >>= 1) {
    x = fminf(x, __shfl_down(x, o)); y = fminf(y, __shfl_down(y, o)); z = fminf(z, __shfl_down(z, o));
    X = fmaxf(X, __shfl_down(X, o)); Y = fmaxf(Y, __shfl_down(Y, o)); Z = fmaxf(Z, __shfl_down(Z, o));
  }
  __shared__ float red[kC / 64][6];
  __shared__ int s_last;
  __shared__ unsigned s_seen[6];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = x; red[w][1] = y; red[w][2] = z; red[w][3] = X; red[w][4] = Y; red[w][5] = Z; }
  __syncthreads();
  // Only workgroups that saw points touch the box, and they use RETURNING atomics whose result is consumed (stored to LDS): the wave waits
  // for them, so they are performed — at the coherence point, like every device-scope atomic — before the arrival below.  (A release fence
  // here, 256 workgroups strong, made this launch 19 us.)
  if (threadIdx.x < 6 && (long long)blockIdx.x * kC < (long long)n) {
    float v = red[0][threadIdx.x];
    for (int k = 1; k < kC / 64; ++k) v = threadIdx.x < 3 ? fminf(v, red[k][threadIdx.x]) : fmaxf(v, red[k][threadIdx.x]);
    s_seen[threadIdx.x] = threadIdx.x < 3 ? atomicMin(bounds + threadIdx.x, f2ord_c(v)) : atomicMax(bounds + threadIdx.x, f2ord_c(v));
  }
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&st->tick[su.slot], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  float lo[3], hi[3];
  bool finite = true;
  for (int k = 0; k < 3; ++k) {
    lo[k] = ord2f_dev(__hip_atomic_load(bounds + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    hi[k] = ord2f_dev(__hip_atomic_load(bounds + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    finite = finite && isfinite(lo[k]) && isfinite(hi[k]);
  }
  if (n > 0 && !finite) atomicOr(&st->err, kDcErrBounds);
  const bool live = n > 0 && finite;
  if (su.mode == 0) {                                  // lvf_cloud_voxel_filter's host arithmetic
    VoxP v; v.inv_leaf = su.f0;
    long long ncell = 1;
    for (int k = 0; k < 3; ++k) {
      v.minb[k] = live ? (int)floorf(lo[k] * v.inv_leaf) : 0;
      const int maxb = live ? (int)floorf(hi[k] * v.inv_leaf) : 0;
      v.div[k] = maxb - v.minb[k] + 1;
      ncell *= v.div[k];
    }
    if (live && !(ncell > 0 && ncell <= (1ll << 26))) { atomicOr(&st->err, kDcErrVoxel); ncell = 1; }
    int bits = 1;
    while ((1ll << bits) < ncell) ++bits;
    if (bits > kSortMaxDigitBits * su.i0) { atomicOr(&st->err, kDcErrBits); bits = kSortMaxDigitBits * su.i0; }
    SortP sp; sp.n = live ? n : 0; sp.bits = bits; sp.db = (bits + su.i0 - 1) / su.i0; sp.nbins = 1 << sp.db;
    st->vox[su.which] = v; st->sortp[su.which] = sp;
  } else if (su.mode == 1) {                           // lvf_cloud_radius_outlier_filter's
    float cell = su.f0 * 1.0001f;
    int d[3] = {1, 1, 1};
    if (live) {
      for (;;) {
        long long t = 1;
        for (int k = 0; k < 3; ++k) { d[k] = (int)floorf((hi[k] - lo[k]) / cell) + 1; t *= d[k]; }
        if (t <= (long long)su.i0) break;
        cell *= 1.5f;                                  // bigger cells stay correct, only slower
      }
    }
    GridC g; g.ox = live ? lo[0] : 0.0f; g.oy = live ? lo[1] : 0.0f; g.oz = live ? lo[2] : 0.0f; g.inv_cell = 1.0f / cell; g.nx = d[0]; g.ny = d[1]; g.nz = d[2];
    st->grid = g; st->grid_ncell = d[0] * d[1] * d[2];
  } else {                                             // lvf_cloud_segment_plane's: |coordinate| < 2^e  =>  |x y| 2^shift < 2^60
    float maxabs = 0.0f;
    if (live) for (int k = 0; k < 3; ++k) maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
    int e2 = 0;
    (void)frexpf(maxabs, &e2);
    const int shift = 60 - 2 * max(e2, 0);
    st->plane[0].mom_shift = shift; st->plane[0].mom_scale = ldexp(1.0, shift);
  }
}
// pcl::RandomSampleConsensus::computeModel's bookkeeping over the hypothesis counts, in hypothesis order, and the winning plane from its three
// sample points: lvf_cloud_segment_plane's host code, one thread
__global__ void k_ransac_pick(const int* __restrict__ n_dev, int cap, const float4* __restrict__ pts, const int* __restrict__ counts, int max_iterations,
                              unsigned long long seed, DcState* __restrict__ st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = min(cap, *n_dev);
  PlaneP out = st->plane[0];
  out.valid = 0; out.co[0] = out.co[1] = out.co[2] = out.co[3] = 0.0f;
  int best = -1, best_count = 0, used = 0;
  if (n >= 3) {
    double k = 1.0;
    const double log_probability = log(1.0 - 0.99);
    for (int h = 0; h < max_iterations && (double)h < k; ++h) {
      used = h + 1;
      const int c = counts[h];
      if (c > best_count) {
        best_count = c; best = h;
        const double w = (double)best_count / (double)n;
        double p_no_outliers = 1.0 - w * w * w;
        p_no_outliers = fmax(2.220446049250313e-16, p_no_outliers);
        p_no_outliers = fmin(1.0 - 2.220446049250313e-16, p_no_outliers);
        k = log_probability / log(p_no_outliers);
      }
    }
  }
  if (best >= 0) {
    int id[3];
    sample3(seed, best, n, id);
    float co[4] = {0, 0, 0, 0};
    if (plane_from3(pts[id[0]], pts[id[1]], pts[id[2]], co)) { out.valid = 1; out.co[0] = co[0]; out.co[1] = co[1]; out.co[2] = co[2]; out.co[3] = co[3]; }
  }
  st->ransac_used = used;
  st->plane[0] = out;
}

int dc_state_bytes() { return (int)sizeof(DcState); }
int dc_state_counts_offset() { return (int)offsetof(DcState, cnt); }

// capacity-sized scratch of one voxel filter
struct DcVoxelTmp { DevBuf<int> flags, pos, val, order; DevBuf<unsigned> key, key_sorted; SortScratch sort; };
// One voxel filter of the tail: in (pts, *n_in <= cap) -> out (capacity cap), *n_out, enqueued on q (scan lane `lane`); `which`: vox[] / sortp[]
// slot, `slot`: bounds[] / tick[] slot.  The jobs of a call are enqueued launch by launch in turn (see device_sort_pairs_u32_dc_multi).
struct DcVoxelJob { hipStream_t q; int lane, which, slot; const float4* in; const int* n_in; DcVoxelTmp* t; float4* out; int* n_out; };
static int dc_voxel_filters(lvf_ctx* ctx, DcState* st, int n_jobs, DcVoxelJob* jobs, int cap, float leaf, int passes) {
  SortJobDc sj[2];
  for (int k = 0; k < n_jobs; ++k) {
    DcVoxelTmp& t = *jobs[k].t;
    LVF_TRY(t.flags.alloc(cap)); LVF_TRY(t.pos.alloc((size_t)cap + 1)); LVF_TRY(t.key.alloc(cap)); LVF_TRY(t.key_sorted.alloc(cap)); LVF_TRY(t.val.alloc(cap)); LVF_TRY(t.order.alloc(cap));
    sj[k] = SortJobDc{jobs[k].q, t.key.p, t.key_sorted.p, t.val.p, t.order.p, &st->sortp[jobs[k].which], &t.sort, nullptr, nullptr, nullptr, nullptr};
  }
  for (int k = 0; k < n_jobs; ++k) {
    const DcVoxelJob& J = jobs[k];
    DcSetup su; su.mode = 0; su.slot = J.slot; su.which = J.which; su.f0 = 1.0f / leaf; su.i0 = passes;
    hipLaunchKernelGGL(k_dc_bounds, dim3(std::min(kDcBoundsBlocks, gridc(cap))), dim3(kC), 0, J.q, cap, J.n_in, J.in, st, su);
  }
  for (int k = 0; k < n_jobs; ++k) {
    const DcVoxelJob& J = jobs[k];
    hipLaunchKernelGGL(k_voxel_key, dim3(gridc(cap)), dim3(kC), 0, J.q, cap, J.n_in, J.in, VoxP{}, &st->vox[J.which], J.t->key.p, J.t->val.p);
  }
  LVF_HIP(hipGetLastError());
  LVF_TRY(device_sort_pairs_u32_dc_multi(ctx, n_jobs, sj, cap, passes));
  for (int k = 0; k < n_jobs; ++k) {
    const DcVoxelJob& J = jobs[k];
    hipLaunchKernelGGL(k_voxel_heads, dim3(gridc(cap)), dim3(kC), 0, J.q, cap, J.n_in, J.t->key_sorted.p, J.t->flags.p);
  }
  LVF_HIP(hipGetLastError());
  for (int k = 0; k < n_jobs; ++k) {
    const DcVoxelJob& J = jobs[k];
    LVF_TRY(device_scan1_on(ctx, J.q, J.lane, J.t->flags.p, cap, J.n_in, J.t->pos.p, J.n_out, nullptr, nullptr));
  }
  for (int k = 0; k < n_jobs; ++k) {
    const DcVoxelJob& J = jobs[k];
    hipLaunchKernelGGL(k_voxel_emit, dim3(gridc(cap)), dim3(kC), 0, J.q, cap, J.n_in, J.in, J.t->key_sorted.p, J.t->order.p, J.t->flags.p, J.t->pos.p, J.out);
  }
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

// The tail.  surf_raw / ground_raw: the picks (capacity `cap`, counts on the device).  Every coordinate is known to lie within max_range of the
// origin (Preprocess' range gate; centroids stay inside), which bounds the voxel keys' width and the radius grid's size BEFORE anything is
// measured; a scan that breaks the bound raises st->err.  The surf chain (VoxelGrid -> RadiusOutlierRemoval) and the ground chain (VoxelGrid ->
// plane) share nothing: the surf chain goes on the context's stream, the ground chain on its side stream, forked behind what produced the picks
// and joined before this returns — two chains of ~25 launches of a few microseconds each run beside each other instead of one after the other.
// Scratch of BOTH chains lives in `keep` until the caller has waited for the stream (the pool hands a released block to the next allocation
// in HOST order, which is only safe on one stream).  On return (nothing waited for): surf_out / ground_out hold the features before the
// extrinsic transform, st->cnt their counts (and the two intermediate ones), st->err the verdict.
struct DcTailKeep {
  DevBuf<int> ransac_counts, cell_of, counts2, start, flags_s, flags_g;
  DevBuf<float4> sv, gv, sorted;
  DcVoxelTmp vs, vg;
};
// begin: the a-priori bounds (plan->supported = false when the scan's parameters are outside them: nothing is launched), the state block and its
// clears — enqueued by the caller ahead of everything that depends on the scan, so they run while the host prepares the upload
int dc_tail_begin(lvf_ctx* ctx, int cap, float resolution, float max_range, DevBuf<unsigned char>& state, std::shared_ptr<void>& keep_out, DcTailPlan* plan) {
  hipStream_t s = ctx->stream;
  plan->supported = false;
  plan->cap = cap;
  plan->leaf = 2 * resolution; plan->radius = 4 * resolution; plan->thr = 0.1f * resolution;
  plan->max_iterations = 100; plan->min_neighbors = 4;
  // a-priori widths: coordinates in [-max_range, max_range]
  const double span_v = std::floor((double)max_range / plan->leaf) - std::floor(-(double)max_range / plan->leaf) + 1.0 + 2.0;      // (+2: float rounding of p * inv_leaf at the ends)
  const double ncell_v = span_v * span_v * span_v;
  if (!(ncell_v > 0 && ncell_v <= (double)(1ll << 26))) return LVF_OK;
  int bits = 1;
  while ((double)(1ll << bits) < ncell_v) ++bits;
  plan->passes = (bits + kSortMaxDigitBits - 1) / kSortMaxDigitBits;
  const double span_g = std::floor(2.0 * max_range / (plan->radius * 1.0001f)) + 2.0;
  const double ncell_g = span_g * span_g * span_g;
  if (!(ncell_g <= (double)(1 << 22)) || cap <= 0 || cap > (1 << 18)) return LVF_OK;
  plan->grid_cells = (int)ncell_g;
  static const bool one_stream = [] { const char* e = std::getenv("LVF_EXTRACT_ONE_STREAM"); return e && e[0] == '1'; }();      // A/B
  plan->two_streams = !one_stream;
  plan->lane_ground = one_stream ? 0 : 1;
  auto keep = std::make_shared<DcTailKeep>();
  keep_out = keep;
  DcTailKeep& K = *keep;
  LVF_TRY(state.alloc(sizeof(DcState)));
  LVF_TRY(K.ransac_counts.alloc(plan->max_iterations));
  LVF_TRY(K.counts2.alloc((size_t)2 * plan->grid_cells));
  hipLaunchKernelGGL(k_dc_init, dim3(1), dim3(64), 0, s, reinterpret_cast<DcState*>(state.p), K.ransac_counts.p, plan->max_iterations, 0);
  LVF_HIP(hipMemsetAsync(K.counts2.p, 0, (size_t)8 * plan->grid_cells, s));      // the radius grid's counts | cursor (one block, one clear)
  LVF_HIP(hipGetLastError());
  plan->supported = true;
  return LVF_OK;
}
// fork: *q = the stream of the ground half — the side stream, ordered behind everything enqueued on the context's stream so far
int dc_tail_fork(lvf_ctx* ctx, const DcTailPlan& plan, hipStream_t* q) {
  *q = ctx->stream;
  if (!plan.two_streams) return LVF_OK;
  LVF_TRY(side_stream(ctx, q));
  LVF_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  LVF_HIP(hipStreamWaitEvent(*q, ctx->ev_fork, 0));
  return LVF_OK;
}
// run: the two chains and the join (the context's stream waits for the side stream's last launch)
int dc_tail_run(lvf_ctx* ctx, const DcTailPlan& plan, hipStream_t q, const float4* surf_raw, const int* n_surf_raw, const float4* ground_raw, const int* n_ground_raw,
                unsigned long long seed, DevBuf<unsigned char>& state, std::shared_ptr<void>& keep, DevBuf<float4>& surf_out, DevBuf<float4>& ground_out) {
  hipStream_t s = ctx->stream;
  DcTailKeep& K = *static_cast<DcTailKeep*>(keep.get());
  DcState* st = reinterpret_cast<DcState*>(state.p);
  const int cap = plan.cap, lane_g = plan.lane_ground;
  LVF_TRY(surf_out.alloc(cap)); LVF_TRY(ground_out.alloc(cap));
  LVF_TRY(K.sv.alloc(cap)); LVF_TRY(K.gv.alloc(cap));
  // ---- both VoxelGrids (2 res), surf on the context's stream and ground on the side stream                 association.cpp:210-215,222-224
  {
    DcVoxelJob jobs[2] = {{q, lane_g, 1, 2, ground_raw, n_ground_raw, &K.vg, K.gv.p, &st->cnt[2]}, {s, 0, 0, 0, surf_raw, n_surf_raw, &K.vs, K.sv.p, &st->cnt[0]}};
    LVF_TRY(dc_voxel_filters(ctx, st, 2, jobs, cap, plan.leaf, plan.passes));
  }
  // ---- ground: SegmentGround (RANSAC plane, 100 its, 0.1 res)  association.cpp:225-234,249-268 | surf: RadiusOutlierRemoval(4 res, 4)  :217-221
  // (launch by launch in turn, as above)
  const int grid_cells = plan.grid_cells;
  DevBuf<int>& rcounts = K.ransac_counts;              // (cleared by k_dc_init)
  LVF_TRY(K.flags_g.alloc(cap));
  LVF_TRY(K.cell_of.alloc(cap)); LVF_TRY(K.start.alloc((size_t)grid_cells + 1)); LVF_TRY(K.flags_s.alloc(cap)); LVF_TRY(K.sorted.alloc(cap));
  int* counts = K.counts2.p; int* cursor = K.counts2.p + grid_cells;      // (cleared by dc_tail_begin)
  const int* n_gv = &st->cnt[2]; const int* n_sv = &st->cnt[0];
  const int gb = std::min(kDcBoundsBlocks, gridc(cap));
  DcSetup sug; sug.mode = 2; sug.slot = 3; sug.which = 0; sug.f0 = 0.0f; sug.i0 = 0;
  DcSetup sus; sus.mode = 1; sus.slot = 1; sus.which = 0; sus.f0 = plan.radius; sus.i0 = grid_cells;
  hipLaunchKernelGGL(k_dc_bounds, dim3(gb), dim3(kC), 0, q, cap, n_gv, K.gv.p, st, sug);
  hipLaunchKernelGGL(k_dc_bounds, dim3(gb), dim3(kC), 0, s, cap, n_sv, K.sv.p, st, sus);
  hipLaunchKernelGGL(k_ransac_count, dim3(std::min(gridc(cap), 8), plan.max_iterations), dim3(kC), 0, q, cap, n_gv, K.gv.p, seed, plan.thr, rcounts.p);
  hipLaunchKernelGGL(k_grid_count, dim3(gridc(cap)), dim3(kC), 0, s, cap, n_sv, K.sv.p, GridC{}, &st->grid, K.cell_of.p, counts);
  hipLaunchKernelGGL(k_ransac_pick, dim3(1), dim3(64), 0, q, n_gv, cap, K.gv.p, rcounts.p, plan.max_iterations, seed, st);
  LVF_HIP(hipGetLastError());
  LVF_TRY(device_scan1_on(ctx, s, 0, counts, grid_cells, &st->grid_ncell, K.start.p, nullptr, nullptr, nullptr));
  hipLaunchKernelGGL(k_plane_inliers, dim3(gb), dim3(kC), 0, q, cap, n_gv, K.gv.p, 0.0f, 0.0f, 0.0f, 0.0f, &st->plane[0], plan.thr, K.flags_g.p, &st->mom, 0.0, st);
  hipLaunchKernelGGL(k_grid_scatter, dim3(gridc(cap)), dim3(kC), 0, s, cap, n_sv, K.sv.p, K.cell_of.p, K.start.p, cursor, K.sorted.p);
  hipLaunchKernelGGL(k_plane_inliers, dim3(gb), dim3(kC), 0, q, cap, n_gv, K.gv.p, 0.0f, 0.0f, 0.0f, 0.0f, &st->plane[1], plan.thr, K.flags_g.p, (MomI*)nullptr, 0.0, (DcState*)nullptr);
  hipLaunchKernelGGL(k_radius_count, dim3(gridc(cap)), dim3(kC), 0, s, cap, n_sv, K.sv.p, GridC{}, &st->grid, K.start.p, K.sorted.p, plan.radius * plan.radius, plan.min_neighbors,
                     K.flags_s.p);
  LVF_HIP(hipGetLastError());
  LVF_TRY(device_scan1_on(ctx, q, lane_g, K.flags_g.p, cap, n_gv, nullptr, &st->cnt[3], K.gv.p, ground_out.p));
  LVF_TRY(device_scan1_on(ctx, s, 0, K.flags_s.p, cap, n_sv, nullptr, &st->cnt[1], K.sv.p, surf_out.p));
  if (plan.two_streams) LVF_HIP(hipEventRecord(ctx->ev_join, q));
  if (plan.two_streams) LVF_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));
  return LVF_OK;
}
// out = a new cloud of n points: pose * in[0 .. n)  (Sensor2Robot, association.cpp:236-247)
int transform_points(lvf_ctx* ctx, const float4* in, int n, const double* pose, lvf_cloud** out) {
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(ctx, n, &c));
  TfArgC tf;
  for (int k = 0; k < 7; ++k) tf.v[k] = (float)pose[k];
  if (n) hipLaunchKernelGGL(k_cloud_transform, dim3(gridc(n)), dim3(kC), 0, ctx->stream, n, in, tf, c->pts.p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { delete c; return ::lvf::hip_fail(e, "k_cloud_transform", __FILE__, __LINE__); }
  *out = c;
  return LVF_OK;
}
}  // namespace lvf

using namespace lvf;

extern "C" {

int lvf_cloud_create(lvf_ctx* ctx, const float* points, int n, int stride_floats, int intensity_offset, lvf_cloud** out) {
  LVF_REQUIRE(ctx && out, "lvf_cloud_create: null ctx/out");
  LVF_REQUIRE(n >= 0 && (n == 0 || points) && stride_floats >= 3 && intensity_offset < stride_floats, "lvf_cloud_create: bad cloud (n=%d stride=%d ioff=%d)", n,
              stride_floats, intensity_offset);
  LVF_TRY(lvf::enter(ctx));
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(ctx, n, &c));
  if (n) {
    DevBuf<float> src;
    HostPin<float> stage;                // (pooled pinned staging: DevBuf::upload_staged)
    StreamWaitGuard stage_guard(ctx->stream);      // ... which every path out of this scope waits for before the block returns to the pool
    int rc = src.upload_staged(points, (size_t)n * stride_floats, ctx->stream, stage);
    if (rc != LVF_OK) { delete c; return rc; }
    hipLaunchKernelGGL(k_cloud_pack, dim3(gridc(n)), dim3(kC), 0, ctx->stream, n, src.p, stride_floats, intensity_offset, c->pts.p);
    { const hipError_t e = hipGetLastError(); if (e != hipSuccess) { delete c; return ::lvf::hip_fail(e, "k_cloud_pack", __FILE__, __LINE__); } }      // (no leak of `c`)
  }
  *out = c;
  return LVF_OK;
}
int lvf_cloud_destroy(lvf_cloud* c) { delete c; return LVF_OK; }
int lvf_cloud_size(const lvf_cloud* c) { return c ? c->n : -1; }
int lvf_cloud_download(const lvf_cloud* c, float* xyzi) {
  LVF_REQUIRE(c && (c->n == 0 || xyzi), "lvf_cloud_download: null argument");
  if (c->n) LVF_HIP(hipMemcpyAsync(xyzi, c->pts.p, (size_t)c->n * sizeof(float4), hipMemcpyDeviceToHost, c->ctx->stream));
  LVF_HIP(hipStreamSynchronize(c->ctx->stream));
  return LVF_OK;
}

int lvf_cloud_transform(const lvf_cloud* in, const double* pose, lvf_cloud** out) {
  LVF_REQUIRE(in && pose && out, "lvf_cloud_transform: null argument");
  LVF_TRY(lvf::enter(in->ctx));
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(in->ctx, in->n, &c));
  TfArgC tf;
  for (int k = 0; k < 7; ++k) tf.v[k] = (float)pose[k];          // Sophus SE3d::cast<float>()  mapping.cpp:195
  if (in->n) hipLaunchKernelGGL(k_cloud_transform, dim3(gridc(in->n)), dim3(kC), 0, in->ctx->stream, in->n, in->pts.p, tf, c->pts.p);
  LVF_HIP(hipGetLastError());
  *out = c;
  return LVF_OK;
}

int lvf_cloud_concat(lvf_ctx* ctx, const lvf_cloud* const* parts, int n_parts, lvf_cloud** out) {
  LVF_REQUIRE(ctx && out && n_parts >= 0 && (n_parts == 0 || parts), "lvf_cloud_concat: bad argument");
  long long total = 0;
  for (int k = 0; k < n_parts; ++k) { LVF_REQUIRE(parts[k] && parts[k]->ctx == ctx, "lvf_cloud_concat: part %d is null or of another context", k); total += parts[k]->n; }
  LVF_REQUIRE(total < (1ll << 31), "lvf_cloud_concat: too many points");
  LVF_TRY(lvf::enter(ctx));
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(ctx, (int)total, &c));
  size_t off = 0;
  for (int k = 0; k < n_parts; ++k) {
    if (parts[k]->n) LVF_HIP(hipMemcpyAsync(c->pts.p + off, parts[k]->pts.p, (size_t)parts[k]->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    off += parts[k]->n;
  }
  *out = c;
  return LVF_OK;
}

// FeatureAssociation::AlignScan (association.cpp:39-64): the keyframe's sweep cut out of two consecutive raw revolutions.
// pc1 / pc2 = the revolutions stamped stamp1 < stamp2 (the map entries either side of raw_point_clouds_.upper_bound(time)); a revolution
// stamped t covers [t - cycle/2, t + cycle/2].  The slice bounds are the reference's own index arithmetic, evaluated in double and
// truncated toward zero like its iterator offsets:  size * (time - start - cycle/2) / (end - start)  and  size * (time - start + cycle/2) / (end - start)
// with start = stamp1 - cycle/2, end = stamp2 + cycle/2, size = |pc1| + |pc2|.  *aligned = 0 (and an empty cloud) where the reference
// returns false: the keyframe's sweep is not covered by the two revolutions.
int lvf_cloud_align_scan(const lvf_cloud* pc1, double stamp1, const lvf_cloud* pc2, double stamp2, double cycle_time, double time, lvf_cloud** out, int* aligned) {
  LVF_REQUIRE(pc1 && pc2 && out && aligned, "lvf_cloud_align_scan: null argument");
  LVF_REQUIRE(pc1->ctx == pc2->ctx, "lvf_cloud_align_scan: clouds of two contexts");
  LVF_REQUIRE(cycle_time > 0.0 && stamp2 > stamp1, "lvf_cloud_align_scan: stamps must increase and the cycle time be positive");
  lvf_ctx* ctx = pc1->ctx;
  LVF_TRY(lvf::enter(ctx));
  *aligned = 0;
  const double end_time = stamp2 + cycle_time / 2, start_time = stamp1 - cycle_time / 2;
  const int size = pc1->n + pc2->n;
  if (time - cycle_time / 2 < start_time || time + cycle_time / 2 > end_time) return new_cloud(ctx, 0, out);
  const long long first = (long long)(size * (time - start_time - cycle_time / 2) / (end_time - start_time));
  const long long last = (long long)(size * (time - start_time + cycle_time / 2) / (end_time - start_time));
  LVF_REQUIRE(first >= 0 && last >= first && last <= size, "lvf_cloud_align_scan: slice [%lld, %lld) outside the %d points", first, last, size);
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(ctx, (int)(last - first), &c));
  if (last > first) {
    hipLaunchKernelGGL(k_align_slice, dim3(gridc((int)(last - first))), dim3(kC), 0, ctx->stream, (int)first, (int)(last - first), pc1->pts.p, pc1->n, pc2->pts.p, c->pts.p);
    LVF_HIP(hipGetLastError());
  }
  *aligned = 1;
  *out = c;
  return LVF_OK;
}

int lvf_cloud_voxel_filter(const lvf_cloud* in, float leaf, lvf_cloud** out) {
  LVF_REQUIRE(in && out, "lvf_cloud_voxel_filter: null argument");
  LVF_REQUIRE(leaf > 0.0f && std::isfinite(leaf), "lvf_cloud_voxel_filter: leaf must be finite > 0");
  lvf_ctx* ctx = in->ctx;
  LVF_TRY(lvf::enter(ctx));
  if (in->n == 0) return new_cloud(ctx, 0, out);
  hipStream_t s = ctx->stream;
  float lo[3], hi[3];
  LVF_TRY(cloud_bounds(in, lo, hi));
  VoxP v;
  v.inv_leaf = 1.0f / leaf;
  long long ncell = 1;
  for (int k = 0; k < 3; ++k) {
    v.minb[k] = (int)std::floor(lo[k] * v.inv_leaf);
    const int maxb = (int)std::floor(hi[k] * v.inv_leaf);
    v.div[k] = maxb - v.minb[k] + 1;
    ncell *= v.div[k];
  }
  // PCL refuses such grids too ("Leaf size is too small for the input dataset. Integer indices would overflow.")
  LVF_REQUIRE(ncell > 0 && ncell <= (1ll << 26), "lvf_cloud_voxel_filter: %lld voxels: leaf size too small for the cloud's extent", ncell);
  DevBuf<int> flags, pos, val, order; DevBuf<unsigned> key, key_sorted;
  LVF_TRY(flags.alloc(in->n)); LVF_TRY(pos.alloc((size_t)in->n + 1));
  LVF_TRY(key.alloc(in->n)); LVF_TRY(key_sorted.alloc(in->n)); LVF_TRY(val.alloc(in->n)); LVF_TRY(order.alloc(in->n));
  hipLaunchKernelGGL(k_voxel_key, dim3(gridc(in->n)), dim3(kC), 0, s, in->n, (const int*)nullptr, in->pts.p, v, (const VoxP*)nullptr, key.p, val.p);
  LVF_HIP(hipGetLastError());
  int bits = 1;
  while ((1ll << bits) < ncell) ++bits;
  LVF_TRY(device_sort_pairs_u32(ctx, key.p, key_sorted.p, val.p, order.p, in->n, bits));      // stable: ascending input index inside a voxel
  hipLaunchKernelGGL(k_voxel_heads, dim3(gridc(in->n)), dim3(kC), 0, s, in->n, (const int*)nullptr, key_sorted.p, flags.p);
  LVF_TRY(scan_i32(ctx, flags.p, in->n, pos.p));
  int total = 0;
  LVF_TRY(read_back(ctx, &total, pos.p + in->n, sizeof(int)));
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(ctx, total, &c));
  hipLaunchKernelGGL(k_voxel_emit, dim3(gridc(in->n)), dim3(kC), 0, s, in->n, (const int*)nullptr, in->pts.p, key_sorted.p, order.p, flags.p, pos.p, c->pts.p);
  LVF_HIP(hipGetLastError());
  // (no wait: the temporaries return to the context's pool, which reuses them in stream order; the cloud is consumed on the same stream)
  *out = c;
  return LVF_OK;
}

int lvf_cloud_radius_outlier_filter(const lvf_cloud* in, float radius, int min_neighbors, lvf_cloud** out) {
  LVF_REQUIRE(in && out, "lvf_cloud_radius_outlier_filter: null argument");
  LVF_REQUIRE(radius > 0.0f && std::isfinite(radius) && min_neighbors >= 0, "lvf_cloud_radius_outlier_filter: bad radius / min_neighbors");
  lvf_ctx* ctx = in->ctx;
  LVF_TRY(lvf::enter(ctx));
  if (in->n == 0) return new_cloud(ctx, 0, out);
  hipStream_t s = ctx->stream;
  const int n = in->n;
  if (n <= kScan1Max) {
    // One launch chain and ONE wait (round 4: the bounding box came back first, to size the grid).  The grid arrays are sized by the cloud — 8
    // cells per point, at least 64 k — and the last workgroup of the bounds pass fits the grid into them (cell = radius, enlarged by 1.5 until
    // it fits: bigger cells stay correct, only slower; the kept points do not depend on the cell size).
    const int grid_cells = (int)std::min<long long>(1ll << 22, std::max<long long>(1ll << 16, 8ll * n));
    DevBuf<unsigned char> state; DevBuf<int> cell_of, counts2, start, flags; DevBuf<float4> sorted;
    LVF_TRY(state.alloc(sizeof(DcState)));
    LVF_TRY(cell_of.alloc(n)); LVF_TRY(counts2.alloc((size_t)2 * grid_cells)); LVF_TRY(start.alloc((size_t)grid_cells + 1)); LVF_TRY(flags.alloc(n)); LVF_TRY(sorted.alloc(n));
    DcState* st = reinterpret_cast<DcState*>(state.p);
    const int* n_dev = &st->cnt[2];
    int* counts = counts2.p; int* cursor = counts2.p + grid_cells;
    hipLaunchKernelGGL(k_dc_init, dim3(1), dim3(64), 0, s, st, (int*)nullptr, 0, n);
    LVF_HIP(hipMemsetAsync(counts2.p, 0, (size_t)8 * grid_cells, s));
    DcSetup su; su.mode = 1; su.slot = 1; su.which = 0; su.f0 = radius; su.i0 = grid_cells;
    hipLaunchKernelGGL(k_dc_bounds, dim3(std::min(n > (1 << 18) ? kCapBlocks : kDcBoundsBlocks, gridc(n))), dim3(kC), 0, s, n, n_dev, in->pts.p, st, su);
    hipLaunchKernelGGL(k_grid_count, dim3(gridc(n)), dim3(kC), 0, s, n, n_dev, in->pts.p, GridC{}, &st->grid, cell_of.p, counts);
    LVF_HIP(hipGetLastError());
    LVF_TRY(device_scan1(ctx, counts, grid_cells, &st->grid_ncell, start.p, nullptr, nullptr, nullptr));
    hipLaunchKernelGGL(k_grid_scatter, dim3(gridc(n)), dim3(kC), 0, s, n, n_dev, in->pts.p, cell_of.p, start.p, cursor, sorted.p);
    hipLaunchKernelGGL(k_radius_count, dim3(gridc(n)), dim3(kC), 0, s, n, n_dev, in->pts.p, GridC{}, &st->grid, start.p, sorted.p, radius * radius, min_neighbors, flags.p);
    LVF_HIP(hipGetLastError());
    lvf_cloud* c = nullptr;
    LVF_TRY(new_cloud(ctx, n, &c));
    int h[5] = {0, 0, 0, 0, 0};                       // cnt[4], err
    int rc = device_scan1(ctx, flags.p, n, nullptr, nullptr, &st->cnt[3], in->pts.p, c->pts.p);
    if (rc == LVF_OK) rc = read_back(ctx, h, state.p + offsetof(DcState, cnt), sizeof(h));
    if (rc != LVF_OK) { delete c; return rc; }
    if (h[4] & kDcErrBounds) { delete c; set_error("cloud has non-finite coordinates"); return LVF_ERR_INVALID; }
    c->n = h[3]; c->pts.n = (size_t)h[3];
    *out = c;
    return LVF_OK;
  }
  float lo[3], hi[3];
  LVF_TRY(cloud_bounds(in, lo, hi));
  float cell = radius * 1.0001f;               // strictly larger than the radius: the 27-cell stencil is exhaustive
  auto dims = [&](float c, int d[3]) { long long t = 1; for (int k = 0; k < 3; ++k) { d[k] = (int)std::floor((hi[k] - lo[k]) / c) + 1; t *= d[k]; } return t; };
  int d[3];
  while (dims(cell, d) > (1ll << 25)) cell *= 1.5f;   // bigger cells stay correct, only slower
  const int ncell = d[0] * d[1] * d[2];
  const GridC g{lo[0], lo[1], lo[2], 1.0f / cell, d[0], d[1], d[2]};
  DevBuf<int> cell_of, counts2, start, flags;
  DevBuf<float4> sorted;
  LVF_TRY(cell_of.alloc(in->n)); LVF_TRY(counts2.alloc((size_t)2 * ncell)); LVF_TRY(start.alloc((size_t)ncell + 1));
  LVF_TRY(flags.alloc(in->n)); LVF_TRY(sorted.alloc(in->n));
  struct { int* p; } counts{counts2.p}, cursor{counts2.p + ncell};      // (one block, one clear)
  LVF_HIP(hipMemsetAsync(counts2.p, 0, (size_t)8 * ncell, s));
  hipLaunchKernelGGL(k_grid_count, dim3(gridc(in->n)), dim3(kC), 0, s, in->n, (const int*)nullptr, in->pts.p, g, (const GridC*)nullptr, cell_of.p, counts.p);
  LVF_HIP(hipGetLastError());
  LVF_TRY(scan_i32(ctx, counts.p, ncell, start.p));
  hipLaunchKernelGGL(k_grid_scatter, dim3(gridc(in->n)), dim3(kC), 0, s, in->n, (const int*)nullptr, in->pts.p, cell_of.p, start.p, cursor.p, sorted.p);
  hipLaunchKernelGGL(k_radius_count, dim3(gridc(in->n)), dim3(kC), 0, s, in->n, (const int*)nullptr, in->pts.p, g, (const GridC*)nullptr, start.p, sorted.p, radius * radius, min_neighbors, flags.p);
  LVF_HIP(hipGetLastError());
  return compact_cloud(in, flags.p, out);
}

int lvf_cloud_segment_plane(const lvf_cloud* in, float distance_threshold, int max_iterations, uint64_t seed, lvf_cloud** out, double* coefficients4,
                            int* iterations_used) {
  LVF_REQUIRE(in && out, "lvf_cloud_segment_plane: null argument");
  LVF_REQUIRE(distance_threshold > 0.0f && max_iterations > 0 && max_iterations <= 4096, "lvf_cloud_segment_plane: bad threshold / iteration count");
  lvf_ctx* ctx = in->ctx;
  LVF_TRY(lvf::enter(ctx));
  if (coefficients4) for (int k = 0; k < 4; ++k) coefficients4[k] = 0.0;
  if (iterations_used) *iterations_used = 0;
  if (in->n < 3) return new_cloud(ctx, 0, out);          // SACSegmentation cannot fit a model: no inliers
  hipStream_t s = ctx->stream;
  const int n = in->n;
  // One launch chain and ONE wait (round 4: five — the bounding box for the moment scale, the hypothesis counts, the winner's three points, the
  // moments, the inlier count — with RANSAC's bookkeeping and the refit on the host in between).  The chain is the ground half of the feature
  // extraction's tail: the moment scale by the last workgroup of the bounds pass, pcl::RandomSampleConsensus' bookkeeping and the winning plane
  // by a one-thread launch (k_ransac_pick), the least-squares refit by the last workgroup of the first inlier pass, the inliers' compaction by
  // device_scan1 — the host arithmetic of round 4, on the device.
  DevBuf<unsigned char> state; DevBuf<int> counts, flags;
  LVF_TRY(state.alloc(sizeof(DcState))); LVF_TRY(counts.alloc(max_iterations)); LVF_TRY(flags.alloc(n));
  DcState* st = reinterpret_cast<DcState*>(state.p);
  const int* n_dev = &st->cnt[2];
  hipLaunchKernelGGL(k_dc_init, dim3(1), dim3(64), 0, s, st, counts.p, max_iterations, n);
  DcSetup su; su.mode = 2; su.slot = 3; su.which = 0; su.f0 = 0.0f; su.i0 = 0;
  const int gb = std::min(n > (1 << 18) ? kCapBlocks : kDcBoundsBlocks, gridc(n));
  hipLaunchKernelGGL(k_dc_bounds, dim3(gb), dim3(kC), 0, s, n, n_dev, in->pts.p, st, su);
  hipLaunchKernelGGL(k_ransac_count, dim3(std::min(gridc(n), 8), max_iterations), dim3(kC), 0, s, n, n_dev, in->pts.p, (unsigned long long)seed, distance_threshold, counts.p);
  hipLaunchKernelGGL(k_ransac_pick, dim3(1), dim3(64), 0, s, n_dev, n, in->pts.p, counts.p, max_iterations, (unsigned long long)seed, st);
  hipLaunchKernelGGL(k_plane_inliers, dim3(gb), dim3(kC), 0, s, n, n_dev, in->pts.p, 0.0f, 0.0f, 0.0f, 0.0f, &st->plane[0], distance_threshold, flags.p, &st->mom, 0.0, st);
  hipLaunchKernelGGL(k_plane_inliers, dim3(gb), dim3(kC), 0, s, n, n_dev, in->pts.p, 0.0f, 0.0f, 0.0f, 0.0f, &st->plane[1], distance_threshold, flags.p, (MomI*)nullptr, 0.0, (DcState*)nullptr);
  LVF_HIP(hipGetLastError());
  lvf_cloud* c = nullptr;
  LVF_TRY(new_cloud(ctx, n, &c));
  DcState h;
  int rc = n <= kScan1Max ? device_scan1(ctx, flags.p, n, nullptr, nullptr, &st->cnt[3], in->pts.p, c->pts.p) : LVF_OK;
  static_assert(sizeof(DcState) <= 4096, "DcState comes back through the context's mailbox");
  if (rc == LVF_OK) rc = read_back(ctx, &h, st, sizeof(DcState));
  if (rc != LVF_OK) { delete c; return rc; }
  if (h.err & kDcErrBounds) { delete c; set_error("cloud has non-finite coordinates"); return LVF_ERR_INVALID; }
  if (iterations_used) *iterations_used = h.ransac_used;
  if (coefficients4) for (int q = 0; q < 4; ++q) coefficients4[q] = h.plane[1].valid ? (double)h.plane[1].co[q] : 0.0;
  if (n > kScan1Max) { delete c; return compact_cloud(in, flags.p, out); }      // (beyond the one-launch scan: the three-launch compaction)
  c->n = h.cnt[3]; c->pts.n = (size_t)h.cnt[3];
  *out = c;
  return LVF_OK;
}

}  // extern "C"
