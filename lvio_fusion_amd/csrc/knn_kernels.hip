// knn_kernels.hip — scan-to-map association: float32 SE3 transform + exact 3-nearest-neighbour + gate.
//
// Replaces, per scan-to-map call of src/lvio_fusion/src/association.cpp:
//   pcl::KdTreeFLANN<PointI>::setInputCloud(deep copy)        :278-279, :336-337   -> lvf_map_create
//   SE3TransformPoint<float>(tf, p_i) ; nearestKSearch(.,3,..) :294-296, :352-354   -> k_knn3
//   all-three d2 < threshold gate                              :298-300, :356-358   -> valid flag
//
// MI355X design: a kd-tree traversal is pointer-chasing; instead the map is bucketed into a uniform grid
// (counting sort on device: histogram -> 3-phase exclusive scan -> scatter into a cell-sorted float4 array
// {x,y,z,original index}); a query walks cubic shells of cells outward and stops as soon as the 3rd-best d2 is
// provably smaller than anything unvisited, or the shell lies beyond the gate radius.  x-adjacent cells are
// contiguous in memory, so a shell row is ONE contiguous float4 range.  Arithmetic is the reference's, bit for
// bit: the transform is the float instantiation of the Ceres-1.x normalise-then-rotate polynomial, distances are
// (dx*dx + dy*dy) + dz*dz with explicit round-to-nearest mul/add (no FMA contraction); ties resolve by
// ascending (d2, map index), which makes the result independent of traversal order.
#include <cmath>
#include "lvf_internal.hpp"

// The float32 arithmetic below must round after every multiply and add (bit-exact d2 / transform): the default
// -ffp-contract=fast would fuse them (HIP's own __fmul_rn/__fadd_rn are plain operators compiled with the
// `contract` flag and DO get fused after inlining — checked in the ISA), so contraction is switched off for this
// whole translation unit and the primitives are re-declared below it.
#pragma clang fp contract(off)

namespace lvf {

constexpr int kB = 256;

// round-to-nearest primitives defined UNDER the pragma above (the HIP header versions carry the `contract` flag)
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return __builtin_sqrtf(a); }

struct GridP { float ox, oy, oz, cell, inv_cell; int nx, ny, nz; };

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f; memcpy(&f, &u, 4); return f;
}

// pack strided xyz records into float4 and reduce the bounding box (ordered-uint atomics)
__global__ __launch_bounds__(kB) void k_pack_bounds(int M, const float* __restrict__ src, int stride, float4* __restrict__ dst,
                                                    unsigned* __restrict__ bounds /* min xyz, max xyz */) {
  const int i = blockIdx.x * kB + threadIdx.x;
  float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
  if (i < M) {
    const float* s = src + (size_t)i * stride;
    x = X = s[0]; y = Y = s[1]; z = Z = s[2];
    dst[i] = make_float4(x, y, z, 0.0f);
  }
  for (int o = 32; o > 0; o >>= 1) {
    x = fminf(x, __shfl_down(x, o)); y = fminf(y, __shfl_down(y, o)); z = fminf(z, __shfl_down(z, o));
    X = fmaxf(X, __shfl_down(X, o)); Y = fmaxf(Y, __shfl_down(Y, o)); Z = fmaxf(Z, __shfl_down(Z, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(bounds + 0, f2ord(x)); atomicMin(bounds + 1, f2ord(y)); atomicMin(bounds + 2, f2ord(z));
    atomicMax(bounds + 3, f2ord(X)); atomicMax(bounds + 4, f2ord(Y)); atomicMax(bounds + 5, f2ord(Z));
  }
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_cell, int n) {
  int c = (int)floorf((v - o) * inv_cell);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ __launch_bounds__(kB) void k_cell_count(int M, const float4* __restrict__ pts, GridP g, int* __restrict__ cell_of,
                                                   int* __restrict__ counts) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= M) return;
  const float4 p = pts[i];
  const int c = (cell_coord(p.z, g.oz, g.inv_cell, g.nz) * g.ny + cell_coord(p.y, g.oy, g.inv_cell, g.ny)) * g.nx +
                cell_coord(p.x, g.ox, g.inv_cell, g.nx);
  cell_of[i] = c;
  atomicAdd(counts + c, 1);
}

__global__ __launch_bounds__(kB) void k_count_nonempty(int n, const int* __restrict__ counts, int* __restrict__ total) {
  const int i = blockIdx.x * kB + threadIdx.x;
  const unsigned long long m = __ballot(i < n && counts[i] > 0);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(total, __popcll(m));
}

// 3-phase exclusive scan over `n` ints, 1024 elements per workgroup
constexpr int kScanT = 256, kScanE = 4, kScanChunk = kScanT * kScanE;
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_warp, int& total) {
  // wave64 inclusive scan via shuffles, then across the 4 waves of the workgroup
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int x = v;
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) s_warp[wid] = x;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < kScanT / 64; ++w) { const int s = s_warp[w]; if (w < wid) base += s; tot += s; }
  __syncthreads();
  total = tot;
  return base + x - v;
}
__global__ __launch_bounds__(kScanT) void k_scan_reduce(int n, const int* __restrict__ in, int* __restrict__ block_sums) {
  __shared__ int s_warp[kScanT / 64];
  const int base = blockIdx.x * kScanChunk + threadIdx.x * kScanE;
  int v = 0;
  for (int e = 0; e < kScanE; ++e) if (base + e < n) v += in[base + e];
  int total;
  (void)block_exclusive_scan(v, s_warp, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanT) void k_scan_sums(int nb, int* __restrict__ block_sums, int* __restrict__ grand_total) {
  __shared__ int s_warp[kScanT / 64];
  int carry = 0;
  for (int start = 0; start < nb; start += kScanT) {
    const int i = start + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, s_warp, total);
    if (i < nb) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *grand_total = carry;
}
__global__ __launch_bounds__(kScanT) void k_scan_apply(int n, const int* __restrict__ in, const int* __restrict__ block_sums,
                                                       int* __restrict__ out) {
  __shared__ int s_warp[kScanT / 64];
  const int base = blockIdx.x * kScanChunk + threadIdx.x * kScanE;
  int e_v[kScanE];
  int v = 0;
  for (int e = 0; e < kScanE; ++e) { e_v[e] = (base + e < n) ? in[base + e] : 0; v += e_v[e]; }
  int total;
  int run = block_sums[blockIdx.x] + block_exclusive_scan(v, s_warp, total);
  for (int e = 0; e < kScanE; ++e) { if (base + e < n) out[base + e] = run; run += e_v[e]; }
}

__global__ __launch_bounds__(kB) void k_cell_scatter(int M, const float4* __restrict__ pts, const int* __restrict__ cell_of,
                                                     const int* __restrict__ cell_start, int* __restrict__ cursor,
                                                     float4* __restrict__ sorted) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= M) return;
  const int c = cell_of[i];
  const int pos = cell_start[c] + atomicAdd(cursor + c, 1);
  float4 p = pts[i];
  p.w = __int_as_float(i);
  sorted[pos] = p;
}

__global__ __launch_bounds__(kB) void k_pack(int Q, const float* __restrict__ src, int stride, float4* __restrict__ dst) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= Q) return;
  const float* s = src + (size_t)i * stride;
  dst[i] = make_float4(s[0], s[1], s[2], 0.0f);
}

// ---- the query ----
struct Tf32 { float a[9]; float t[3]; };   // out = 2*((a0 p0 + a1 p1) + a2 p2) + p0 + t0 ...

// float instantiation of QuaternionRotatePoint's coefficients (oracle/se3_ops.h UnitQuatRotate_wxyz), RN ops only
__device__ __forceinline__ Tf32 make_tf32(const float tf[7]) {
  const float q0 = tf[3], q1 = tf[0], q2 = tf[1], q3 = tf[2];
  const float ss = add_rn(add_rn(add_rn(mul_rn(q0, q0), mul_rn(q1, q1)), mul_rn(q2, q2)), mul_rn(q3, q3));
  const float scale = div_rn(1.0f, sqrt_rn(ss));
  const float u0 = mul_rn(scale, q0), u1 = mul_rn(scale, q1), u2 = mul_rn(scale, q2), u3 = mul_rn(scale, q3);
  const float t2 = mul_rn(u0, u1), t3 = mul_rn(u0, u2), t4 = mul_rn(u0, u3);
  const float t5 = mul_rn(-u1, u1), t6 = mul_rn(u1, u2), t7 = mul_rn(u1, u3);
  const float t8 = mul_rn(-u2, u2), t9 = mul_rn(u2, u3), t1 = mul_rn(-u3, u3);
  Tf32 T;
  T.a[0] = add_rn(t8, t1); T.a[1] = sub_rn(t6, t4); T.a[2] = add_rn(t3, t7);
  T.a[3] = add_rn(t4, t6); T.a[4] = add_rn(t5, t1); T.a[5] = sub_rn(t9, t2);
  T.a[6] = sub_rn(t7, t3); T.a[7] = add_rn(t2, t9); T.a[8] = add_rn(t5, t8);
  T.t[0] = tf[4]; T.t[1] = tf[5]; T.t[2] = tf[6];
  return T;
}
__device__ __forceinline__ float tf_row(const float* a, float p0, float p1, float p2, float pk, float t) {
  const float s = add_rn(add_rn(mul_rn(a[0], p0), mul_rn(a[1], p1)), mul_rn(a[2], p2));
  return add_rn(add_rn(mul_rn(2.0f, s), pk), t);
}

__device__ __forceinline__ bool lex_less(float da, int ia, float db, int ib) { return da < db || (da == db && (unsigned)ia < (unsigned)ib); }
__device__ __forceinline__ void best3_push(float d, int i, float bd[3], int bi[3]) {
  if (!lex_less(d, i, bd[2], bi[2])) return;
  if (lex_less(d, i, bd[1], bi[1])) {
    bd[2] = bd[1]; bi[2] = bi[1];
    if (lex_less(d, i, bd[0], bi[0])) { bd[1] = bd[0]; bi[1] = bi[0]; bd[0] = d; bi[0] = i; }
    else { bd[1] = d; bi[1] = i; }
  } else { bd[2] = d; bi[2] = i; }
}

__device__ __forceinline__ void scan_range(const float4* __restrict__ sorted, int lo, int hi, float qx, float qy, float qz,
                                           float bd[3], int bi[3]) {
  for (int j = lo; j < hi; ++j) {
    const float4 m = sorted[j];
    const float dx = sub_rn(qx, m.x), dy = sub_rn(qy, m.y), dz = sub_rn(qz, m.z);
    const float d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
    best3_push(d, __float_as_int(m.w), bd, bi);
  }
}

struct TfArg { float v[7]; };

__global__ __launch_bounds__(kB) void k_knn3(int Q, const float4* __restrict__ scan, const TfArg tfa,
                                             const float4* __restrict__ sorted, const int* __restrict__ cell_start,
                                             const GridP g, float thr, int* __restrict__ idx, float* __restrict__ d2,
                                             uint8_t* __restrict__ valid) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= Q) return;
  const Tf32 T = make_tf32(tfa.v);
  const float4 p = scan[i];
  const float qx = tf_row(T.a + 0, p.x, p.y, p.z, p.x, T.t[0]);
  const float qy = tf_row(T.a + 3, p.x, p.y, p.z, p.y, T.t[1]);
  const float qz = tf_row(T.a + 6, p.x, p.y, p.z, p.z, T.t[2]);
  const int cx = cell_coord(qx, g.ox, g.inv_cell, g.nx), cy = cell_coord(qy, g.oy, g.inv_cell, g.ny), cz = cell_coord(qz, g.oz, g.inv_cell, g.nz);
  float bd[3] = {INFINITY, INFINITY, INFINITY};
  int bi[3] = {-1, -1, -1};
  const int rmax = max(g.nx, max(g.ny, g.nz));
  for (int r = 0; r <= rmax; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int z = cz + dz;
      if (z < 0 || z >= g.nz) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= g.ny) continue;
        const int row = (z * g.ny + y) * g.nx;
        if (abs(dz) == r || abs(dy) == r) {   // full x-run of the shell: one contiguous range
          const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
          if (x0 <= x1) scan_range(sorted, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, bd, bi);
        } else {                              // interior row: only the two end caps
          const int xa = cx - r, xb = cx + r;
          if (xa >= 0) scan_range(sorted, cell_start[row + xa], cell_start[row + xa + 1], qx, qy, qz, bd, bi);
          if (xb < g.nx) scan_range(sorted, cell_start[row + xb], cell_start[row + xb + 1], qx, qy, qz, bd, bi);
        }
      }
    }
    // lower bound on the distance to anything not yet visited
    float m = INFINITY;
    if (cx + r + 1 <= g.nx - 1) m = fminf(m, (g.ox + (float)(cx + r + 1) * g.cell) - qx);
    if (cx - r - 1 >= 0) m = fminf(m, qx - (g.ox + (float)(cx - r) * g.cell));
    if (cy + r + 1 <= g.ny - 1) m = fminf(m, (g.oy + (float)(cy + r + 1) * g.cell) - qy);
    if (cy - r - 1 >= 0) m = fminf(m, qy - (g.oy + (float)(cy - r) * g.cell));
    if (cz + r + 1 <= g.nz - 1) m = fminf(m, (g.oz + (float)(cz + r + 1) * g.cell) - qz);
    if (cz - r - 1 >= 0) m = fminf(m, qz - (g.oz + (float)(cz - r) * g.cell));
    if (!(m < INFINITY)) break;                       // every cell visited
    const float ms = fmaxf(m, 0.0f) * 0.9999f - 1e-6f * g.cell;  // conservative against cell-assignment rounding
    const float ms2 = ms > 0.0f ? ms * ms : 0.0f;
    if (bd[2] < ms2) break;                           // 3rd best is closer than anything unvisited
    if (ms2 >= thr) break;                            // nothing unvisited can pass the gate
  }
  idx[3 * i + 0] = bi[0]; idx[3 * i + 1] = bi[1]; idx[3 * i + 2] = bi[2];
  d2[3 * i + 0] = bd[0]; d2[3 * i + 1] = bd[1]; d2[3 * i + 2] = bd[2];
  valid[i] = (bi[0] >= 0 && bd[0] < thr && bi[1] >= 0 && bd[1] < thr && bi[2] >= 0 && bd[2] < thr) ? 1 : 0;
}

}  // namespace lvf

using namespace lvf;

extern "C" {

int lvf_map_create(lvf_ctx* ctx, const float* map_xyz, int M, int stride_floats, float max_radius2, lvf_map** out) {
  LVF_REQUIRE(ctx && out, "lvf_map_create: null ctx/out");
  LVF_REQUIRE(M >= 0 && (M == 0 || map_xyz) && stride_floats >= 3, "lvf_map_create: bad cloud (M=%d stride=%d)", M, stride_floats);
  LVF_REQUIRE(max_radius2 > 0.0f && std::isfinite(max_radius2), "lvf_map_create: max_radius2 must be finite > 0");
  LVF_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  auto* m = new lvf_map();
  m->ctx = ctx; m->M = M;
  auto fail = [&](int rc) { delete m; return rc; };
  int rc;
  if (M == 0) {
    if ((rc = m->cell_start.alloc(2)) != LVF_OK) return fail(rc);
    LVF_HIP(hipMemsetAsync(m->cell_start.p, 0, 2 * sizeof(int), s));
    m->cell = std::sqrt(max_radius2); m->inv_cell = 1.0f / m->cell;
    *out = m;
    return LVF_OK;
  }
  DevBuf<float> src; DevBuf<unsigned> bounds; DevBuf<int> cell_of, counts, cursor, bsums, total;
  if ((rc = src.upload(map_xyz, (size_t)M * stride_floats, s)) != LVF_OK) return fail(rc);
  if ((rc = m->raw.alloc(M)) != LVF_OK || (rc = m->sorted.alloc(M)) != LVF_OK || (rc = bounds.alloc(6)) != LVF_OK) return fail(rc);
  const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  LVF_HIP(hipMemcpyAsync(bounds.p, init, sizeof(init), hipMemcpyHostToDevice, s));
  const int gridM = (M + kB - 1) / kB;
  hipLaunchKernelGGL(k_pack_bounds, dim3(gridM), dim3(kB), 0, s, M, src.p, stride_floats, m->raw.p, bounds.p);
  unsigned hb[6];
  LVF_HIP(hipMemcpyAsync(hb, bounds.p, sizeof(hb), hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  float lo[3], hi[3];
  for (int k = 0; k < 3; ++k) { lo[k] = ord2f(hb[k]); hi[k] = ord2f(hb[3 + k]); }
  for (int k = 0; k < 3; ++k)
    if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) { set_error("lvf_map_create: non-finite map coordinates"); return fail(LVF_ERR_INVALID); }
  // Cell edge: start at gate radius / 2 and HALVE while the mean occupancy of non-empty cells stays above
  // kTargetOcc (lidar clouds are surfaces: a coarse cell holds hundreds of points and every query would scan
  // thousands of candidates; ~4-8 points per occupied cell keeps the first shells at ~100 candidates).
  const double kMaxCells = 4.0 * 1024 * 1024;
  const double kTargetOcc = 6.0;
  auto dims_for = [&](float c, double& nx, double& ny, double& nz) {
    nx = std::floor((hi[0] - lo[0]) / c) + 1; ny = std::floor((hi[1] - lo[1]) / c) + 1; nz = std::floor((hi[2] - lo[2]) / c) + 1;
    return nx * ny * nz;
  };
  float cell = std::sqrt(max_radius2) * 0.5f;
  double dnx, dny, dnz;
  while (dims_for(cell, dnx, dny, dnz) > kMaxCells) cell *= 1.25f;
  if ((rc = cell_of.alloc(M)) != LVF_OK || (rc = total.alloc(1)) != LVF_OK) return fail(rc);
  GridP g{};
  int ncells = 0;
  for (int trial = 0; trial < 8; ++trial) {
    dims_for(cell, dnx, dny, dnz);
    m->nx = (int)dnx; m->ny = (int)dny; m->nz = (int)dnz;
    m->cell = cell; m->inv_cell = 1.0f / cell; m->ox = lo[0]; m->oy = lo[1]; m->oz = lo[2];
    ncells = m->nx * m->ny * m->nz;
    g = GridP{m->ox, m->oy, m->oz, m->cell, m->inv_cell, m->nx, m->ny, m->nz};
    if ((rc = counts.alloc(ncells)) != LVF_OK) return fail(rc);
    LVF_HIP(hipMemsetAsync(counts.p, 0, (size_t)ncells * sizeof(int), s));
    LVF_HIP(hipMemsetAsync(total.p, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_cell_count, dim3(gridM), dim3(kB), 0, s, M, m->raw.p, g, cell_of.p, counts.p);
    hipLaunchKernelGGL(k_count_nonempty, dim3((ncells + kB - 1) / kB), dim3(kB), 0, s, ncells, counts.p, total.p);
    int nonempty = 0;
    LVF_HIP(hipMemcpyAsync(&nonempty, total.p, sizeof(int), hipMemcpyDeviceToHost, s));
    LVF_HIP(hipStreamSynchronize(s));
    const double occ = nonempty > 0 ? (double)M / nonempty : 0.0;
    double tnx, tny, tnz;
    if (occ <= kTargetOcc || dims_for(cell * 0.5f, tnx, tny, tnz) > kMaxCells) break;
    cell *= 0.5f;
  }
  const int nb = (ncells + kScanChunk - 1) / kScanChunk;
  if ((rc = cursor.alloc(ncells)) != LVF_OK || (rc = bsums.alloc(nb)) != LVF_OK || (rc = m->cell_start.alloc((size_t)ncells + 1)) != LVF_OK)
    return fail(rc);
  LVF_HIP(hipMemsetAsync(cursor.p, 0, (size_t)ncells * sizeof(int), s));
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(kScanT), 0, s, ncells, counts.p, bsums.p);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanT), 0, s, nb, bsums.p, total.p);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kScanT), 0, s, ncells, counts.p, bsums.p, m->cell_start.p);
  LVF_HIP(hipMemcpyAsync(m->cell_start.p + ncells, total.p, sizeof(int), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_cell_scatter, dim3(gridM), dim3(kB), 0, s, M, m->raw.p, cell_of.p, m->cell_start.p, cursor.p, m->sorted.p);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipStreamSynchronize(s));   // temporaries are freed on return
  *out = m;
  return LVF_OK;
}

int lvf_map_destroy(lvf_map* m) { delete m; return LVF_OK; }

int lvf_scan_create(lvf_ctx* ctx, const float* scan_xyz, int Q, int stride_floats, lvf_scan** out) {
  LVF_REQUIRE(ctx && out, "lvf_scan_create: null ctx/out");
  LVF_REQUIRE(Q >= 0 && (Q == 0 || scan_xyz) && stride_floats >= 3, "lvf_scan_create: bad cloud (Q=%d stride=%d)", Q, stride_floats);
  LVF_HIP(hipSetDevice(ctx->device));
  auto* sc = new lvf_scan();
  sc->ctx = ctx; sc->Q = Q;
  int rc = LVF_OK;
  DevBuf<float> src;
  if (Q > 0) {
    if ((rc = src.upload(scan_xyz, (size_t)Q * stride_floats, ctx->stream)) != LVF_OK || (rc = sc->pts.alloc(Q)) != LVF_OK ||
        (rc = sc->idx.alloc((size_t)3 * Q)) != LVF_OK || (rc = sc->d2.alloc((size_t)3 * Q)) != LVF_OK || (rc = sc->valid.alloc(Q)) != LVF_OK) {
      delete sc; return rc;
    }
    hipLaunchKernelGGL(k_pack, dim3((Q + kB - 1) / kB), dim3(kB), 0, ctx->stream, Q, src.p, stride_floats, sc->pts.p);
    LVF_HIP(hipGetLastError());
    LVF_HIP(hipStreamSynchronize(ctx->stream));
  }
  *out = sc;
  return LVF_OK;
}

int lvf_scan_destroy(lvf_scan* s) { delete s; return LVF_OK; }

int lvf_knn3(lvf_map* m, lvf_scan* sc, const double* pose, float thr) {
  LVF_REQUIRE(m && sc && pose, "lvf_knn3: null argument");
  LVF_REQUIRE(m->ctx == sc->ctx, "lvf_knn3: map and scan belong to different contexts");
  LVF_REQUIRE(!(thr != thr) && thr > 0.0f, "lvf_knn3: thr must be > 0");
  LVF_HIP(hipSetDevice(m->ctx->device));
  if (sc->Q > 0) {
    TfArg tf;
    for (int k = 0; k < 7; ++k) tf.v[k] = (float)pose[k];   // Sophus SE3d::cast<float>()  association.cpp:287
    const GridP g{m->ox, m->oy, m->oz, m->cell, m->inv_cell, m->nx, m->ny, m->nz};
    hipLaunchKernelGGL(k_knn3, dim3((sc->Q + kB - 1) / kB), dim3(kB), 0, m->ctx->stream, sc->Q, sc->pts.p, tf, m->sorted.p,
                       m->cell_start.p, g, thr, sc->idx.p, sc->d2.p, sc->valid.p);
    LVF_HIP(hipGetLastError());
  }
  sc->searched = true;
  return LVF_OK;
}

int lvf_scan_download(lvf_scan* sc, int32_t* idx3, float* d2_3, uint8_t* valid) {
  LVF_REQUIRE(sc, "lvf_scan_download: null scan");
  if (!sc->searched) { set_error("lvf_scan_download: no lvf_knn3 result yet"); return LVF_ERR_STATE; }
  hipStream_t s = sc->ctx->stream;
  if (sc->Q > 0) {
    if (idx3) LVF_HIP(hipMemcpyAsync(idx3, sc->idx.p, (size_t)3 * sc->Q * sizeof(int), hipMemcpyDeviceToHost, s));
    if (d2_3) LVF_HIP(hipMemcpyAsync(d2_3, sc->d2.p, (size_t)3 * sc->Q * sizeof(float), hipMemcpyDeviceToHost, s));
    if (valid) LVF_HIP(hipMemcpyAsync(valid, sc->valid.p, (size_t)sc->Q, hipMemcpyDeviceToHost, s));
  }
  LVF_HIP(hipStreamSynchronize(s));
  return LVF_OK;
}

}  // extern "C"
