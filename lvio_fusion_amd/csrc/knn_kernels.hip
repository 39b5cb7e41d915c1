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
#include <cstdlib>
#include "lvf_internal.hpp"
#include "scan_match_dev.hpp"
#include "lidar_eval.hpp"

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

// (GridP / LevelP / LevelsP: scan_match_dev.hpp)

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f; memcpy(&f, &u, 4); return f;
}

// pack strided xyz records into float4 and reduce the bounding box (ordered-uint atomics).  The grid is capped and strides over
// the cloud, and every workgroup reduces through LDS first: atomics on ONE address serialise at ~65 ns each in L2 (measured: one
// atomic per wave = 5.3 k per address cost 365 us on a 340 k-point cloud).
constexpr int kBoundsMaxBlocks = 256;
__device__ __forceinline__ void pack_bounds_body(const int bx, const int nbx, int M, const float* __restrict__ src, int stride, float4* __restrict__ dst,
                                                 unsigned* __restrict__ bounds /* min xyz, max xyz */) {
  float x = INFINITY, y = INFINITY, z = INFINITY, X = -INFINITY, Y = -INFINITY, Z = -INFINITY;
  for (int i = bx * kB + threadIdx.x; i < M; i += nbx * kB) {
    const float* s = src + (size_t)i * stride;
    const float a = s[0], b = s[1], c = s[2];
    dst[i] = make_float4(a, b, c, 0.0f);
    x = fminf(x, a); y = fminf(y, b); z = fminf(z, c); X = fmaxf(X, a); Y = fmaxf(Y, b); Z = fmaxf(Z, c);
  }
  for (int o = 32; o > 0; o >>= 1) {
    x = fminf(x, __shfl_down(x, o)); y = fminf(y, __shfl_down(y, o)); z = fminf(z, __shfl_down(z, o));
    X = fmaxf(X, __shfl_down(X, o)); Y = fmaxf(Y, __shfl_down(Y, o)); Z = fmaxf(Z, __shfl_down(Z, o));
  }
  __shared__ float red[kB / 64][6];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = x; red[w][1] = y; red[w][2] = z; red[w][3] = X; red[w][4] = Y; red[w][5] = Z; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int k = 1; k < kB / 64; ++k) v = threadIdx.x < 3 ? fminf(v, red[k][threadIdx.x]) : fmaxf(v, red[k][threadIdx.x]);
    if (threadIdx.x < 3) atomicMin(bounds + threadIdx.x, f2ord(v)); else atomicMax(bounds + threadIdx.x, f2ord(v));
  }
}
// table forms (blockIdx.y = map) of the index build's launches: lvf_map_create_batch enqueues ONE launch per step for all the maps of a
// round instead of one per map (16 loop-closure maps x 3 grid levels x 6 steps were ~290 launches of a few microseconds each)
struct PackJob { int M, stride, nbx, pad; GP<const float> src; GP<float4> dst; GP<unsigned> bounds; };
__global__ __launch_bounds__(kB) void k_pack_bounds_t(const PackJob* __restrict__ jobs) {
  const PackJob J = jobs[blockIdx.y];
  if ((int)blockIdx.x >= J.nbx) return;
  pack_bounds_body(blockIdx.x, J.nbx, J.M, J.src, J.stride, J.dst, J.bounds);
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_cell, int n) {
  int c = (int)floorf((v - o) * inv_cell);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__device__ __forceinline__ void cell_count_body(const int bx, int M, const float4* __restrict__ pts, const GridP& g, int* __restrict__ cell_of,
                                                int* __restrict__ counts) {
  const int i = bx * kB + threadIdx.x;
  int c = -1;
  if (i < M) {
    const float4 p = pts[i];
    c = (cell_coord(p.z, g.oz, g.inv_cell, g.nz) * g.ny + cell_coord(p.y, g.oy, g.inv_cell, g.ny)) * g.nx + cell_coord(p.x, g.ox, g.inv_cell, g.nx);
    cell_of[i] = c;
  }
  int start, len;
  const bool head = cell_runs(c, start, len);
  if (head && c >= 0) atomicAdd(counts + c, len);
}

// (sum over cells of count^2) / M is the population of the cell a random map point lives in, i.e. the candidates a query in a
// typical (point-weighted) place has to scan per cell; it is produced by the scan passes below.

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
// block_sq (optional): per-workgroup sum of squares of the inputs — the grid build's occupancy statistic rides along the pass
// that reads every count anyway (it used to be its own kernel with a same-address atomic per workgroup)
__device__ __forceinline__ void scan_reduce_body(const int bx, int n, const int* __restrict__ in, int* __restrict__ block_sums,
                                                 unsigned long long* __restrict__ block_sq) {
  __shared__ int s_warp[kScanT / 64];
  __shared__ unsigned long long s_sq[kScanT / 64];
  const int base = bx * kScanChunk + threadIdx.x * kScanE;
  int v = 0;
  unsigned long long sq = 0;
  for (int e = 0; e < kScanE; ++e) if (base + e < n) { const int c = in[base + e]; v += c; sq += (unsigned long long)c * (unsigned long long)c; }
  int total;
  (void)block_exclusive_scan(v, s_warp, total);
  if (threadIdx.x == 0) block_sums[bx] = total;
  if (block_sq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o);
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int k = 0; k < kScanT / 64; ++k) t += s_sq[k]; block_sq[bx] = t; }
  }
}
__global__ __launch_bounds__(kScanT) void k_scan_reduce(int n, const int* __restrict__ in, int* __restrict__ block_sums,
                                                        unsigned long long* __restrict__ block_sq) {
  scan_reduce_body(blockIdx.x, n, in, block_sums, block_sq);
}
__device__ __forceinline__ void scan_sums_body(int nb, int* __restrict__ block_sums, int* __restrict__ grand_total,
                                               const unsigned long long* __restrict__ block_sq, unsigned long long* __restrict__ sq_total) {
  __shared__ int s_warp[kScanT / 64];
  if (block_sq) {                                  // total of the per-workgroup square sums
    __shared__ unsigned long long s_sq[kScanT / 64];
    unsigned long long sq = 0;
    for (int i = threadIdx.x; i < nb; i += kScanT) sq += block_sq[i];
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o);
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int k = 0; k < kScanT / 64; ++k) t += s_sq[k]; *sq_total = t; }
    __syncthreads();
  }
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
__global__ __launch_bounds__(kScanT) void k_scan_sums(int nb, int* __restrict__ block_sums, int* __restrict__ grand_total,
                                                      const unsigned long long* __restrict__ block_sq, unsigned long long* __restrict__ sq_total) {
  scan_sums_body(nb, block_sums, grand_total, block_sq, sq_total);
}
// clear_in: zero the input after reading it (the grid build re-uses the count array as the scatter cursor)
__device__ __forceinline__ void scan_apply_body(const int bx, int n, int* in, const int* __restrict__ block_sums, int* __restrict__ out, int clear_in) {
  __shared__ int s_warp[kScanT / 64];
  const int base = bx * kScanChunk + threadIdx.x * kScanE;
  int e_v[kScanE];
  int v = 0;
  for (int e = 0; e < kScanE; ++e) { e_v[e] = (base + e < n) ? in[base + e] : 0; v += e_v[e]; }
  int total;
  int run = block_sums[bx] + block_exclusive_scan(v, s_warp, total);
  for (int e = 0; e < kScanE; ++e) { if (base + e < n) { out[base + e] = run; if (clear_in) in[base + e] = 0; } run += e_v[e]; }
}
__global__ __launch_bounds__(kScanT) void k_scan_apply(int n, int* in, const int* __restrict__ block_sums,
                                                       int* __restrict__ out, int clear_in) {
  scan_apply_body(blockIdx.x, n, in, block_sums, out, clear_in);
}

__device__ __forceinline__ void cell_scatter_body(const int bx, int M, const float4* __restrict__ pts, const int* __restrict__ cell_of,
                                                  const int* __restrict__ cell_start, int* __restrict__ cursor,
                                                  float4* __restrict__ sorted) {
  const int i = bx * kB + threadIdx.x;
  const int c = (i < M) ? cell_of[i] : -1;
  int start, len;
  const bool head = cell_runs(c, start, len);
  int base = 0;
  if (head && c >= 0) base = atomicAdd(cursor + c, len);        // the run reserves `len` consecutive slots of its cell
  base = __shfl(base, start);
  if (c < 0) return;
  float4 p = pts[i];
  p.w = __int_as_float(i);
  sorted[cell_start[c] + base + ((int)(threadIdx.x & 63) - start)] = p;
}
// one grid level of one map (table entry of the batched index build)
struct LevelJob {
  int M, ncells, gridM, nb;
  GridP g;
  GP<const float4> raw; GP<int> cell_of; GP<int> counts; GP<int> bsums; GP<unsigned long long> bsq; GP<int> cell_start; GP<float4> sorted; GP<unsigned long long> sumsq;
};
__global__ __launch_bounds__(kB) void k_level_zero_t(const LevelJob* __restrict__ jobs) {            // counts = 0
  const LevelJob& J = jobs[blockIdx.y];
  for (int i = blockIdx.x * kB + threadIdx.x; i < J.ncells; i += gridDim.x * kB) J.counts[i] = 0;
}
__global__ __launch_bounds__(kB) void k_cell_count_t(const LevelJob* __restrict__ jobs) {
  const LevelJob& J = jobs[blockIdx.y];
  if ((int)blockIdx.x >= J.gridM) return;
  cell_count_body(blockIdx.x, J.M, J.raw, J.g, J.cell_of, J.counts);
}
__global__ __launch_bounds__(kScanT) void k_scan_reduce_t(const LevelJob* __restrict__ jobs) {
  const LevelJob& J = jobs[blockIdx.y];
  if ((int)blockIdx.x >= J.nb) return;
  scan_reduce_body(blockIdx.x, J.ncells, J.counts, J.bsums, J.bsq);
}
__global__ __launch_bounds__(kScanT) void k_scan_sums_t(const LevelJob* __restrict__ jobs) {
  const LevelJob& J = jobs[blockIdx.y];
  scan_sums_body(J.nb, J.bsums, J.cell_start + J.ncells, J.bsq, J.sumsq);
}
__global__ __launch_bounds__(kScanT) void k_scan_apply_t(const LevelJob* __restrict__ jobs) {
  const LevelJob& J = jobs[blockIdx.y];
  if ((int)blockIdx.x >= J.nb) return;
  scan_apply_body(blockIdx.x, J.ncells, J.counts, J.bsums, J.cell_start, 1);
}
__global__ __launch_bounds__(kB) void k_cell_scatter_t(const LevelJob* __restrict__ jobs) {
  const LevelJob& J = jobs[blockIdx.y];
  if ((int)blockIdx.x >= J.gridM) return;
  cell_scatter_body(blockIdx.x, J.M, J.raw, J.cell_of, J.cell_start, J.counts, J.sorted);
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
  if (!lex_less(d, i, bd[2], bi[2]) || i == bi[0] || i == bi[1]) return;   // worse than 3rd, or already held
  if (lex_less(d, i, bd[1], bi[1])) {
    bd[2] = bd[1]; bi[2] = bi[1];
    if (lex_less(d, i, bd[0], bi[0])) { bd[1] = bd[0]; bi[1] = bi[0]; bd[0] = d; bi[0] = i; }
    else { bd[1] = d; bi[1] = i; }
  } else { bd[2] = d; bi[2] = i; }
}

struct KnnStats { int candidates, lookups, level, shells, t_start, t_end; };   // t_*: wall_clock64() (100 MHz) of the query's wave, low 31 bits

constexpr int kGroup = 8;   // lanes cooperating on one query (8 queries per wave64)

// Candidates of one contiguous cell range.  Loads are issued four at a time so that four 16-B gathers are in flight
// per lane (the loop is otherwise a chain of dependent L2/MALL round trips).
__device__ __forceinline__ void scan_range(const float4* __restrict__ sorted, int lo, int hi, float qx, float qy, float qz,
                                           float bd[3], int bi[3], KnnStats* st = nullptr) {
  for (int j = lo; j < hi; j += 4) {
    const int last = hi - 1;
    const float4 m0 = sorted[j], m1 = sorted[min(j + 1, last)], m2 = sorted[min(j + 2, last)], m3 = sorted[min(j + 3, last)];
    const float4 mm[4] = {m0, m1, m2, m3};
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // a clamped duplicate of the last point is rejected by best3_push (same index)
      const float dx = sub_rn(qx, mm[u].x), dy = sub_rn(qy, mm[u].y), dz = sub_rn(qz, mm[u].z);
      const float d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
      best3_push(d, __float_as_int(mm[u].w), bd, bi);
    }
  }
}

// The whole group scans ONE long range: lane g takes candidates lo+g, lo+g+8, ... (coalesced 128-B rows), four
// strides in flight per lane.  Used for ranges a single lane would take hundreds of dependent round trips to walk.
__device__ __forceinline__ void scan_range_group(const float4* __restrict__ sorted, int lo, int hi, int g_lane, float qx,
                                                 float qy, float qz, float bd[3], int bi[3]) {
  for (int j = lo + g_lane; j < hi; j += 4 * kGroup) {
    const int last = hi - 1;
    const float4 mm[4] = {sorted[j], sorted[min(j + kGroup, last)], sorted[min(j + 2 * kGroup, last)], sorted[min(j + 3 * kGroup, last)]};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = sub_rn(qx, mm[u].x), dy = sub_rn(qy, mm[u].y), dz = sub_rn(qz, mm[u].z);
      const float d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
      best3_push(d, __float_as_int(mm[u].w), bd, bi);
    }
  }
}

constexpr int kLongRange = 12;   // candidates; longer ranges are deferred to the cooperative scan

struct TfArg { float v[7]; };


// One cubic shell of cells around (cx,cy,cz); the (dz,dy) rows of the shell are dealt round-robin to the kGroup lanes
// of the query's group (lane `g` takes rows g, g+kGroup, ...).  Returns the lower bound on the distance to any point
// of THIS level's grid outside shells 0..r (INFINITY when the shells already cover the whole grid).
__device__ __forceinline__ float scan_shell(const LevelP& L, int cx, int cy, int cz, int r, int g_lane, float qx, float qy,
                                            float qz, float bd[3], int bi[3], KnnStats* st) {
  const GridP& g = L.g;
  const int side = 2 * r + 1;
  // Pruning radius: the group's 3rd-best d2 as of the previous shell (all lanes hold the merged list).  A cell whose
  // closest corner is farther than that cannot contribute (ties included: only strictly farther cells are skipped);
  // gaps are shrunk by 1e-4 relative + 1e-6 cell to stay conservative against float cell-assignment rounding.
  const float prune2 = bd[2];
  const float slack = 1e-6f * g.cell;
  const int lane_in_wave = threadIdx.x & 63;
  const int group_shift = lane_in_wave & ~(kGroup - 1);
  // shell 1 has 8 perimeter rows (one range each) + the centre row (two end caps): packed into ONE lock-step pass — the eight
  // lanes take the perimeter rows and lanes 0 / 1 carry the centre row's caps in their second range slot — instead of a second
  // pass in which a single lane works (every pass is a chain of dependent look-ups)
  const bool packed = (r == 1) && (kGroup == 8);
  for (int it = 0; it * kGroup < (packed ? kGroup : side * side); ++it) {       // lock-step over the group: lane g owns row it*kGroup+g
    const int row_id = packed ? (g_lane < 4 ? g_lane : g_lane + 1) : it * kGroup + g_lane;
    int la = 0, ha = 0, lb = 0, hb = 0;                    // up to two candidate ranges for this lane's row
    if (row_id < side * side) {
      const int dz = row_id / side - r, dy = row_id % side - r;
      const int z = cz + dz, y = cy + dy;
      if (z >= 0 && z < g.nz && y >= 0 && y < g.ny) {
        float gy = 0.0f, gz = 0.0f;
        if (y > cy) gy = (g.oy + (float)y * g.cell) - qy; else if (y < cy) gy = qy - (g.oy + (float)(y + 1) * g.cell);
        if (z > cz) gz = (g.oz + (float)z * g.cell) - qz; else if (z < cz) gz = qz - (g.oz + (float)(z + 1) * g.cell);
        gy = fmaxf(gy * 0.9999f - slack, 0.0f); gz = fmaxf(gz * 0.9999f - slack, 0.0f);
        const float base2 = gy * gy + gz * gz;
        if (!(base2 > prune2)) {
          int xlo = 0, xhi = g.nx - 1;
          if (prune2 < INFINITY) {
            const float dxm = sqrtf(prune2 - base2) * 1.0001f + slack;
            xlo = max(xlo, (int)floorf((qx - dxm - g.ox) * g.inv_cell - 1e-3f));
            xhi = min(xhi, (int)floorf((qx + dxm - g.ox) * g.inv_cell + 1e-3f));
          }
          const int row = (z * g.ny + y) * g.nx;
          if (abs(dz) == r || abs(dy) == r) {   // full x-run of the shell: one contiguous range
            const int x0 = max(cx - r, xlo), x1 = min(cx + r, xhi);
            if (x0 <= x1) { la = L.cell_start[row + x0]; ha = L.cell_start[row + x1 + 1]; }
          } else {                              // interior row: only the two end caps
            const int xa = cx - r, xb = cx + r;
            if (xa >= xlo && xa <= xhi) { la = L.cell_start[row + xa]; ha = L.cell_start[row + xa + 1]; }
            if (xb <= xhi && xb >= xlo) { lb = L.cell_start[row + xb]; hb = L.cell_start[row + xb + 1]; }
          }
        }
      }
    }
    if (packed && g_lane < 2) {                            // centre row (dy = dz = 0: base2 = 0), cap cx - 1 or cx + 1
      int xlo = 0, xhi = g.nx - 1;
      if (prune2 < INFINITY) {
        const float dxm = sqrtf(prune2) * 1.0001f + slack;
        xlo = max(xlo, (int)floorf((qx - dxm - g.ox) * g.inv_cell - 1e-3f));
        xhi = min(xhi, (int)floorf((qx + dxm - g.ox) * g.inv_cell + 1e-3f));
      }
      const int xc = g_lane == 0 ? cx - 1 : cx + 1;
      if (xc >= xlo && xc <= xhi) { const int row = (cz * g.ny + cy) * g.nx; lb = L.cell_start[row + xc]; hb = L.cell_start[row + xc + 1]; }
    }
    if (st) { st->lookups += (ha > la) + (hb > lb); st->candidates += (ha - la) + (hb - lb); }
    const bool long_a = (ha - la) > kLongRange, long_b = (hb - lb) > kLongRange;
    if (!long_a && ha > la) scan_range(L.sorted, la, ha, qx, qy, qz, bd, bi);
    if (!long_b && hb > lb) scan_range(L.sorted, lb, hb, qx, qy, qz, bd, bi);
    // long ranges: every lane of the group helps, one owner at a time
    unsigned owners = (unsigned)((__ballot(long_a || long_b) >> group_shift) & ((1u << kGroup) - 1u));
    while (owners) {
      const int k = __ffs(owners) - 1;
      owners &= owners - 1;
      const int src = group_shift + k;
      const int rla = __shfl(la, src), rha = __shfl(ha, src), rlb = __shfl(lb, src), rhb = __shfl(hb, src);
      if (rha - rla > kLongRange) scan_range_group(L.sorted, rla, rha, g_lane, qx, qy, qz, bd, bi);
      if (rhb - rlb > kLongRange) scan_range_group(L.sorted, rlb, rhb, g_lane, qx, qy, qz, bd, bi);
    }
  }
  float m = INFINITY;
  if (cx + r + 1 <= g.nx - 1) m = fminf(m, (g.ox + (float)(cx + r + 1) * g.cell) - qx);
  if (cx - r - 1 >= 0) m = fminf(m, qx - (g.ox + (float)(cx - r) * g.cell));
  if (cy + r + 1 <= g.ny - 1) m = fminf(m, (g.oy + (float)(cy + r + 1) * g.cell) - qy);
  if (cy - r - 1 >= 0) m = fminf(m, qy - (g.oy + (float)(cy - r) * g.cell));
  if (cz + r + 1 <= g.nz - 1) m = fminf(m, (g.oz + (float)(cz + r + 1) * g.cell) - qz);
  if (cz - r - 1 >= 0) m = fminf(m, qz - (g.oz + (float)(cz - r) * g.cell));
  return m;
}

// butterfly all-reduce of the per-lane best-3 lists inside each aligned kGroup-lane group; the (d2, index) order with
// same-index rejection makes the merge commutative/associative, so every lane ends with the identical list.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ void merge_step(float bd[3], int bi[3]) {
  float od[3]; int oi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { od[k] = __int_as_float(dpp_i<CTRL>(__float_as_int(bd[k]))); oi[k] = dpp_i<CTRL>(bi[k]); }
#pragma unroll
  for (int k = 0; k < 3; ++k) if (oi[k] >= 0) best3_push(od[k], oi[k], bd, bi);
}
__device__ __forceinline__ void group_merge_best3(float bd[3], int bi[3]) {
  if (kGroup == 8) {
    // partner lanes through DPP (VALU latency) instead of ds_bpermute round trips: lane ^ 1, lane ^ 2 (quad_perm), then the mirror
    // image inside the 8-lane half row (i <-> 7 - i), which pairs the two quads just as well as lane ^ 4
    merge_step<0xB1>(bd, bi);
    merge_step<0x4E>(bd, bi);
    merge_step<0x141>(bd, bi);
    return;
  }
#pragma unroll
  for (int mask = 1; mask < kGroup; mask <<= 1) {
    float od[3]; int oi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { od[k] = __shfl_xor(bd[k], mask); oi[k] = __shfl_xor(bi[k], mask); }
#pragma unroll
    for (int k = 0; k < 3; ++k) if (oi[k] >= 0) best3_push(od[k], oi[k], bd, bi);
  }
}

// BUILD: the query's writer lane also builds the point-to-plane correspondence of an accepted query (association.cpp:303-314: the scan point
// and its first neighbour as doubles, the unit normal of the plane through the three neighbours) into corr = P | PA | N (SoA [3][Q] each) — what
// k_icp_build did in a launch of its own: one launch and the round trip of the indices less per sub-problem.
template <bool STATS, bool BUILD = false>
__device__ __forceinline__ void knn3_body(const int bx, int Q, const float4* __restrict__ scan, const float* tfv, const LevelsP& L,
                                          float thr, int* __restrict__ idx, float* __restrict__ d2,
                                          uint8_t* __restrict__ valid, KnnStats* __restrict__ stats,
                                          const float4* __restrict__ map_raw = nullptr, double* __restrict__ corr = nullptr) {
  // (taking the blocks of queries from both ends alternately, or last to first, so that the expensive end of a scan does not start last:
  // measured 15 % / 40 % SLOWER — neighbouring workgroups share map cells in L2, and the dense near field is better met by a chip that is full)
  const int t = bx * kB + threadIdx.x;
  const int g_lane = t & (kGroup - 1);
  const int i = min(t / kGroup, Q - 1);          // surplus groups of the last block shadow the last query (no divergent exit
  const bool writer = (t / kGroup) < Q && g_lane == 0;   // before the shuffles)
  const Tf32 T = make_tf32(tfv);
  const float4 p = scan[i];
  const float qx = tf_row(T.a + 0, p.x, p.y, p.z, p.x, T.t[0]);
  const float qy = tf_row(T.a + 3, p.x, p.y, p.z, p.y, T.t[1]);
  const float qz = tf_row(T.a + 6, p.x, p.y, p.z, p.z, T.t[2]);
  float bd[3] = {INFINITY, INFINITY, INFINITY};
  int bi[3] = {-1, -1, -1};
  KnnStats st{0, 0, 0, 0, 0, 0};
  if (STATS) st.t_start = (int)((long long)wall_clock64() & 0x7fffffff);
  KnnStats* stp = STATS ? &st : nullptr;
  // fine -> coarse: shells 0..1 of the finest grid, shells 0..2 of every coarser one; dense neighbourhoods finish in the
  // finest grid, sparse ones escalate to a grid whose cells are 2x larger instead of walking dozens of empty fine shells.
  // Re-visited points are rejected by best3_push (same index), so levels can overlap freely.  Measured at configs[2]
  // (ground / surf gate): pyramid ratio 4 with 2 shells per level 626 / 657 Mpairs/s; ratio 2: 756 / 849; ratio 2 with one shell pair
  // per level 706 / 999; ratio 2 with this schedule 755 / 952.  (Choosing the starting level from the occupancy of the query's
  // own cell at every level — one more round trip — was slower: 463 / 748.)  Then the finest level itself: the point-weighted
  // occupancy target kTargetOcc 12 -> 24 -> 48 -> 96..192 gave 755 -> 913 -> 927 -> 1100 Mpairs/s (384: 515, the grid becomes too
  // coarse), and handing ranges longer than kLongRange 24 -> 12 candidates to the cooperative scan another 2-5 %: dependent
  // cell look-ups cost more than streaming a few dozen extra candidates.
  for (int lv = 0; lv < L.n; ++lv) {
    const LevelP& lev = L.l[lv];
    const GridP& g = lev.g;
    const int cx = cell_coord(qx, g.ox, g.inv_cell, g.nx), cy = cell_coord(qy, g.oy, g.inv_cell, g.ny), cz = cell_coord(qz, g.oz, g.inv_cell, g.nz);
    const int rcap = (lv == L.n - 1) ? max(g.nx, max(g.ny, g.nz)) : (lv == 0 ? 1 : 2);
    bool done = false;
    for (int r = 0; r <= rcap; ++r) {
      const float m = scan_shell(lev, cx, cy, cz, r, g_lane, qx, qy, qz, bd, bi, stp);
      group_merge_best3(bd, bi);
      if (STATS) { st.level = lv; st.shells += 1; }
      if (!(m < INFINITY)) { done = true; break; }                   // every point of the map has been seen
      const float ms = fmaxf(m, 0.0f) * 0.9999f - 1e-6f * g.cell;   // conservative against cell-assignment rounding
      const float ms2 = ms > 0.0f ? ms * ms : 0.0f;
      if (bd[2] < ms2 || ms2 >= thr) { done = true; break; }         // 3rd best beats anything unvisited / beyond the gate
    }
    if (done) break;
  }
  if (STATS) {
#pragma unroll
    for (int mask = 1; mask < kGroup; mask <<= 1) { st.candidates += __shfl_xor(st.candidates, mask); st.lookups += __shfl_xor(st.lookups, mask); }
    st.t_end = (int)((long long)wall_clock64() & 0x7fffffff);
    if (writer) stats[i] = st;
  }
  if (writer) {
    idx[3 * i + 0] = bi[0]; idx[3 * i + 1] = bi[1]; idx[3 * i + 2] = bi[2];
    d2[3 * i + 0] = bd[0]; d2[3 * i + 1] = bd[1]; d2[3 * i + 2] = bd[2];
    const bool ok = bi[0] >= 0 && bd[0] < thr && bi[1] >= 0 && bd[1] < thr && bi[2] >= 0 && bd[2] < thr;
    valid[i] = ok ? 1 : 0;
    if (BUILD && ok) {
      const float4 a = map_raw[bi[0]], b = map_raw[bi[1]], c = map_raw[bi[2]];
      const double pa[3] = {(double)a.x, (double)a.y, (double)a.z}, pb[3] = {(double)b.x, (double)b.y, (double)b.z}, pc[3] = {(double)c.x, (double)c.y, (double)c.z};
      double n[3];
      plane_normal(pa, pb, pc, n);
      double* P = corr; double* PA = corr + (size_t)3 * Q; double* N = corr + (size_t)6 * Q;
      P[i] = (double)p.x; P[Q + i] = (double)p.y; P[2 * Q + i] = (double)p.z;
      PA[i] = pa[0]; PA[Q + i] = pa[1]; PA[2 * Q + i] = pa[2];
      N[i] = n[0]; N[Q + i] = n[1]; N[2 * Q + i] = n[2];
    }
  }
}
template <bool STATS>
__global__ __launch_bounds__(kB) void k_knn3(int Q, const float4* __restrict__ scan, const TfArg tfa, const LevelsP L,
                                             float thr, int* __restrict__ idx, float* __restrict__ d2,
                                             uint8_t* __restrict__ valid, KnnStats* __restrict__ stats) {
  knn3_body<STATS>(blockIdx.x, Q, scan, tfa.v, L, thr, idx, d2, valid, stats);
}
// association + correspondence build in one launch (lvf_icp_solve)
__global__ __launch_bounds__(kB) void k_knn3_build(int Q, const float4* __restrict__ scan, const TfArg tfa, const LevelsP L, float thr, int* __restrict__ idx,
                                                   float* __restrict__ d2, uint8_t* __restrict__ valid, const float4* __restrict__ map_raw, double* __restrict__ corr) {
  knn3_body<false, true>(blockIdx.x, Q, scan, tfa.v, L, thr, idx, d2, valid, nullptr, map_raw, corr);
}
// Many associations in one launch: blockIdx.y = candidate, the job (scan, map pyramid, gate, outputs) comes from a device table and the
// float transform from the candidate's device record — the pose the previous sub-problem left there never visits the host.
__global__ __launch_bounds__(kB) void k_knn3_b(const KnnJob* __restrict__ jobs, const SmDev* __restrict__ devs, int sub) {
  const KnnJob& J = jobs[2 * blockIdx.y + sub];
  if ((long long)blockIdx.x * (kB / kGroup) >= (long long)J.Q) return;      // (uniform per workgroup; also Q == 0: nothing to associate)
  const SmDev& D = devs[blockIdx.y];
  if (!D.has[sub]) return;
  knn3_body<false, true>(blockIdx.x, J.Q, J.scan, D.tf, J.L, J.thr, J.idx, J.d2, J.valid, nullptr, J.map_raw, J.corr);
}

// exclusive prefix sum of n ints on the context's stream; out has n + 1 entries (out[n] = grand total).  in != out.
int device_exclusive_scan_i32(lvf_ctx* ctx, const int* in, int n, int* out) {
  hipStream_t s = ctx->stream;
  if (n <= 0) { LVF_HIP(hipMemsetAsync(out, 0, sizeof(int), s)); return LVF_OK; }
  const int nb = (n + kScanChunk - 1) / kScanChunk;
  DevBuf<int> bsums;
  LVF_TRY(bsums.alloc(nb));
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(kScanT), 0, s, n, in, bsums.p, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanT), 0, s, nb, bsums.p, out + n, (const unsigned long long*)nullptr, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kScanT), 0, s, n, const_cast<int*>(in), bsums.p, out, 0);
  LVF_HIP(hipGetLastError());
  // (bsums goes back to the context's pool on return and is handed out again in stream order: nothing to wait for)
  return LVF_OK;
}

}  // namespace lvf

using namespace lvf;

static inline int knn_grid(int Q) { return (int)(((long long)Q * kGroup + kB - 1) / kB); }

namespace lvf {
LevelsP levels_of(const lvf_map* m) {
  LevelsP L;
  std::memset(&L, 0, sizeof(L));
  L.n = m->n_levels;
  for (int k = 0; k < m->n_levels; ++k) {
    const auto& lv = m->levels[k];
    L.l[k] = LevelP{lv.sorted.p, lv.cell_start.p, GridP{lv.ox, lv.oy, lv.oz, lv.cell, lv.inv_cell, lv.nx, lv.ny, lv.nz}};
  }
  return L;
}
int launch_knn3_build(lvf_map* m, lvf_scan* sc, const double* pose, float thr, double* corr) {
  if (sc->Q <= 0) return LVF_OK;
  TfArg tf;
  for (int k = 0; k < 7; ++k) tf.v[k] = (float)pose[k];   // Sophus SE3d::cast<float>()  association.cpp:287
  hipLaunchKernelGGL(k_knn3_build, dim3(knn_grid(sc->Q)), dim3(kB), 0, m->ctx->stream, sc->Q, sc->pts.p, tf, levels_of(m), thr, sc->idx.p, sc->d2.p, sc->valid.p,
                     m->raw.p, corr);
  LVF_HIP(hipGetLastError());
  sc->searched = true;
  return LVF_OK;
}
int launch_knn3_batch(hipStream_t q, const KnnJob* jobs, const SmDev* devs, int n, int sub, int max_Q) {
  if (n <= 0 || max_Q <= 0) return LVF_OK;
  hipLaunchKernelGGL(k_knn3_b, dim3(knn_grid(max_Q), n), dim3(kB), 0, q, jobs, devs, sub);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}
}  // namespace lvf

extern "C" {

// Enqueues the build of one grid level (counting sort of the map by cell) into lv; nothing is waited for.  *sumsq_dev receives
// sum(count^2) over the cells (point-weighted mean cell population = that / M); `t` holds the temporaries and must outlive the launches.
struct LevelTmp { DevBuf<int> cell_of, counts, bsums; DevBuf<unsigned long long> bsq; };
// allocations and geometry of one level -> the job record the kernels take (nothing is launched)
static int prepare_level(lvf_map* m, lvf_map::Level& lv, float cell, const float lo[3], const float hi[3], LevelTmp& t, unsigned long long* sumsq_dev, LevelJob* job) {
  const int M = m->M;
  lv.nx = (int)(std::floor((hi[0] - lo[0]) / cell) + 1); lv.ny = (int)(std::floor((hi[1] - lo[1]) / cell) + 1);
  lv.nz = (int)(std::floor((hi[2] - lo[2]) / cell) + 1);
  lv.cell = cell; lv.inv_cell = 1.0f / cell; lv.ox = lo[0]; lv.oy = lo[1]; lv.oz = lo[2];
  const int ncells = lv.nx * lv.ny * lv.nz;
  const int gridM = (M + kB - 1) / kB, nb = (ncells + kScanChunk - 1) / kScanChunk;
  LVF_TRY(t.cell_of.alloc(M)); LVF_TRY(t.counts.alloc(ncells)); LVF_TRY(t.bsums.alloc(nb)); LVF_TRY(t.bsq.alloc(nb));
  LVF_TRY(lv.cell_start.alloc((size_t)ncells + 1)); LVF_TRY(lv.sorted.alloc(M));
  job->M = M; job->ncells = ncells; job->gridM = gridM; job->nb = nb;
  job->g = GridP{lv.ox, lv.oy, lv.oz, lv.cell, lv.inv_cell, lv.nx, lv.ny, lv.nz};
  job->raw = m->raw.p; job->cell_of = t.cell_of.p; job->counts = t.counts.p; job->bsums = t.bsums.p; job->bsq = t.bsq.p;
  job->cell_start = lv.cell_start.p; job->sorted = lv.sorted.p; job->sumsq = sumsq_dev;
  return LVF_OK;
}
__global__ void k_bounds_init(int n, unsigned* __restrict__ bounds) {      // [n][min xyz | max xyz] as ordered uints
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * n) bounds[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
}

// n map indices at once (one, or the old-frame maps of a set of loop-closure candidates: relocator.cpp:196-206 -> mapping.cpp:251-262, one
// BuildOldMapFrame + kd-tree build per candidate and cloud).  The host waits of a build are shared between the maps — one wait for all
// bounding boxes — and, round 5, between the LEVELS: the pyramid rule (coarsest cell = gate radius / 2; a finer level, cell halved, while
// the point-weighted cell population is above the target) needs a level's occupancy before it knows whether the next one is wanted, which
// made a build one stream wait and six launches PER LEVEL.  Now every level that might be wanted is built in the SAME round — the levels of
// one map are jobs of the table launches like the maps of a batch are — as long as it has at most 2 cells per map point; one wait returns all the
// occupancies, the levels behind the first one that meets the target are dropped.  The pyramid kept is the one the level-by-level build
// keeps; 16 maps of 3 levels: 64 stream waits -> 4 in round 4 -> 2.
// src_is_device[i]: map_xyz[i] already lives in HBM (a lvf_cloud): no upload.
static int map_create_many(lvf_ctx* ctx, int n, const float* const* map_xyz, const bool* src_is_device, const int* M, int stride_floats, const float* max_radius2, lvf_map** out) {
  hipStream_t s = ctx->stream;
  struct Item { std::unique_ptr<lvf_map> m; float lo[3], hi[3], cell; lvf_map::Level built[LVF_MAX_GRID_LEVELS]; int nb = 0; bool growing = false; int round_first = 0, round_n = 0; };
  std::vector<Item> it((size_t)n);
  std::vector<DevBuf<float>> src((size_t)n);
  std::vector<HostPin<float>> stage((size_t)n);
  // job tables travel through one pinned block per step (every step ends with a stream wait before the block is written again)
  HostPin<unsigned char> h_tab; DevBuf<unsigned char> d_tab;
  StreamWaitGuard stage_guard(s);          // (declared AFTER every pinned block: destroyed first) every path out waits for the copies before the blocks return to the pool
  DevBuf<unsigned> bounds; DevBuf<unsigned long long> sumsq;
  const size_t max_jobs = (size_t)n * LVF_MAX_GRID_LEVELS;
  LVF_TRY(bounds.alloc((size_t)6 * n)); LVF_TRY(sumsq.alloc(max_jobs));
  hipLaunchKernelGGL(k_bounds_init, dim3((6 * n + 255) / 256), dim3(256), 0, s, n, bounds.p);
  LVF_TRY(h_tab.reserve(max_jobs * std::max(sizeof(PackJob), sizeof(LevelJob)))); LVF_TRY(d_tab.alloc(max_jobs * std::max(sizeof(PackJob), sizeof(LevelJob))));
  PackJob* pj = reinterpret_cast<PackJob*>(h_tab.p);
  int n_pack = 0, max_pack_blocks = 0;
  for (int i = 0; i < n; ++i) {
    it[i].m.reset(new lvf_map());
    lvf_map* m = it[i].m.get();
    m->ctx = ctx; m->M = M[i];
    if (M[i] == 0) {                       // an empty map: one 1-cell level with no points
      auto& lv = m->levels[0];
      LVF_TRY(lv.cell_start.alloc(2));
      LVF_HIP(hipMemsetAsync(lv.cell_start.p, 0, 2 * sizeof(int), s));
      lv.cell = std::sqrt(max_radius2[i]); lv.inv_cell = 1.0f / lv.cell;
      m->n_levels = 1;
      continue;
    }
    const bool dev = src_is_device && src_is_device[i];
    if (!dev) LVF_TRY(src[i].upload_staged(map_xyz[i], (size_t)M[i] * stride_floats, s, stage[i]));
    LVF_TRY(m->raw.alloc(M[i]));
    const int nbx = std::min(kBoundsMaxBlocks, (M[i] + kB - 1) / kB);
    pj[n_pack++] = PackJob{M[i], stride_floats, nbx, 0, dev ? map_xyz[i] : src[i].p, m->raw.p, bounds.p + 6 * i};
    max_pack_blocks = std::max(max_pack_blocks, nbx);
  }
  if (n_pack) {
    LVF_HIP(hipMemcpyAsync(d_tab.p, h_tab.p, (size_t)n_pack * sizeof(PackJob), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_bounds_t, dim3(max_pack_blocks, n_pack), dim3(kB), 0, s, reinterpret_cast<const PackJob*>(d_tab.p));
  }
  LVF_HIP(hipGetLastError());
  std::vector<unsigned> hb((size_t)6 * n);
  LVF_TRY(read_back(ctx, hb.data(), bounds.p, hb.size() * sizeof(unsigned)));          // ONE wait for every bounding box
  // Grid pyramid.  Coarsest level: cell = gate radius / 2, so its first two shells cover the whole gate.  Each finer level halves the cell
  // and is kept while the POINT-WEIGHTED cell population (sum count^2 / M) of the level above it is over kTargetOcc: lidar density varies
  // by 100x between 5 m and 30 m range, so the plain mean over cells is dominated by the sparse far field while most queries sit in the
  // dense near field.
  static const double occ_env = [] { const char* e = std::getenv("LVF_KNN_OCC"); return e ? std::atof(e) : 0.0; }();      // experiment knob
  static const int spec_levels = [] { const char* e = std::getenv("LVF_MAP_SPEC_LEVELS"); return e ? std::max(1, std::atoi(e)) : LVF_MAX_GRID_LEVELS; }();      // 1: the level-by-level build of round 4 (A/B)
  const double kMaxCells = 32.0 * 1024 * 1024, kSpecCells = 4.0 * 1024 * 1024, kTargetOcc = occ_env > 0.0 ? occ_env : 128.0;
  auto ncells_for = [](const Item& a, float c) {
    return (std::floor((a.hi[0] - a.lo[0]) / c) + 1) * (std::floor((a.hi[1] - a.lo[1]) / c) + 1) * (std::floor((a.hi[2] - a.lo[2]) / c) + 1);
  };
  int growing = 0;
  for (int i = 0; i < n; ++i) {
    if (M[i] == 0) continue;
    Item& a = it[i];
    for (int k = 0; k < 3; ++k) { a.lo[k] = ord2f(hb[6 * i + k]); a.hi[k] = ord2f(hb[6 * i + 3 + k]); }
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(a.lo[k]) || !std::isfinite(a.hi[k])) { set_error("lvf_map_create: non-finite coordinates in map %d", i); return LVF_ERR_INVALID; }
    a.cell = std::sqrt(max_radius2[i]) * 0.5f;
    while (ncells_for(a, a.cell) > kMaxCells) a.cell *= 1.25f;
    a.growing = true; ++growing;
  }
  std::vector<unsigned long long> hs(max_jobs);
  while (growing > 0) {
    std::vector<LevelTmp> tmp(max_jobs);                             // (live until the round's wait)
    LevelJob* lj = reinterpret_cast<LevelJob*>(h_tab.p);
    int nj = 0, gM = 0, gS = 0, gZ = 0;
    for (int i = 0; i < n; ++i) {
      Item& a = it[i];
      if (!a.growing) continue;
      // the levels of this round: the next one, and every finer one that could still be wanted and is cheap enough to build on spec
      a.round_first = nj; a.round_n = 0;
      float c = a.cell;
      for (int k = 0; a.nb + k < LVF_MAX_GRID_LEVELS && k < spec_levels; ++k, c *= 0.5f) {
        if (k > 0 && ncells_for(a, c) > std::min(kSpecCells, 2.0 * M[i])) break;      // (the finest level kept has about as many cells as the map has points — measured 0.1 .. 1.05 M on the lidar maps of configs[2] and [4]; a level beyond 2 M is 8x that and rarely wanted: it costs a second round when it is)
        LVF_TRY(prepare_level(a.m.get(), a.built[a.nb + k], c, a.lo, a.hi, tmp[nj], sumsq.p + nj, &lj[nj]));
        gM = std::max(gM, lj[nj].gridM); gS = std::max(gS, lj[nj].nb); gZ = std::max(gZ, std::min(512, (lj[nj].ncells + kB - 1) / kB));
        ++nj; ++a.round_n;
      }
    }
    LVF_HIP(hipMemcpyAsync(d_tab.p, h_tab.p, (size_t)nj * sizeof(LevelJob), hipMemcpyHostToDevice, s));
    const LevelJob* dj = reinterpret_cast<const LevelJob*>(d_tab.p);
    // the six steps of a level build, each ONE launch over all the (map, level) jobs of the round
    hipLaunchKernelGGL(k_level_zero_t, dim3(gZ, nj), dim3(kB), 0, s, dj);
    hipLaunchKernelGGL(k_cell_count_t, dim3(gM, nj), dim3(kB), 0, s, dj);
    hipLaunchKernelGGL(k_scan_reduce_t, dim3(gS, nj), dim3(kScanT), 0, s, dj);
    hipLaunchKernelGGL(k_scan_sums_t, dim3(1, nj), dim3(kScanT), 0, s, dj);
    hipLaunchKernelGGL(k_scan_apply_t, dim3(gS, nj), dim3(kScanT), 0, s, dj);
    hipLaunchKernelGGL(k_cell_scatter_t, dim3(gM, nj), dim3(kB), 0, s, dj);
    LVF_HIP(hipGetLastError());
    LVF_TRY(read_back(ctx, hs.data(), sumsq.p, (size_t)nj * sizeof(unsigned long long)));      // ONE wait per round
    for (int i = 0; i < n; ++i) {
      Item& a = it[i];
      if (!a.growing) continue;
      int k = 0;
      for (; k < a.round_n; ++k) {
        const double occ = (double)hs[a.round_first + k] / (double)M[i];
        ++a.nb;
        if (occ <= kTargetOcc || a.nb == LVF_MAX_GRID_LEVELS || ncells_for(a, a.cell * 0.5f) > kMaxCells) { a.growing = false; --growing; ++k; break; }
        a.cell *= 0.5f;
      }
      // levels built on spec behind the last one kept: dropped (their buffers go back to the pool)
      for (int q = k; q < a.round_n; ++q) a.built[a.nb + (q - k)] = lvf_map::Level();
    }
  }
  static const bool info = std::getenv("LVF_MAP_INFO") != nullptr;
  for (int i = 0; i < n; ++i) {
    Item& a = it[i];
    if (M[i] > 0) {
      if (info) { std::fprintf(stderr, "map %d: M %d, %d levels:", i, M[i], a.nb); for (int k = 0; k < a.nb; ++k) std::fprintf(stderr, " %.3f/%d", a.built[k].cell, a.built[k].nx * a.built[k].ny * a.built[k].nz); std::fprintf(stderr, "  (next would be %.0f cells)\n", ncells_for(a, a.built[a.nb - 1].cell * 0.5f)); }
      a.m->n_levels = a.nb;
      for (int k = 0; k < a.nb; ++k) a.m->levels[k] = std::move(a.built[a.nb - 1 - k]);   // finest first
    }
  }
  LVF_HIP(hipStreamSynchronize(s));        // (the empty maps' memsets; the sources of the copies)
  for (int i = 0; i < n; ++i) out[i] = it[i].m.release();
  return LVF_OK;
}

static int map_create_impl(lvf_ctx* ctx, const float* map_xyz, bool src_is_device, int M, int stride_floats, float max_radius2, lvf_map** out) {
  LVF_REQUIRE(ctx && out, "lvf_map_create: null ctx/out");
  LVF_REQUIRE(M >= 0 && (M == 0 || map_xyz) && stride_floats >= 3, "lvf_map_create: bad cloud (M=%d stride=%d)", M, stride_floats);
  LVF_REQUIRE(max_radius2 > 0.0f && std::isfinite(max_radius2), "lvf_map_create: max_radius2 must be finite > 0");
  LVF_TRY(lvf::enter(ctx));
  *out = nullptr;
  return map_create_many(ctx, 1, &map_xyz, &src_is_device, &M, stride_floats, &max_radius2, out);
}

int lvf_map_create(lvf_ctx* ctx, const float* map_xyz, int M, int stride_floats, float max_radius2, lvf_map** out) {
  return map_create_impl(ctx, map_xyz, false, M, stride_floats, max_radius2, out);
}

int lvf_map_create_batch(lvf_ctx* ctx, int n, const float* const* map_xyz, const int* M, int stride_floats, const float* max_radius2, lvf_map** out) {
  if (out && n > 0) for (int i = 0; i < n; ++i) out[i] = nullptr;
  LVF_REQUIRE(ctx && out && (n == 0 || (map_xyz && M && max_radius2)) && n >= 0 && stride_floats >= 3, "lvf_map_create_batch: bad arguments");
  for (int i = 0; i < n; ++i) {
    LVF_REQUIRE(M[i] >= 0 && (M[i] == 0 || map_xyz[i]), "lvf_map_create_batch: bad cloud %d (M=%d)", i, M[i]);
    LVF_REQUIRE(max_radius2[i] > 0.0f && std::isfinite(max_radius2[i]), "lvf_map_create_batch: max_radius2[%d] must be finite > 0", i);
  }
  if (n == 0) return LVF_OK;
  LVF_TRY(lvf::enter(ctx));
  return map_create_many(ctx, n, map_xyz, nullptr, M, stride_floats, max_radius2, out);
}
int lvf_map_create_from_cloud(const lvf_cloud* c, float max_radius2, lvf_map** out) {
  LVF_REQUIRE(c, "lvf_map_create_from_cloud: null cloud");
  return map_create_impl(c->ctx, reinterpret_cast<const float*>(c->pts.p), true, c->n, 4, max_radius2, out);
}

// the indices of n device-resident clouds in one call (Mapping::BuildOldMapFrame's merged clouds of every loop-closure candidate, mapping.cpp:78-137,
// when the per-keyframe clouds live in HBM as lvf_cloud): lvf_map_create_batch without the 6 MB of map points crossing PCIe
int lvf_map_create_batch_from_clouds(lvf_ctx* ctx, int n, const lvf_cloud* const* clouds, const float* max_radius2, lvf_map** out) {
  if (out && n > 0) for (int i = 0; i < n; ++i) out[i] = nullptr;
  LVF_REQUIRE(ctx && out && (n == 0 || (clouds && max_radius2)) && n >= 0, "lvf_map_create_batch_from_clouds: bad arguments");
  if (n == 0) return LVF_OK;
  std::vector<const float*> src((size_t)n);
  std::vector<int> M((size_t)n);
  std::unique_ptr<bool[]> dev(new bool[(size_t)n]);
  for (int i = 0; i < n; ++i) {
    LVF_REQUIRE(clouds[i] && clouds[i]->ctx == ctx, "lvf_map_create_batch_from_clouds: cloud %d is null or of another context", i);
    LVF_REQUIRE(max_radius2[i] > 0.0f && std::isfinite(max_radius2[i]), "lvf_map_create_batch_from_clouds: max_radius2[%d] must be finite > 0", i);
    src[i] = reinterpret_cast<const float*>(clouds[i]->pts.p); M[i] = clouds[i]->n; dev[i] = true;
  }
  LVF_TRY(lvf::enter(ctx));
  return map_create_many(ctx, n, src.data(), dev.get(), M.data(), 4, max_radius2, out);
}

int lvf_map_destroy(lvf_map* m) { delete m; return LVF_OK; }

static int scan_create_impl(lvf_ctx* ctx, const float* scan_xyz, bool src_is_device, int Q, int stride_floats, lvf_scan** out) {
  LVF_REQUIRE(ctx && out, "lvf_scan_create: null ctx/out");
  LVF_REQUIRE(Q >= 0 && (Q == 0 || scan_xyz) && stride_floats >= 3, "lvf_scan_create: bad cloud (Q=%d stride=%d)", Q, stride_floats);
  LVF_TRY(lvf::enter(ctx));
  auto* sc = new lvf_scan();
  sc->ctx = ctx; sc->Q = Q;
  int rc = LVF_OK;
  DevBuf<float> src;
  HostPin<float> stage;
  StreamWaitGuard stage_guard(ctx->stream);      // every path out of this function waits for the copy before the pinned block returns to the pool
  if (Q > 0) {
    if ((!src_is_device && (rc = src.upload_staged(scan_xyz, (size_t)Q * stride_floats, ctx->stream, stage)) != LVF_OK) || (rc = sc->pts.alloc(Q)) != LVF_OK ||
        (rc = sc->idx.alloc((size_t)3 * Q)) != LVF_OK || (rc = sc->d2.alloc((size_t)3 * Q)) != LVF_OK || (rc = sc->valid.alloc(Q)) != LVF_OK) {
      delete sc; return rc;
    }
    hipLaunchKernelGGL(k_pack, dim3((Q + kB - 1) / kB), dim3(kB), 0, ctx->stream, Q, src_is_device ? scan_xyz : src.p, stride_floats, sc->pts.p);
    if (hipGetLastError() != hipSuccess) { delete sc; set_error("lvf_scan_create: launch failed"); return LVF_ERR_HIP; }
    if (!src_is_device) {
      // no wait: everything that touches the scan later runs on this stream, behind the copy.  The staging moves into the scan (a
      // creation per loop-closure candidate and cloud used to cost one stream wait each: 16 per configs[4] batch)
      if (hipEventCreateWithFlags(&sc->create_done, hipEventDisableTiming) == hipSuccess && hipEventRecord(sc->create_done, ctx->stream) == hipSuccess) {
        sc->create_src.swap(src); sc->create_stage.swap(stage);
        stage_guard.dismiss();
      }                                              // (no event: the guard waits, as before)
    }
  } else stage_guard.dismiss();
  *out = sc;
  return LVF_OK;
}
int lvf_scan_create(lvf_ctx* ctx, const float* scan_xyz, int Q, int stride_floats, lvf_scan** out) {
  return scan_create_impl(ctx, scan_xyz, false, Q, stride_floats, out);
}
int lvf_scan_create_from_cloud(const lvf_cloud* c, lvf_scan** out) {
  LVF_REQUIRE(c, "lvf_scan_create_from_cloud: null cloud");
  return scan_create_impl(c->ctx, reinterpret_cast<const float*>(c->pts.p), true, c->n, 4, out);
}

int lvf_scan_destroy(lvf_scan* s) { delete s; return LVF_OK; }      // (~lvf_scan waits for the creation's upload before its pinned block returns to the pool)

// diagnostic (not part of the reference surface): per-query search statistics {candidates, range lookups, last level,
// shells}, and the grid pyramid geometry {cell, nx, ny, nz} per level.
// lvf_knn3_debug_stats2: `stride` int32 per query, 4 <= stride <= 6: {candidates, range lookups, last level, shells[, wave start, wave end clock]}.
// lvf_knn3_debug_stats keeps its round-3 contract ([Q][4]): the record grew to six fields in round 4 under the old name, which overran a
// caller's [Q][4] buffer.
static int knn3_debug_stats_impl(lvf_map* m, lvf_scan* sc, const double* pose, float thr, int32_t* stats, int stride, float* levels4, int* n_levels);
int lvf_knn3_debug_stats(lvf_map* m, lvf_scan* sc, const double* pose, float thr, int32_t* stats4, float* levels4, int* n_levels) {
  return knn3_debug_stats_impl(m, sc, pose, thr, stats4, 4, levels4, n_levels);
}
int lvf_knn3_debug_stats2(lvf_map* m, lvf_scan* sc, const double* pose, float thr, int32_t* stats, int stride, float* levels4, int* n_levels) {
  LVF_REQUIRE(stride >= 4 && stride <= (int)(sizeof(KnnStats) / sizeof(int32_t)), "lvf_knn3_debug_stats2: stride must be 4..%d", (int)(sizeof(KnnStats) / sizeof(int32_t)));
  return knn3_debug_stats_impl(m, sc, pose, thr, stats, stride, levels4, n_levels);
}
static int knn3_debug_stats_impl(lvf_map* m, lvf_scan* sc, const double* pose, float thr, int32_t* stats6, int stride, float* levels4, int* n_levels) {
  LVF_REQUIRE(m && sc && pose && stats6, "lvf_knn3_debug_stats: null argument");
  LVF_TRY(lvf::enter(m->ctx));
  if (n_levels) *n_levels = m->n_levels;
  if (levels4) for (int k = 0; k < m->n_levels; ++k) { levels4[4 * k] = m->levels[k].cell; levels4[4 * k + 1] = (float)m->levels[k].nx; levels4[4 * k + 2] = (float)m->levels[k].ny; levels4[4 * k + 3] = (float)m->levels[k].nz; }
  if (sc->Q == 0) return LVF_OK;
  DevBuf<KnnStats> st;
  LVF_TRY(st.alloc(sc->Q));
  TfArg tf;
  for (int k = 0; k < 7; ++k) tf.v[k] = (float)pose[k];
  const LevelsP L = levels_of(m);
  hipLaunchKernelGGL(k_knn3<true>, dim3(knn_grid(sc->Q)), dim3(kB), 0, m->ctx->stream, sc->Q, sc->pts.p, tf, L, thr,
                     sc->idx.p, sc->d2.p, sc->valid.p, st.p);
  LVF_HIP(hipGetLastError());
  constexpr int kW = (int)(sizeof(KnnStats) / sizeof(int32_t));
  if (stride == kW) {
    LVF_HIP(hipMemcpyAsync(stats6, st.p, (size_t)sc->Q * sizeof(KnnStats), hipMemcpyDeviceToHost, m->ctx->stream));
    LVF_HIP(hipStreamSynchronize(m->ctx->stream));
  } else {
    std::vector<int32_t> h((size_t)sc->Q * kW);
    LVF_HIP(hipMemcpyAsync(h.data(), st.p, (size_t)sc->Q * sizeof(KnnStats), hipMemcpyDeviceToHost, m->ctx->stream));
    LVF_HIP(hipStreamSynchronize(m->ctx->stream));
    for (int q = 0; q < sc->Q; ++q) for (int k = 0; k < stride; ++k) stats6[(size_t)q * stride + k] = h[(size_t)q * kW + k];
  }
  sc->searched = true;
  return LVF_OK;
}

int lvf_knn3(lvf_map* m, lvf_scan* sc, const double* pose, float thr) {
  LVF_REQUIRE(m && sc && pose, "lvf_knn3: null argument");
  LVF_REQUIRE(m->ctx == sc->ctx, "lvf_knn3: map and scan belong to different contexts");
  LVF_REQUIRE(!(thr != thr) && thr > 0.0f, "lvf_knn3: thr must be > 0");
  LVF_TRY(lvf::enter(m->ctx));
  if (sc->Q > 0) {
    TfArg tf;
    for (int k = 0; k < 7; ++k) tf.v[k] = (float)pose[k];   // Sophus SE3d::cast<float>()  association.cpp:287
    const LevelsP L = levels_of(m);
    hipLaunchKernelGGL(k_knn3<false>, dim3(knn_grid(sc->Q)), dim3(kB), 0, m->ctx->stream, sc->Q, sc->pts.p, tf, L, thr,
                       sc->idx.p, sc->d2.p, sc->valid.p, (KnnStats*)nullptr);
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
