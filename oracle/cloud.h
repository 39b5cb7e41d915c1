// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  PARITY UNPINNED for the PCL algorithms (third-party semantics, declared below);
// the reference-authored pieces — Sensor2Robot (association.cpp:236-247) and AlignScan (:39-64) — are pinned to the reference's own text
// through tests/golden/ref_v3.npz (tests/test_oracle_ref.py).
//
// cloud.h — map-cloud maintenance as the reference performs it with PCL (un-vendored; semantics DECLARED here from the
// upstream implementation, PCL 1.8-1.10):
//   Mapping::MergeScan / Sensor2Robot           src/lvio_fusion/src/mapping.cpp:193-205, association.cpp:236-247
//   pcl::VoxelGrid<PointXYZI>::applyFilter      association.cpp:210-215   (leaf = 2 x resolution, all fields averaged)
//   pcl::RadiusOutlierRemoval<PointXYZI>        association.cpp:217-221   (radius 4 x resolution, min 4 neighbours)
//   pcl::SACSegmentation (plane, RANSAC, 100, optimise) + ExtractIndices   association.cpp:249-268
// Declared: VoxelGrid accumulates each voxel's centroid in FLOAT over the points of the voxel (PCL's order is whatever its
// unstable std::sort leaves; the oracle fixes ascending input index) and emits voxels by ascending index; the plane refit's moments
// are summed exactly in scaled integers (order-independent, see segment_plane); RadiusOutlierRemoval
// keeps a point iff more than min_neighbors points (itself included) lie at squared distance < r^2; RANSAC bookkeeping as
// in pcl::RandomSampleConsensus::computeModel (probability 0.99), sampling by splitmix64(seed, hypothesis, draw) because
// PCL's boost::mt19937 stream cannot be reproduced without PCL.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <vector>
#include "knn.h"

namespace lvo {

// points are [n][4] float (x, y, z, intensity)
inline void cloud_transform(const float* in, int n, const double* pose, float* out) {
  float tf[7];
  for (int k = 0; k < 7; ++k) tf[k] = (float)pose[k];
  for (int i = 0; i < n; ++i) {
    transform_query_f32(tf, in + 4 * (size_t)i, out + 4 * (size_t)i);
    out[4 * (size_t)i + 3] = in[4 * (size_t)i + 3];
  }
}

// FeatureAssociation::AlignScan  src/lvio_fusion/src/association.cpp:39-64: the slice [first, last) of pc1 + pc2 a keyframe at `time`
// gets (iterator offsets computed in double and truncated, as `pc.begin() + size * (...) / (...)` does).  Returns false where the
// reference does.
inline bool align_scan_range(int n1, double stamp1, int n2, double stamp2, double cycle_time, double time, long long* first, long long* last) {
  const double end_time = stamp2 + cycle_time / 2, start_time = stamp1 - cycle_time / 2;
  const int size = n1 + n2;
  if (time - cycle_time / 2 < start_time || time + cycle_time / 2 > end_time) return false;
  *first = (long long)(size * (time - start_time - cycle_time / 2) / (end_time - start_time));
  *last = (long long)(size * (time - start_time + cycle_time / 2) / (end_time - start_time));
  return true;
}

inline std::vector<float> voxel_filter(const float* in, int n, float leaf) {
  std::vector<float> out;
  if (n == 0) return out;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], in[4 * (size_t)i + k]); hi[k] = std::fmax(hi[k], in[4 * (size_t)i + k]); }
  const float inv_leaf = 1.0f / leaf;
  int minb[3], div[3];
  for (int k = 0; k < 3; ++k) { minb[k] = (int)std::floor(lo[k] * inv_leaf); div[k] = (int)std::floor(hi[k] * inv_leaf) - minb[k] + 1; }
  struct Acc { float s[4]; int n; };
  std::map<long long, Acc> vox;     // ordered by voxel index
  for (int i = 0; i < n; ++i) {
    const float* p = in + 4 * (size_t)i;
    const int a = (int)(std::floor(p[0] * inv_leaf) - (float)minb[0]), b = (int)(std::floor(p[1] * inv_leaf) - (float)minb[1]),
              c = (int)(std::floor(p[2] * inv_leaf) - (float)minb[2]);
    const long long idx = a + (long long)b * div[0] + (long long)c * div[0] * div[1];
    auto it = vox.find(idx);
    if (it == vox.end()) { Acc z{{p[0], p[1], p[2], p[3]}, 1}; vox.emplace(idx, z); }
    else { for (int k = 0; k < 4; ++k) it->second.s[k] += p[k]; it->second.n++; }
  }
  out.reserve(vox.size() * 4);
  for (auto& kv : vox) for (int k = 0; k < 4; ++k) out.push_back(kv.second.s[k] / (float)kv.second.n);
  return out;
}

inline std::vector<unsigned char> radius_outlier_keep(const float* in, int n, float radius, int min_neighbors) {
  std::vector<unsigned char> keep(n, 0);
  const float r2 = radius * radius;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float* p = in + 4 * (size_t)i;
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
      const float* q = in + 4 * (size_t)j;
      const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
      const float d = (dx * dx + dy * dy) + dz * dz;
      cnt += d < r2 ? 1 : 0;
    }
    keep[i] = cnt > min_neighbors ? 1 : 0;
  }
  return keep;
}

inline unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline void sample3(unsigned long long seed, int h, int n, int idx[3]) {
  int got = 0;
  for (unsigned draw = 0; got < 3; ++draw) {
    const int v = (int)(splitmix64(seed ^ ((unsigned long long)h << 32) ^ draw) % (unsigned long long)n);
    bool dup = false;
    for (int k = 0; k < got; ++k) dup |= idx[k] == v;
    if (!dup) idx[got++] = v;
  }
}
inline bool plane_from3(const float* a, const float* b, const float* c, float co[4]) {
  const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
  float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
  const float nn = (nx * nx + ny * ny) + nz * nz;
  if (!(nn > 0.0f)) return false;
  const float s = 1.0f / std::sqrt(nn);
  nx *= s; ny *= s; nz *= s;
  co[0] = nx; co[1] = ny; co[2] = nz; co[3] = -((nx * a[0] + ny * a[1]) + nz * a[2]);
  return true;
}
inline int count_inliers(const float* in, int n, const float co[4], float thr, std::vector<unsigned char>* mask) {
  int cnt = 0;
  if (mask) mask->assign(n, 0);
  for (int i = 0; i < n; ++i) {
    const float* p = in + 4 * (size_t)i;
    const float d = ((co[0] * p[0] + co[1] * p[1]) + co[2] * p[2]) + co[3];
    if (std::fabs(d) < thr) { ++cnt; if (mask) (*mask)[i] = 1; }
  }
  return cnt;
}
// symmetric 3x3 eigen decomposition by cyclic Jacobi; returns the eigenvector of the smallest eigenvalue
inline void smallest_eigvec3(const double A_in[9], double v[3]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) A[k] = A_in[k];
  for (int sweep = 0; sweep < 60; ++sweep) {
    if (A[1] * A[1] + A[2] * A[2] + A[5] * A[5] < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double x = A[3 * k + p], y = A[3 * k + q]; A[3 * k + p] = c * x - s * y; A[3 * k + q] = s * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = A[3 * p + k], y = A[3 * q + k]; A[3 * p + k] = c * x - s * y; A[3 * q + k] = s * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = V[3 * k + p], y = V[3 * k + q]; V[3 * k + p] = c * x - s * y; V[3 * k + q] = s * x + c * y; }
      }
  }
  int m = 0;
  for (int k = 1; k < 3; ++k) if (A[4 * k] < A[4 * m]) m = k;
  for (int k = 0; k < 3; ++k) v[k] = V[3 * k + m];
}
// returns the inlier mask; coeff4 = refined plane (normal with n_z >= 0), *iters = hypotheses evaluated
inline std::vector<unsigned char> segment_plane(const float* in, int n, float thr, int max_iterations, unsigned long long seed, double coeff4[4], int* iters) {
  std::vector<unsigned char> mask(n, 0);
  for (int k = 0; k < 4; ++k) coeff4[k] = 0.0;
  *iters = 0;
  if (n < 3) return mask;
  int best = -1, best_count = 0;
  double k = 1.0;
  const double log_probability = std::log(1.0 - 0.99);
  float best_co[4] = {0, 0, 0, 0};
  for (int h = 0; h < max_iterations && (double)h < k; ++h) {
    *iters = h + 1;
    int id[3];
    sample3(seed, h, n, id);
    float co[4];
    if (!plane_from3(in + 4 * (size_t)id[0], in + 4 * (size_t)id[1], in + 4 * (size_t)id[2], co)) continue;
    const int c = count_inliers(in, n, co, thr, nullptr);
    if (c > best_count) {
      best_count = c; best = h;
      for (int q = 0; q < 4; ++q) best_co[q] = co[q];
      const double w = (double)c / (double)n;
      double pno = 1.0 - w * w * w;
      pno = std::max(std::numeric_limits<double>::epsilon(), pno);
      pno = std::min(1.0 - std::numeric_limits<double>::epsilon(), pno);
      k = log_probability / std::log(pno);
    }
  }
  if (best < 0) return mask;
  count_inliers(in, n, best_co, thr, &mask);
  // count and first / second moments of the inliers, accumulated EXACTLY (declared; PCL's own summation order is unpinned): the
  // coordinates are floats, so x and x y are exact doubles; every term is scaled by 2^shift (|coordinate| < 2^e over the WHOLE cloud,
  // shift = 60 - 2 max(e, 0)), rounded once to an integer and summed in integers — the sum does not depend on the order of the
  // additions, which is what lets a parallel implementation reproduce it bit for bit.
  float maxabs = 0.0f;
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) maxabs = std::fmax(maxabs, std::fabs(in[4 * (size_t)i + k]));
  int e2 = 0;
  (void)std::frexp(maxabs, &e2);
  const int shift = 60 - 2 * std::max(e2, 0);
  long long hi[9] = {0}; unsigned long long lo[9] = {0}; long long cnt = 0;
  auto add = [&](int k, double t) { const long long q = std::llrint(std::ldexp(t, shift)); hi[k] += q >> 24; lo[k] += (unsigned long long)(q & 0xffffff); };
  for (int i = 0; i < n; ++i) if (mask[i]) {
    const double x = in[4 * (size_t)i], y = in[4 * (size_t)i + 1], z = in[4 * (size_t)i + 2];
    ++cnt;
    add(0, x); add(1, y); add(2, z); add(3, x * x); add(4, x * y); add(5, x * z); add(6, y * y); add(7, y * z); add(8, z * z);
  }
  double m[10];
  m[0] = (double)cnt;
  for (int k = 0; k < 9; ++k) m[1 + k] = std::ldexp((double)hi[k] * 16777216.0 + (double)lo[k], -shift);
  float co[4] = {best_co[0], best_co[1], best_co[2], best_co[3]};
  if (m[0] >= 3.0) {
    const double inv = 1.0 / m[0], cx = m[1] * inv, cy = m[2] * inv, cz = m[3] * inv;
    const double C[9] = {m[4] * inv - cx * cx, m[5] * inv - cx * cy, m[6] * inv - cx * cz, m[5] * inv - cx * cy, m[7] * inv - cy * cy, m[8] * inv - cy * cz,
                         m[6] * inv - cx * cz, m[8] * inv - cy * cz, m[9] * inv - cz * cz};
    double nv[3];
    smallest_eigvec3(C, nv);
    if (nv[2] < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
    co[0] = (float)nv[0]; co[1] = (float)nv[1]; co[2] = (float)nv[2]; co[3] = (float)(-(nv[0] * cx + nv[1] * cy + nv[2] * cz));
    count_inliers(in, n, co, thr, &mask);
  }
  for (int q = 0; q < 4; ++q) coeff4[q] = co[q];
  return mask;
}

}  // namespace lvo
