// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  The reference-authored loops around the search (transform, gate, block order) are
// pinned through oracle/icp.h's fixture (ref_v3.npz); the SEARCH itself is PCL / FLANN — third-party semantics, PARITY UNPINNED, declared below.
//
// knn.h — scan-to-map association, restating the loops at
//   src/lvio_fusion/src/association.cpp:278-301 (ground) and :336-359 (surf):
//   point = SE3TransformPoint<float>(tf, p_i); kdtree.nearestKSearch(point, 3, idx, d2);
//   accept iff all three d2 < threshold.
// PCL/FLANN are un-vendored and un-pinned.  DECLARED semantics:
//   pcl::KdTreeFLANN<PointXYZI> searches (x,y,z) with flann::L2_Simple<float>: d2 accumulates
//   (dx*dx + dy*dy) + dz*dz in float, no FMA; the search is exact (eps 0) and results are
//   sorted ascending by d2.  Ties are ordered by ascending map index (our declaration).
//   tf = frame->pose.cast<float>() rounds each SE3d coefficient to nearest float
//   (association.cpp:287).
// Two implementations: brute force (the checker) and a leaf-15 single-index kd-tree that
// mirrors FLANN's KDTreeSingleIndex shape (used only as the CPU timing baseline; it is
// itself verified against brute force in tests).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>
#include "se3_ops.h"

namespace lvo {

struct Best3 {
  float d[3]; int i[3];
  Best3() { for (int k = 0; k < 3; ++k) { d[k] = INFINITY; i[k] = -1; } }
  static inline bool less(float da, int ia, float db, int ib) { return da < db || (da == db && (unsigned)ia < (unsigned)ib); }
  inline void push(float dd, int ii) {
    if (!less(dd, ii, d[2], i[2])) return;
    if (less(dd, ii, d[1], i[1])) {
      d[2] = d[1]; i[2] = i[1];
      if (less(dd, ii, d[0], i[0])) { d[1] = d[0]; i[1] = i[0]; d[0] = dd; i[0] = ii; }
      else { d[1] = dd; i[1] = ii; }
    } else { d[2] = dd; i[2] = ii; }
  }
};

inline float dist2_f32(const float* a, const float* b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;  // compiled with -ffp-contract=off
}

// float transform of one query (base.hpp:33-38 instantiated on float)
inline void transform_query_f32(const float tf[7], const float* p, float* out) { Se3Apply<float>(tf, p, out); }

// map / query clouds: xyz stride `stride` floats (4 for pcl::PointXYZI's leading float4)
inline void knn3_brute(const float* map, int M, int mstride, const float* q_world, Best3* out) {
  Best3 b;
  for (int j = 0; j < M; ++j) b.push(dist2_f32(q_world, map + (size_t)j * mstride), j);
  *out = b;
}

// ---- leaf-15 kd-tree (timing baseline) ----
struct KdTree {
  struct Node { int lo, hi; int dim; float split_lo, split_hi; int left, right; };
  const float* pts; int stride; int n;
  std::vector<int> order; std::vector<Node> nodes;
  float bb_lo[3], bb_hi[3];

  void build(const float* p, int n_, int stride_) {
    pts = p; n = n_; stride = stride_;
    order.resize(n); std::iota(order.begin(), order.end(), 0);
    nodes.clear(); nodes.reserve(2 * n / 8 + 16);
    for (int d = 0; d < 3; ++d) { bb_lo[d] = INFINITY; bb_hi[d] = -INFINITY; }
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < 3; ++d) { const float v = p[(size_t)i * stride + d]; bb_lo[d] = std::min(bb_lo[d], v); bb_hi[d] = std::max(bb_hi[d], v); }
    if (n > 0) split(0, n);
  }
  int split(int lo, int hi) {
    const int id = (int)nodes.size();
    nodes.push_back(Node{lo, hi, -1, 0, 0, -1, -1});
    if (hi - lo <= 15) return id;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i)
      for (int d = 0; d < 3; ++d) { const float v = pts[(size_t)order[i] * stride + d]; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); }
    int dim = 0; float span = mx[0] - mn[0];
    for (int d = 1; d < 3; ++d) if (mx[d] - mn[d] > span) { span = mx[d] - mn[d]; dim = d; }
    const int mid = (lo + hi) / 2;
    std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                     [&](int a, int b) { return pts[(size_t)a * stride + dim] < pts[(size_t)b * stride + dim]; });
    float lmax = -INFINITY, rmin = INFINITY;
    for (int i = lo; i < mid; ++i) lmax = std::max(lmax, pts[(size_t)order[i] * stride + dim]);
    for (int i = mid; i < hi; ++i) rmin = std::min(rmin, pts[(size_t)order[i] * stride + dim]);
    const int l = split(lo, mid);
    const int r = split(mid, hi);
    nodes[id].dim = dim; nodes[id].split_lo = lmax; nodes[id].split_hi = rmin; nodes[id].left = l; nodes[id].right = r;
    return id;
  }
  void search(int id, const float* q, Best3& b) const {
    const Node& nd = nodes[id];
    if (nd.dim < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) { const int j = order[i]; b.push(dist2_f32(q, pts + (size_t)j * stride), j); }
      return;
    }
    const float v = q[nd.dim];
    const float dl = v - nd.split_lo, dr = nd.split_hi - v;   // distance to the far child's slab
    const bool go_left = (dl + (v - nd.split_hi)) < 0.0f;     // nearer to the left child
    const int first = go_left ? nd.left : nd.right, second = go_left ? nd.right : nd.left;
    search(first, q, b);
    const float gap = go_left ? dr : dl;
    if (gap <= 0.0f || gap * gap <= b.d[2]) search(second, q, b);  // <= keeps ties reachable
  }
  Best3 query(const float* q) const { Best3 b; if (n > 0) search(0, q, b); return b; }
};

}  // namespace lvo
