// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  PINNED to the reference's own text (round 4): src/projection.cpp and
// src/association.cpp are compiled UNMODIFIED into oracle/_ref (oracle/ref_driver_lidar.cpp, container stand-ins under oracle/ref_shim/) and
// lidar_extract<true> below equals them bit for bit — range image, ground / label images, segmented cloud, ring indices, curvatures, picks —
// live in the build container and everywhere through tests/golden/ref_v3.npz.  The PCL tail (lidar_extract_tail) stays third-party semantics.
//
// extract.h — LiDAR feature extraction as the reference performs it per keyframe scan
//   FeatureAssociation::Process -> Preprocess / ImageProjection::Process / Extract
//   src/lvio_fusion/src/association.cpp:86-235, src/lvio_fusion/src/projection.cpp:26-320,
//   include/lvio_fusion/lidar/projection.h:36-86, include/lvio_fusion/utility.h:70-90
// restated literally (sequential loops, BFS labelling with the reference's queue order, float arithmetic with the reference's
// float/double promotions).  DECLARED where the reference is not self-contained:
//   * `abs(angle) <= 10` (projection.cpp:130) is the float overload (libstdc++ <cmath>), not abs(int);
//   * `curvatures` is a never-cleared heap array (association.h:23): entries the current scan does not compute (k < 5,
//     k >= size - 5) hold whatever an earlier scan left; the oracle reads them as 0 (fresh allocation);
//   * pcl::removeNaNFromPointCloud drops points with a non-finite x, y or z.
#pragma once
#include "cr_math.h"
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>
#include "cloud.h"

namespace lvo {

struct LidarParams {
  int num_scans, horizon_scan, ground_rows;
  float ang_res_y, ang_bottom;
  float min_range, max_range;
  float resolution;
  double cycle_time;                    // FeatureAssociation::cycle_time_ is a double (association.h:60)
};

struct ExtractDebug {
  std::vector<float> filtered;          // [n][4] after Preprocess
  std::vector<float> range_mat;         // [rows*cols], FLT_MAX = empty
  std::vector<int> pixel_src;           // [rows*cols] index into filtered, -1 = empty
  std::vector<signed char> ground_mat;  // 1 = ground
  std::vector<int> label_mat;           // -1 ground/empty, 999999 outlier, > 0 segment id
  std::vector<float> segmented;         // [m][4]
  std::vector<unsigned char> seg_ground;
  std::vector<float> seg_range;
  std::vector<int> seg_col, start_ring, end_ring;
  std::vector<float> curvature;
  std::vector<float> ground_raw, surf_raw;   // ExtractFeatures' picks before the PCL filters, [k][4]
};

// LIBM = true: the per-point atan2 is libm's float atan2 (what the reference's unqualified atan2(float, float) resolves to) — with it the
// restatement equals the reference's own text BIT FOR BIT (pinned: tests/test_oracle_ref.py against oracle/_ref and tests/golden/ref_v3.npz).
// LIBM = false (default, what the GPU is compared with): cr_atan2f, the correctly rounded single-precision arc tangent (oracle/cr_math.h) —
// glibc's atan2f is within 1 ulp but not correctly rounded, and no device can call it; the two differ in the last bit of a handful of
// AdjustDistortion intensities per scan and never moved a range-image pixel, a ground flag or a segment label on the test scans (counted
// and bounded in the same test: the deviation is a tested exception, not a silent one).
template <bool LIBM>
inline float atan2_sel(float y, float x) { return LIBM ? std::atan2(y, x) : cr_atan2f(y, x); }
#define AT2 atan2_sel<LIBM>
template <bool LIBM = false>
inline void lidar_extract(const float* pts, int n, int stride, const LidarParams& P, ExtractDebug& D) {
  const int R = P.num_scans, Cn = P.horizon_scan;
  const float ang_res_x = 360.0 / float(Cn);
  const float ang_res_y = P.ang_res_y, ang_bottom = P.ang_bottom;
  const float alpha_x = ang_res_x / 180.0 * M_PI, alpha_y = ang_res_y / 180.0 * M_PI;
  const float theta = 60.0 / 180.0 * M_PI;
  // ---- Preprocess (association.cpp:97-102, utility.h:70-90)
  D.filtered.clear();
  for (int i = 0; i < n; ++i) {
    const float* p = pts + (size_t)i * stride;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    const float d = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (d > P.min_range * P.min_range && d < P.max_range * P.max_range) { D.filtered.insert(D.filtered.end(), p, p + 3); D.filtered.push_back(0.0f); }
  }
  const int m = (int)D.filtered.size() / 4;
  const float* F = D.filtered.data();
  D.range_mat.assign((size_t)R * Cn, FLT_MAX); D.pixel_src.assign((size_t)R * Cn, -1);
  D.ground_mat.assign((size_t)R * Cn, 0); D.label_mat.assign((size_t)R * Cn, 0);
  std::vector<float> full((size_t)R * Cn * 4, NAN);
  for (size_t i = 0; i < (size_t)R * Cn; ++i) full[4 * i + 3] = -1.0f;
  float start_ori = 0, end_ori = 0, ori_diff = 1;
  if (m > 0) {   // FindStartEndAngle, projection.cpp:42-56
    start_ori = -AT2(F[1], F[0]);
    end_ori = -AT2(F[4 * (size_t)(m - 1) + 1], F[4 * (size_t)(m - 1)]) + 2 * M_PI;
    if (end_ori - start_ori > 3 * M_PI) end_ori -= 2 * M_PI;
    else if (end_ori - start_ori < M_PI) end_ori += 2 * M_PI;
    ori_diff = end_ori - start_ori;
  }
  // ---- ProjectPointCloud, projection.cpp:58-98
  for (int i = 0; i < m; ++i) {
    const float x = F[4 * (size_t)i], y = F[4 * (size_t)i + 1], z = F[4 * (size_t)i + 2];
    const float vertical_angle = AT2(z, std::sqrt(x * x + y * y)) * 180 / M_PI;
    const int row = (vertical_angle + ang_bottom) / ang_res_y;
    if (row < 0 || row >= R) continue;
    const float horizon_angle = AT2(x, y) * 180 / M_PI;
    int col = -std::round((horizon_angle - 90.0) / ang_res_x) + Cn / 2;
    if (col >= Cn) col -= Cn;
    if (col < 0 || col >= Cn) continue;
    const float range = std::sqrt(x * x + y * y + z * z);
    const size_t idx = (size_t)col + (size_t)row * Cn;
    D.range_mat[idx] = range; D.pixel_src[idx] = i;
    full[4 * idx] = x; full[4 * idx + 1] = y; full[4 * idx + 2] = z; full[4 * idx + 3] = (float)row + (float)col / 10000.0;
  }
  // ---- RemoveGround, projection.cpp:100-153
  for (int j = 0; j < Cn; ++j)
    for (int i = 0; i < P.ground_rows; ++i) {
      const size_t lo = (size_t)j + (size_t)i * Cn, up = (size_t)j + (size_t)(i + 1) * Cn;
      if (full[4 * lo + 3] == -1 || full[4 * up + 3] == -1) { D.ground_mat[lo] = -1; continue; }
      const float dx = full[4 * up] - full[4 * lo], dy = full[4 * up + 1] - full[4 * lo + 1], dz = full[4 * up + 2] - full[4 * lo + 2];
      const float angle = AT2(dz, std::sqrt(dx * dx + dy * dy)) * 180 / M_PI;
      if (std::abs(angle) <= 10) { D.ground_mat[lo] = 1; D.ground_mat[up] = 1; }
    }
  for (size_t i = 0; i < (size_t)R * Cn; ++i)
    if (D.ground_mat[i] == 1 || D.range_mat[i] == FLT_MAX) D.label_mat[i] = -1;
  // ---- Segment + LabelComponents, projection.cpp:155-320 (BFS with the reference's neighbour order)
  int label_count = 1;
  std::vector<int> qx((size_t)R * Cn), qy((size_t)R * Cn), ax((size_t)R * Cn), ay((size_t)R * Cn);
  const int nb[4][2] = {{-1, 0}, {0, 1}, {0, -1}, {1, 0}};
  for (int row = 0; row < R; ++row)
    for (int col = 0; col < Cn; ++col) {
      if (D.label_mat[(size_t)row * Cn + col] != 0) continue;
      std::vector<bool> line_flag(R, false);
      qx[0] = row; qy[0] = col;
      int qsize = 1, qstart = 0, qend = 1, all = 1;
      ax[0] = row; ay[0] = col;
      while (qsize > 0) {
        const int fx = qx[qstart], fy = qy[qstart];
        --qsize; ++qstart;
        D.label_mat[(size_t)fx * Cn + fy] = label_count;
        for (int k = 0; k < 4; ++k) {
          const int tx = fx + nb[k][0];
          int ty = fy + nb[k][1];
          if (tx < 0 || tx >= R) continue;
          if (ty < 0) ty = Cn - 1;
          if (ty >= Cn) ty = 0;
          if (D.label_mat[(size_t)tx * Cn + ty] != 0) continue;
          const float ra = D.range_mat[(size_t)fx * Cn + fy], rb = D.range_mat[(size_t)tx * Cn + ty];
          const float d1 = std::max(ra, rb), d2 = std::min(ra, rb);
          const float alpha = nb[k][0] == 0 ? alpha_x : alpha_y;
          const float angle = AT2(d2 * std::sin(alpha), (d1 - d2 * std::cos(alpha)));
          if (angle > theta) {
            qx[qend] = tx; qy[qend] = ty; ++qsize; ++qend;
            D.label_mat[(size_t)tx * Cn + ty] = label_count;
            line_flag[tx] = true;
            ax[all] = tx; ay[all] = ty; ++all;
          }
        }
      }
      bool feasible = false;
      if (all >= 30) feasible = true;
      else if (all >= 5) { int c = 0; for (int i = 0; i < R; ++i) c += line_flag[i] ? 1 : 0; if (c >= 3) feasible = true; }
      if (feasible) ++label_count;
      else for (int i = 0; i < all; ++i) D.label_mat[(size_t)ax[i] * Cn + ay[i]] = 999999;
    }
  D.segmented.clear(); D.seg_ground.clear(); D.seg_range.clear(); D.seg_col.clear();
  D.start_ring.assign(R, 0); D.end_ring.assign(R, 0);
  int num = 0;
  for (int i = 0; i < R; ++i) {
    D.start_ring[i] = num - 1 + 5;
    for (int j = 0; j < Cn; ++j) {
      const size_t idx = (size_t)i * Cn + j;
      if (D.label_mat[idx] > 0 || D.ground_mat[idx] == 1) {
        if (D.label_mat[idx] == 999999) continue;
        D.seg_ground.push_back(D.ground_mat[idx] == 1); D.seg_col.push_back(j); D.seg_range.push_back(D.range_mat[idx]);
        D.segmented.insert(D.segmented.end(), &full[4 * idx], &full[4 * idx] + 4);
        ++num;
      }
    }
    D.end_ring[i] = num - 1 - 5;
  }
  // ---- AdjustDistortion (intensity only; association.cpp:112-148)
  bool half_passed = false;
  for (int i = 0; i < num; ++i) {
    float* p = &D.segmented[4 * (size_t)i];
    float ori = -AT2(p[1], p[0]);
    if (!half_passed) {
      if (ori < start_ori - M_PI / 2) ori += 2 * M_PI;
      else if (ori > start_ori + M_PI * 3 / 2) ori -= 2 * M_PI;
      if (ori - start_ori > M_PI) half_passed = true;
    } else {
      ori += 2 * M_PI;
      if (ori < end_ori - M_PI * 3 / 2) ori += 2 * M_PI;
      else if (ori > end_ori + M_PI / 2) ori -= 2 * M_PI;
    }
    const float rel_time = (ori - start_ori) / ori_diff;
    p[3] = int(p[3]) + P.cycle_time * rel_time;
  }
  // ---- CalculateSmoothness, association.cpp:150-167
  D.curvature.assign(std::max(num, 1), 0.0f);
  const float* rg = D.seg_range.data();
  for (int i = 5; i < num - 5; ++i) {
    const float dr = (rg[i + 5] - rg[i - 5]) / 10;
    const float r1 = rg[i + 4] - rg[i - 5] - 9 * dr, r2 = rg[i + 3] - rg[i - 5] - 8 * dr, r3 = rg[i + 2] - rg[i - 5] - 7 * dr;
    const float r4 = rg[i + 1] - rg[i - 5] - 6 * dr, r5 = rg[i] - rg[i - 5] - 5 * dr, r6 = rg[i - 1] - rg[i - 5] - 4 * dr;
    const float r7 = rg[i - 2] - rg[i - 5] - 3 * dr, r8 = rg[i - 3] - rg[i - 5] - 2 * dr, r9 = rg[i - 4] - rg[i - 5] - 1 * dr;
    const float cov = (r1 * r1 + r2 * r2 + r3 * r3 + r4 * r4 + r5 * r5 + r6 * r6 + r7 * r7 + r8 * r8 + r9 * r9) / 9;
    D.curvature[i] = cov * 10 / rg[i];
  }
  // ---- ExtractFeatures' picks, association.cpp:185-208
  D.ground_raw.clear(); D.surf_raw.clear();
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < 6; ++j) {
      const int sp = (D.start_ring[i] * (6 - j) + D.end_ring[i] * j) / 6;
      const int ep = (D.start_ring[i] * (5 - j) + D.end_ring[i] * (j + 1)) / 6 - 1;
      if (sp >= ep) continue;
      for (int k = sp; k <= ep; ++k) {
        if (k < 0 || k >= num) continue;     // (the reference would index out of range; cannot happen for sp < ep rings)
        const float* p = &D.segmented[4 * (size_t)k];
        if (D.seg_ground[k]) D.ground_raw.insert(D.ground_raw.end(), p, p + 4);
        else if (D.curvature[k] < 1.0f) D.surf_raw.insert(D.surf_raw.end(), p, p + 4);
      }
    }
}

#undef AT2

// the PCL tail of ExtractFeatures (association.cpp:210-234): surf -> VoxelGrid -> RadiusOutlierRemoval; ground -> VoxelGrid ->
// SegmentGround; both -> Sensor2Robot
inline void lidar_extract_tail(const ExtractDebug& D, const LidarParams& P, const double* extrinsic, unsigned long long seed, std::vector<float>& ground,
                               std::vector<float>& surf) {
  std::vector<float> s = voxel_filter(D.surf_raw.data(), (int)D.surf_raw.size() / 4, 2 * P.resolution);
  {
    const auto keep = radius_outlier_keep(s.data(), (int)s.size() / 4, 4 * P.resolution, 4);
    std::vector<float> t;
    for (size_t i = 0; i < keep.size(); ++i) if (keep[i]) t.insert(t.end(), &s[4 * i], &s[4 * i] + 4);
    s.swap(t);
  }
  std::vector<float> g = voxel_filter(D.ground_raw.data(), (int)D.ground_raw.size() / 4, 2 * P.resolution);
  {
    double co[4]; int it;
    const auto mask = segment_plane(g.data(), (int)g.size() / 4, 0.1f * P.resolution, 100, seed, co, &it);
    std::vector<float> t;
    for (size_t i = 0; i < mask.size(); ++i) if (mask[i]) t.insert(t.end(), &g[4 * i], &g[4 * i] + 4);
    g.swap(t);
  }
  ground.resize(g.size()); surf.resize(s.size());
  cloud_transform(g.data(), (int)g.size() / 4, extrinsic, ground.data());
  cloud_transform(s.data(), (int)s.size() / 4, extrinsic, surf.data());
}

}  // namespace lvo
