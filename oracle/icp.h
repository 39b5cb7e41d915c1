// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  The PROBLEM this file builds — float transform, gate, which points become
// LidarPlaneErrorRPZ / YXY blocks in which order, the loss function, the PoseErrorRPZ / YXY block — is PINNED to the reference's own text
// (round 4): association.cpp:270-384 runs unmodified in oracle/_ref (ceres::Problem as a recorder, KdTreeFLANN as the declared exact
// search) and the recorded blocks, evaluated through CostFunction::Evaluate, equal this restatement (tests/test_oracle_ref.py, ref_v3.npz).
// The SOLVE stays PARITY UNPINNED (Ceres is external: solver semantics are DECLARED); independently checked against a numpy dense
// normal-equation solve over all unknowns in tests/test_oracle_lm_numpy.py.
//
// icp.h — one scan-to-map sub-problem as the reference builds and solves it:
//   FeatureAssociation::ScanToMapWithGround / ScanToMapWithSegmented   src/lvio_fusion/src/association.cpp:270-384
//   + ceres::Solve(DENSE_QR, max_num_iterations = 4)                    src/lvio_fusion/src/mapping.cpp:154-178
// Solver semantics as declared in oracle/lm.h (LM with D^2 = clamp(diag J^T J), radius 1e4, min_relative_decrease 1e-3);
// DENSE_QR solves the same damped least-squares step as the normal equations used here.
#pragma once
#include <vector>
#include "factors.h"
#include "knn.h"
#include "robust.h"

namespace lvo {

struct IcpOut { double initial_cost, final_cost; int nres, iters, successes; };

inline void icp_solve(const float* map, int M, int mstride, const float* query, int Q, int qstride, const double* map_pose,
                      const double* frame_pose, double* rpyxyz, int mode, float thr, double weight, double huber_a, double prior_w,
                      int max_iters, bool use_kdtree, IcpOut* out) {
  float tf[7];
  for (int k = 0; k < 7; ++k) tf[k] = (float)frame_pose[k];
  KdTree tree;
  if (use_kdtree) tree.build(map, M, mstride);
  std::vector<double> P, PA, N;
  for (int i = 0; i < Q; ++i) {
    float w[3];
    transform_query_f32(tf, query + (size_t)i * qstride, w);
    Best3 b;
    if (use_kdtree) b = tree.query(w); else knn3_brute(map, M, mstride, w, &b);
    if (!(b.i[0] >= 0 && b.d[0] < thr && b.i[1] >= 0 && b.d[1] < thr && b.i[2] >= 0 && b.d[2] < thr)) continue;
    const float* q = query + (size_t)i * qstride;
    double pa[3], pb[3], pc[3], n[3];
    for (int k = 0; k < 3; ++k) { pa[k] = map[(size_t)b.i[0] * mstride + k]; pb[k] = map[(size_t)b.i[1] * mstride + k]; pc[k] = map[(size_t)b.i[2] * mstride + k]; }
    PlaneNormal(pa, pb, pc, n);
    for (int k = 0; k < 3; ++k) { P.push_back(q[k]); PA.push_back(pa[k]); N.push_back(n[k]); }
  }
  const int nv = (int)P.size() / 3;
  const int i0 = mode == 0 ? 1 : 0, i1 = mode == 0 ? 2 : 3, i2 = mode == 0 ? 5 : 4;
  double x[3] = {rpyxyz[i0], rpyxyz[i1], rpyxyz[i2]};
  const double x0[3] = {x[0], x[1], x[2]};
  const double w2 = prior_w * prior_w;
  auto cost_at = [&](const double* xx) {
    double c = 0.0;
    for (int i = 0; i < nv; ++i) {
      double r;
      if (mode == 0) LidarPlaneRpzResidual<double>(&P[3 * i], &PA[3 * i], &N[3 * i], map_pose, rpyxyz, weight, xx, xx + 1, xx + 2, &r);
      else LidarPlaneYxyResidual<double>(&P[3 * i], &PA[3 * i], &N[3 * i], map_pose, rpyxyz, weight, xx, xx + 1, xx + 2, &r);
      double rho[3]; loss_eval(huber_a, r * r, rho); c += 0.5 * rho[0];
    }
    if (prior_w > 0.0) for (int k = 0; k < 3; ++k) c += 0.5 * w2 * (xx[k] - x0[k]) * (xx[k] - x0[k]);
    return c;
  };
  double radius = 1e4, decrease = 2.0, cost = 0.0;
  out->iters = 0; out->successes = 0; out->nres = nv + (prior_w > 0.0 ? 1 : 0);
  // the loop of ceres::Solve (DENSE_QR, mapping.cpp:159-163), in TrustRegionMinimizer's order — see lm.h lm_solve for the declared semantics
  bool first = true;
  int invalid_run = 0;
  double h0[3] = {0, 0, 0};      // Jacobi scaling: diag(J^T J) at iteration 0, frozen for the solve (lm.h header, robust.h lm_damping)
  for (;;) {
    double H[3][3] = {}, g[3] = {};
    cost = 0.0;
    for (int i = 0; i < nv; ++i) {
      Jet<3> a(x[0], 0), b(x[1], 1), c(x[2], 2), rr;
      if (mode == 0) LidarPlaneRpzResidual(&P[3 * i], &PA[3 * i], &N[3 * i], map_pose, rpyxyz, weight, &a, &b, &c, &rr);
      else LidarPlaneYxyResidual(&P[3 * i], &PA[3 * i], &N[3 * i], map_pose, rpyxyz, weight, &a, &b, &c, &rr);
      double rho[3]; loss_eval(huber_a, rr.a * rr.a, rho); cost += 0.5 * rho[0];
      const double sc = corrector_scale(rho);
      const double rs = sc * rr.a, J[3] = {sc * rr.v[0], sc * rr.v[1], sc * rr.v[2]};
      for (int u = 0; u < 3; ++u) { g[u] += J[u] * rs; for (int v = 0; v < 3; ++v) H[u][v] += J[u] * J[v]; }
    }
    if (prior_w > 0.0) for (int k = 0; k < 3; ++k) { H[k][k] += w2; g[k] += w2 * (x[k] - x0[k]); cost += 0.5 * w2 * (x[k] - x0[k]) * (x[k] - x0[k]); }
    if (first) { out->initial_cost = cost; first = false; for (int u = 0; u < 3; ++u) h0[u] = H[u][u]; }
    if (out->iters >= max_iters) break;
    if (std::fmax(std::fabs(g[0]), std::fmax(std::fabs(g[1]), std::fabs(g[2]))) <= 1e-10) break;
    if (radius < 1e-32) break;
    double A[3][3], D[3];
    for (int u = 0; u < 3; ++u) { D[u] = lm_damping(H[u][u], h0[u]) / radius; for (int v = 0; v < 3; ++v) A[u][v] = H[u][v]; A[u][u] += D[u]; }
    // 3x3 Cholesky solve A dx = -g
    const double l00 = std::sqrt(A[0][0]), l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    const double t11 = A[1][1] - l10 * l10, l11 = std::sqrt(t11), l21 = (A[2][1] - l20 * l10) / l11;
    const double t22 = A[2][2] - l20 * l20 - l21 * l21, l22 = std::sqrt(t22);
    bool ok = A[0][0] > 0 && t11 > 0 && t22 > 0;
    double dx[3] = {0, 0, 0};
    if (ok) {
      const double y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11, y2 = (-g[2] - l20 * y0 - l21 * y1) / l22;
      dx[2] = y2 / l22; dx[1] = (y1 - l21 * dx[2]) / l11; dx[0] = (y0 - l10 * dx[1] - l20 * dx[2]) / l00;
      ok = std::isfinite(dx[0]) && std::isfinite(dx[1]) && std::isfinite(dx[2]);
    }
    double model = 0.0;
    for (int u = 0; u < 3; ++u) { double hd = 0; for (int v = 0; v < 3; ++v) hd += H[u][v] * dx[v]; model -= dx[u] * (g[u] + 0.5 * hd); }
    const double xc[3] = {x[0] + dx[0], x[1] + dx[1], x[2] + dx[2]};
    const double cand = (ok && model > 0.0) ? cost_at(xc) : cost;
    if (!(ok && model > 0.0 && std::isfinite(cand))) {          // invalid step
      out->iters += 1;
      if (++invalid_run >= 5) break;
      radius *= 0.5;
      continue;
    }
    invalid_run = 0;
    const double dn = std::sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]), xn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (dn <= 1e-8 * (xn + 1e-8)) break;                            // parameter tolerance: the candidate is not taken
    if (std::fabs(cost - cand) <= 1e-6 * cost) break;              // function tolerance, BEFORE the step-quality test: not taken either
    out->iters += 1;
    const double rho = (cost - cand) / model;
    if (rho > 1e-3) {
      x[0] = xc[0]; x[1] = xc[1]; x[2] = xc[2]; cost = cand; out->successes += 1;
      const double t = 2.0 * rho - 1.0;
      radius = std::fmin(radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16); decrease = 2.0;
    } else { radius /= decrease; decrease *= 2.0; }
  }
  out->final_cost = cost;
  rpyxyz[i0] = x[0]; rpyxyz[i1] = x[1]; rpyxyz[i2] = x[2];
}

}  // namespace lvo
