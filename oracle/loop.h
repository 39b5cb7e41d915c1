// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  PARITY UNPINNED for the solver semantics (Ceres is external: DECLARED
// as in oracle/lm.h); the RelocateRError functor itself is pinned bit-for-bit against the reference text (tests/test_oracle_ref.py).
//
// loop.h — the loop-correction tail after the multi-GPU candidate gather:
//   Relocator::UpdateNewSubmap's rotation solve   src/lvio_fusion/src/relocator.cpp:247-268
//       one quaternion parameter r (EigenQuaternionParameterization, initial value = the caller's, identity in the reference),
//       one RelocateRError<7,4>(relocated_i, unrelocated_i) per keyframe of the new submap, no loss, ceres::Solve(DENSE_QR) with
//       default options (LM, <= 50 iterations);
//   PoseGraph::ForwardUpdate                       src/lvio_fusion/src/pose_graph.cpp:245-252
//       pose <- transform * pose (Sophus SE3 product: Hamilton product re-normalised, t_T + R(q_T) t), Vw <- R(q_T) Vw.
#pragma once
#include <algorithm>
#include <cmath>
#include "factors.h"
#include "robust.h"

namespace lvo {

struct RelocOut { double initial_cost, final_cost; int iters, successes, termination; };

inline void relocate_rotation_solve(int n, const double* relocated, const double* unrelocated, double* q4, int max_iters, double function_tol,
                                    double gradient_tol, double parameter_tol, double min_rel_decrease, double radius0, RelocOut* out) {
  auto cost_at = [&](const double* q) {
    double c = 0.0;
    for (int i = 0; i < n; ++i) {
      double r[7];
      RelocateRResidual<double>(relocated + 7 * i, unrelocated + 7 * i, q, r);
      for (int k = 0; k < 7; ++k) c += 0.5 * r[k] * r[k];
    }
    return c;
  };
  double q[4] = {q4[0], q4[1], q4[2], q4[3]};
  double radius = radius0, decrease = 2.0, cost = 0.0;
  out->iters = 0; out->successes = 0; out->termination = 1;
  // ceres::Solve's TrustRegionMinimizer order (declared in lm.h lm_solve)
  bool first = true;
  int invalid_run = 0;
  double h0[3] = {0, 0, 0};      // Jacobi scaling: diag(J^T J) at iteration 0, frozen for the solve (lm.h header, robust.h lm_damping)
  for (;;) {
    double H[3][3] = {}, g[3] = {};
    cost = 0.0;
    double P[12];
    eigen_quat_plus_jacobian(q, P);
    for (int i = 0; i < n; ++i) {
      Jet<4> Q[4], rr[7];
      for (int k = 0; k < 4; ++k) Q[k] = Jet<4>(q[k], k);
      RelocateRResidual(relocated + 7 * i, unrelocated + 7 * i, Q, rr);
      for (int k = 0; k < 7; ++k) {
        cost += 0.5 * rr[k].a * rr[k].a;
        double Jl[3];
        for (int c = 0; c < 3; ++c) Jl[c] = rr[k].v[0] * P[c] + rr[k].v[1] * P[3 + c] + rr[k].v[2] * P[6 + c] + rr[k].v[3] * P[9 + c];
        for (int u = 0; u < 3; ++u) { g[u] += Jl[u] * rr[k].a; for (int v = 0; v < 3; ++v) H[u][v] += Jl[u] * Jl[v]; }
      }
    }
    if (first) { out->initial_cost = cost; first = false; for (int u = 0; u < 3; ++u) h0[u] = H[u][u]; }
    if (out->iters >= max_iters) break;                               // NO_CONVERGENCE (termination stays 1)
    if (std::fmax(std::fabs(g[0]), std::fmax(std::fabs(g[1]), std::fabs(g[2]))) <= gradient_tol) { out->termination = 0; break; }
    if (radius < 1e-32) { out->termination = 0; break; }              // "minimum trust region radius reached" is a CONVERGENCE in Ceres
    double A[3][3], D[3];
    for (int u = 0; u < 3; ++u) { D[u] = lm_damping(H[u][u], h0[u]) / radius; for (int v = 0; v < 3; ++v) A[u][v] = H[u][v]; A[u][u] += D[u]; }
    const double l00 = std::sqrt(A[0][0]), l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    const double t11 = A[1][1] - l10 * l10, l11 = std::sqrt(t11), l21 = (A[2][1] - l20 * l10) / l11;
    const double t22 = A[2][2] - l20 * l20 - l21 * l21, l22 = std::sqrt(t22);
    const bool ok = A[0][0] > 0 && t11 > 0 && t22 > 0;
    double dx[3] = {0, 0, 0};
    if (ok) {
      const double y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11, y2 = (-g[2] - l20 * y0 - l21 * y1) / l22;
      dx[2] = y2 / l22; dx[1] = (y1 - l21 * dx[2]) / l11; dx[0] = (y0 - l10 * dx[1] - l20 * dx[2]) / l00;
    }
    double model = 0.0;
    for (int u = 0; u < 3; ++u) { double hd = 0; for (int v = 0; v < 3; ++v) hd += H[u][v] * dx[v]; model -= dx[u] * (g[u] + 0.5 * hd); }
    double qc[4];
    eigen_quat_plus(q, dx, qc);
    if (!(ok && model > 0.0)) {                                        // invalid step
      out->iters += 1;
      if (++invalid_run >= 5) { out->termination = 2; break; }
      radius *= 0.5;
      continue;
    }
    invalid_run = 0;
    double d2 = 0.0; for (int k = 0; k < 4; ++k) d2 += (qc[k] - q[k]) * (qc[k] - q[k]);          // |x - x_plus_delta|, ambient
    const double dn = std::sqrt(d2), xn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (dn <= parameter_tol * (xn + parameter_tol)) { out->termination = 0; break; }
    const double cand = cost_at(qc);
    if (std::fabs(cost - cand) <= function_tol * cost) { out->termination = 0; break; }           // before the step-quality test; candidate not taken
    out->iters += 1;
    const double rho = (cost - cand) / model;
    if (rho > min_rel_decrease) {
      for (int k = 0; k < 4; ++k) q[k] = qc[k];
      cost = cand; out->successes += 1;
      const double t = 2.0 * rho - 1.0;
      radius = std::fmin(radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16); decrease = 2.0;
    } else { radius /= decrease; decrease *= 2.0; }
  }
  out->final_cost = cost;
  for (int k = 0; k < 4; ++k) q4[k] = q[k];
}

// pose_graph.cpp:245-252.  poses [n][7] and vw [n][3] (may be null) are updated in place.
inline void forward_update(const double T[7], int n, double* poses, double* vw) {
  const double qn = std::sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2] + T[3] * T[3]);
  const double ux = T[0] / qn, uy = T[1] / qn, uz = T[2] / qn, uw = T[3] / qn;
  auto rot = [&](const double* v, double* o) {     // Eigen QuaternionBase::_transformVector: uv = 2 u x v ; v + w uv + u x uv
    const double cx = 2.0 * (uy * v[2] - uz * v[1]), cy = 2.0 * (uz * v[0] - ux * v[2]), cz = 2.0 * (ux * v[1] - uy * v[0]);
    o[0] = v[0] + uw * cx + (uy * cz - uz * cy); o[1] = v[1] + uw * cy + (uz * cx - ux * cz); o[2] = v[2] + uw * cz + (ux * cy - uy * cx);
  };
  for (int i = 0; i < n; ++i) {
    double* p = poses + 7 * i;
    const double bx = p[0], by = p[1], bz = p[2], bw = p[3];
    double w = uw * bw - ux * bx - uy * by - uz * bz, x = uw * bx + ux * bw + uy * bz - uz * by, y = uw * by + uy * bw + uz * bx - ux * bz,
           z = uw * bz + uz * bw + ux * by - uy * bx;
    const double nn = std::sqrt(w * w + x * x + y * y + z * z);
    double t[3];
    rot(p + 4, t);
    p[0] = x / nn; p[1] = y / nn; p[2] = z / nn; p[3] = w / nn;
    p[4] = T[4] + t[0]; p[5] = T[5] + t[1]; p[6] = T[6] + t[2];
    if (vw) { double v[3]; rot(vw + 3 * i, v); vw[3 * i] = v[0]; vw[3 * i + 1] = v[1]; vw[3 * i + 2] = v[2]; }
  }
}

}  // namespace lvo
