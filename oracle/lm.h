// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  PARITY UNPINNED (Ceres is external: solver semantics are DECLARED);
// independently checked against a numpy dense normal-equation solve over all unknowns in tests/test_oracle_lm_numpy.py.
//
// lm.h — one sliding-window BA problem as Backend::BuildProblem assembles it
// (src/lvio_fusion/src/backend.cpp:96-183) and one Levenberg-Marquardt iteration as ceres::Solve would
// take it with SPARSE_SCHUR (backend.cpp:206-211).  Ceres is un-vendored; DECLARED solver semantics
// (upstream defaults, restated):
//   * cost = 1/2 sum rho(|r_b|^2); Huber(a) on visual blocks, no loss on ImuError (backend.cpp:98,159);
//   * Corrector with rho'' <= 0: r_b and J_b scaled by sqrt(rho');
//   * pose blocks: ProductParameterization(EigenQuaternion, Identity3): J_local = J_ambient * blockdiag(P(q), I3);
//   * normal equations H = J^T J, g = J^T r; LM diagonal D^2 = clamp(diag H, 1e-6, 1e32) taken on the Jacobi-SCALED system (next bullet but one);
//     step solves (H + D^2 / radius) dx = -g exactly, eliminating the 1x1 inverse-depth blocks first (Schur);
//   * model_cost_change = -dx^T (g + H dx / 2); rho = (cost - cost_new) / model_cost_change;
//     accept iff rho > min_relative_decrease (1e-3): radius /= max(1/3, 1 - (2 rho - 1)^3), decrease_factor = 2;
//     else radius /= decrease_factor, decrease_factor *= 2; an INVALID step (solver failure or model_cost_change <= 0): radius *= 0.5.
//     The whole loop with Ceres' termination order: lm_solve below.
//   * Jacobi column scaling (Solver::Options::jacobi_scaling, a Ceres default the reference leaves on: backend.cpp:206-211 builds its options
//     with defaults).  Upstream (trust_region_minimizer.cc, EvaluateGradientAndJacobian): at ITERATION 0 only, s_j = 1 / (1 + sqrt(H0_jj)),
//     H0_jj = squared norm of column j of the robustified tangent-space Jacobian at the start point; s stays FROZEN for the rest of the
//     solve; every Jacobian is column-scaled by s before the strategy sees it; levenberg_marquardt_strategy.cc clamps the diagonal of the
//     SCALED normal equations, diag_j = clamp(s_j^2 H_jj, 1e-6, 1e32), solves (Js^T Js + diag / radius) y = -Js^T r and the minimizer maps the
//     step back, dx_j = s_j y_j.  Substituting y = dx / s: (H + D^2 / radius) dx = -g with
//         D^2_jj = clamp(s_j^2 H_jj, 1e-6, 1e32) / s_j^2            (lm_damping below),
//     i.e. H_jj itself wherever the clamp is inactive and 1e-6 (1 + sqrt(H0_jj))^2 in a near-zero column.  model_cost_change and the
//     gradient are scale-free (J dx = Js y; the gradient is taken before the columns are scaled).  JacobiScale carries H0 through a solve.
// Unknown ordering of the reduced (camera) system, d = 15 n_kf:
//   [ pose tangent 6 x n_kf (keyframe-major) | (v 3, ba 3, bg 3) x n_kf ].
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include "factors.h"
#include "imu.h"
#include "robust.h"

namespace lvo {

struct Window {
  int n_kf, n_lm;
  double *poses, *vel, *ba, *bg, *inv_depth;  // state (updated in place by an accepted step)
  const double* w_kf;
  Camera cam0, cam1;
  int n_tc; const double *tc_left_ob, *tc_right_ob; const int *tc_lm, *tc_kf;
  int n_tf; const double *tf_first_ob, *tf_ob; const int *tf_lm, *tf_kf1, *tf_kf2;
  int n_po; const double *po_ob; const int *po_kf, *po_pw; const double* po_pwtab;
  int n_imu; const imu::Preint* pre; const int *imu_i, *imu_j;
  const unsigned char* pose_const;  // may be null
  const unsigned char* vbb_const = nullptr;  // may be null; per keyframe bit 0 / 1 / 2: its velocity / accelerometer-bias / gyroscope-bias block is held constant
                                             // (Environment::Optimize, environment.cpp:62-68: SetParameterBlockConstant on all of them)
  // weak-constraint priors (backend.cpp:164-178): prior_a[i] == -2 -> RError (pose_graph.cpp:190), prior_a[i] == -1 -> PoseError(origin = prior_target[i][0..7)) on pose prior_b[i];
  // else PoseGraphError(pose prior_a[i], pose prior_b[i]) with rpyxyz_ = prior_target[i][0..6).  No loss function.
  int n_prior = 0; const int *prior_a = nullptr, *prior_b = nullptr; const double *prior_target = nullptr, *prior_w = nullptr, *prior_v = nullptr;
};

struct Linearization {
  int d, dp;                       // d = 15 n_kf, dp = 6 n_kf
  double cost;
  std::vector<double> B, gc;       // d x d, d
  std::vector<double> E;           // n_lm x dp  (row l = d^2 cost / d rho_l d pose-tangent)
  std::vector<double> C, gr;       // n_lm
};

inline int pose_off(int k) { return 6 * k; }
inline int vbb_off(const Window& w, int k) { return 6 * w.n_kf + 9 * k; }

// ---- residual-only cost at an arbitrary state ----
inline double window_cost(const Window& w, double huber_a, const double* poses, const double* vel, const double* ba,
                          const double* bg, const double* inv_depth) {
  double cost = 0.0;
#pragma omp parallel for reduction(+ : cost) schedule(static)
  for (int i = 0; i < w.n_tc; ++i) {
    double r[2]; const double rho_ = inv_depth[w.tc_lm[i]];
    TwoCameraResidual<double>(w.tc_left_ob + 2 * i, w.tc_right_ob + 2 * i, w.cam0, w.cam1, 5 * w.w_kf[w.tc_kf[i]], &rho_, r);
    double rho[3]; loss_eval(huber_a, r[0] * r[0] + r[1] * r[1], rho); cost += 0.5 * rho[0];
  }
#pragma omp parallel for reduction(+ : cost) schedule(static)
  for (int i = 0; i < w.n_tf; ++i) {
    double r[2]; const double rho_ = inv_depth[w.tf_lm[i]];
    TwoFrameResidual<double>(w.tf_first_ob + 2 * i, w.tf_ob + 2 * i, w.cam0, w.cam1, w.w_kf[w.tf_kf2[i]], &rho_,
                             poses + 7 * w.tf_kf1[i], poses + 7 * w.tf_kf2[i], r);
    double rho[3]; loss_eval(huber_a, r[0] * r[0] + r[1] * r[1], rho); cost += 0.5 * rho[0];
  }
#pragma omp parallel for reduction(+ : cost) schedule(static)
  for (int i = 0; i < w.n_po; ++i) {
    double r[2];
    PoseOnlyResidual<double>(w.po_ob + 2 * i, w.po_pwtab + 3 * w.po_pw[i], w.cam0, w.w_kf[w.po_kf[i]], poses + 7 * w.po_kf[i], r);
    double rho[3]; loss_eval(huber_a, r[0] * r[0] + r[1] * r[1], rho); cost += 0.5 * rho[0];
  }
  for (int f = 0; f < w.n_imu; ++f) {
    const int i = w.imu_i[f], j = w.imu_j[f];
    const double* prm[8] = {poses + 7 * i, vel + 3 * i, ba + 3 * i, bg + 3 * i, poses + 7 * j, vel + 3 * j, ba + 3 * j, bg + 3 * j};
    double r[15];
    imu::imu_error_evaluate(w.pre[f], prm, r, nullptr);
    double s = 0; for (int k = 0; k < 15; ++k) s += r[k] * r[k];
    cost += 0.5 * s;     // loss NULL
  }
  for (int i = 0; i < w.n_prior; ++i) {
    double r[6];
    for (int k = 0; k < 6; ++k) r[k] = 0.0;
    if (w.prior_a[i] >= 0) PoseGraphResidual<double>(w.prior_target + 7 * i, w.prior_w[i], w.prior_v[i], poses + 7 * w.prior_a[i], poses + 7 * w.prior_b[i], r);
    else if (w.prior_a[i] == -2) RErrorResidual<double>(w.prior_target + 7 * i, w.prior_w[i], poses + 7 * w.prior_b[i], r);
    else PosePriorResidual<double>(w.prior_target + 7 * i, w.prior_w[i], w.prior_v[i], poses + 7 * w.prior_b[i], r);
    for (int k = 0; k < 6; ++k) cost += 0.5 * r[k] * r[k];
  }
  return cost;
}

// accumulate one residual block given local Jacobian pieces: up to 4 (offset,width,ptr row-major R x width) camera
// columns and an optional landmark column
struct Piece { int off, width; const double* J; };
inline void accumulate(Linearization& L, int R, const double* r, const Piece* pc, int npc, int lm, const double* Jl) {
  for (int a = 0; a < npc; ++a) {
    for (int i = 0; i < pc[a].width; ++i) {
      double g = 0; for (int k = 0; k < R; ++k) g += pc[a].J[k * pc[a].width + i] * r[k];
      L.gc[pc[a].off + i] += g;
      for (int b = 0; b < npc; ++b)
        for (int j = 0; j < pc[b].width; ++j) {
          double h = 0; for (int k = 0; k < R; ++k) h += pc[a].J[k * pc[a].width + i] * pc[b].J[k * pc[b].width + j];
          L.B[(size_t)(pc[a].off + i) * L.d + pc[b].off + j] += h;
        }
      if (lm >= 0) {
        double e = 0; for (int k = 0; k < R; ++k) e += pc[a].J[k * pc[a].width + i] * Jl[k];
        L.E[(size_t)lm * L.dp + pc[a].off + i] += e;   // only pose pieces ever pair with a landmark
      }
    }
  }
  if (lm >= 0) {
    double c = 0, g = 0; for (int k = 0; k < R; ++k) { c += Jl[k] * Jl[k]; g += Jl[k] * r[k]; }
    L.C[lm] += c; L.gr[lm] += g;
  }
}

inline void window_linearize(const Window& w, double huber_a, Linearization& L) {
  L.d = 15 * w.n_kf; L.dp = 6 * w.n_kf;
  L.B.assign((size_t)L.d * L.d, 0.0); L.gc.assign(L.d, 0.0);
  L.E.assign((size_t)w.n_lm * L.dp, 0.0); L.C.assign(w.n_lm, 0.0); L.gr.assign(w.n_lm, 0.0);
  double cost = 0.0;
  auto is_const = [&](int k) { return w.pose_const && w.pose_const[k]; };
  for (int i = 0; i < w.n_tc; ++i) {
    Jet<1> d(w.inv_depth[w.tc_lm[i]], 0), rr[2];
    TwoCameraResidual(w.tc_left_ob + 2 * i, w.tc_right_ob + 2 * i, w.cam0, w.cam1, 5 * w.w_kf[w.tc_kf[i]], &d, rr);
    double r[2] = {rr[0].a, rr[1].a}, Jl[2] = {rr[0].v[0], rr[1].v[0]};
    double rho[3]; loss_eval(huber_a, r[0] * r[0] + r[1] * r[1], rho); cost += 0.5 * rho[0];
    const double sc = corrector_scale(rho);
    for (int k = 0; k < 2; ++k) { r[k] *= sc; Jl[k] *= sc; }
    accumulate(L, 2, r, nullptr, 0, w.tc_lm[i], Jl);
  }
  for (int i = 0; i < w.n_tf; ++i) {
    const int k1 = w.tf_kf1[i], k2 = w.tf_kf2[i];
    const double* p1 = w.poses + 7 * k1; const double* p2 = w.poses + 7 * k2;
    Jet<15> d(w.inv_depth[w.tf_lm[i]], 0), A[7], Bq[7], rr[2];
    for (int k = 0; k < 7; ++k) { A[k] = Jet<15>(p1[k], 1 + k); Bq[k] = Jet<15>(p2[k], 8 + k); }
    TwoFrameResidual(w.tf_first_ob + 2 * i, w.tf_ob + 2 * i, w.cam0, w.cam1, w.w_kf[k2], &d, A, Bq, rr);
    double r[2] = {rr[0].a, rr[1].a}, Jl[2] = {rr[0].v[0], rr[1].v[0]}, J1[14], J2[14], J1l[12], J2l[12];
    for (int a = 0; a < 2; ++a) for (int k = 0; k < 7; ++k) { J1[7 * a + k] = rr[a].v[1 + k]; J2[7 * a + k] = rr[a].v[8 + k]; }
    double rho[3]; loss_eval(huber_a, r[0] * r[0] + r[1] * r[1], rho); cost += 0.5 * rho[0];
    const double sc = corrector_scale(rho);
    pose_jac_to_local(p1, 2, J1, J1l); pose_jac_to_local(p2, 2, J2, J2l);
    for (int k = 0; k < 2; ++k) { r[k] *= sc; Jl[k] *= sc; }
    for (int k = 0; k < 12; ++k) { J1l[k] *= is_const(k1) ? 0.0 : sc; J2l[k] *= is_const(k2) ? 0.0 : sc; }
    if (k1 == k2) {   // degenerate (never produced by BuildProblem) — fold into one piece
      double Js[12]; for (int k = 0; k < 12; ++k) Js[k] = J1l[k] + J2l[k];
      Piece pc[1] = {{pose_off(k1), 6, Js}};
      accumulate(L, 2, r, pc, 1, w.tf_lm[i], Jl);
    } else {
      Piece pc[2] = {{pose_off(k1), 6, J1l}, {pose_off(k2), 6, J2l}};
      accumulate(L, 2, r, pc, 2, w.tf_lm[i], Jl);
    }
  }
  for (int i = 0; i < w.n_po; ++i) {
    const int k = w.po_kf[i]; const double* p = w.poses + 7 * k;
    Jet<7> T[7], rr[2];
    for (int q = 0; q < 7; ++q) T[q] = Jet<7>(p[q], q);
    PoseOnlyResidual(w.po_ob + 2 * i, w.po_pwtab + 3 * w.po_pw[i], w.cam0, w.w_kf[k], T, rr);
    double r[2] = {rr[0].a, rr[1].a}, J[14], Jl6[12];
    for (int a = 0; a < 2; ++a) for (int q = 0; q < 7; ++q) J[7 * a + q] = rr[a].v[q];
    double rho[3]; loss_eval(huber_a, r[0] * r[0] + r[1] * r[1], rho); cost += 0.5 * rho[0];
    const double sc = corrector_scale(rho);
    pose_jac_to_local(p, 2, J, Jl6);
    for (int q = 0; q < 2; ++q) r[q] *= sc;
    for (int q = 0; q < 12; ++q) Jl6[q] *= is_const(k) ? 0.0 : sc;
    Piece pc[1] = {{pose_off(k), 6, Jl6}};
    accumulate(L, 2, r, pc, 1, -1, nullptr);
  }
  for (int f = 0; f < w.n_imu; ++f) {
    const int i = w.imu_i[f], j = w.imu_j[f];
    const double* prm[8] = {w.poses + 7 * i, w.vel + 3 * i, w.ba + 3 * i, w.bg + 3 * i, w.poses + 7 * j, w.vel + 3 * j, w.ba + 3 * j, w.bg + 3 * j};
    double r[15], Jb[8][105]; double* Jp[8];
    for (int k = 0; k < 8; ++k) Jp[k] = Jb[k];
    imu::imu_error_evaluate(w.pre[f], prm, r, Jp);
    double s = 0; for (int k = 0; k < 15; ++k) s += r[k] * r[k];
    cost += 0.5 * s;
    double Pi[90], Pj[90], Vi[135], Vj[135];
    pose_jac_to_local(prm[0], 15, Jb[0], Pi); pose_jac_to_local(prm[4], 15, Jb[4], Pj);
    if (is_const(i)) std::memset(Pi, 0, sizeof(Pi));
    if (is_const(j)) std::memset(Pj, 0, sizeof(Pj));
    for (int row = 0; row < 15; ++row)
      for (int c = 0; c < 3; ++c) {
        Vi[row * 9 + c] = Jb[1][row * 3 + c]; Vi[row * 9 + 3 + c] = Jb[2][row * 3 + c]; Vi[row * 9 + 6 + c] = Jb[3][row * 3 + c];
        Vj[row * 9 + c] = Jb[5][row * 3 + c]; Vj[row * 9 + 3 + c] = Jb[6][row * 3 + c]; Vj[row * 9 + 6 + c] = Jb[7][row * 3 + c];
      }
    if (w.vbb_const)      // a constant block keeps its residual but gets no Jacobian columns (Ceres drops it from the reduced program)
      for (int row = 0; row < 15; ++row)
        for (int c = 0; c < 9; ++c) {
          if ((w.vbb_const[i] >> (c / 3)) & 1) Vi[row * 9 + c] = 0.0;
          if ((w.vbb_const[j] >> (c / 3)) & 1) Vj[row * 9 + c] = 0.0;
        }
    Piece pc[4] = {{pose_off(i), 6, Pi}, {vbb_off(w, i), 9, Vi}, {pose_off(j), 6, Pj}, {vbb_off(w, j), 9, Vj}};
    accumulate(L, 15, r, pc, 4, -1, nullptr);
  }
  for (int i = 0; i < w.n_prior; ++i) {
    const int a = w.prior_a[i], b = w.prior_b[i];
    double r[6], Ja[42], Jb[42], La[36], Lb[36];
    if (a >= 0) {
      Jet<14> A[7], Bq[7], rr[6];
      for (int k = 0; k < 7; ++k) { A[k] = Jet<14>(w.poses[7 * a + k], k); Bq[k] = Jet<14>(w.poses[7 * b + k], 7 + k); }
      PoseGraphResidual(w.prior_target + 7 * i, w.prior_w[i], w.prior_v[i], A, Bq, rr);
      for (int k = 0; k < 6; ++k) { r[k] = rr[k].a; for (int c = 0; c < 7; ++c) { Ja[7 * k + c] = rr[k].v[c]; Jb[7 * k + c] = rr[k].v[7 + c]; } }
    } else {
      Jet<7> P[7], rr[6];
      for (int k = 0; k < 7; ++k) P[k] = Jet<7>(w.poses[7 * b + k], k);
      if (a == -2) RErrorResidual(w.prior_target + 7 * i, w.prior_w[i], P, rr);   // rows 4,5 stay zero
      else PosePriorResidual(w.prior_target + 7 * i, w.prior_w[i], w.prior_v[i], P, rr);
      for (int k = 0; k < 6; ++k) { r[k] = rr[k].a; for (int c = 0; c < 7; ++c) Jb[7 * k + c] = rr[k].v[c]; }
    }
    for (int k = 0; k < 6; ++k) cost += 0.5 * r[k] * r[k];
    pose_jac_to_local(w.poses + 7 * b, 6, Jb, Lb);
    if (is_const(b)) std::memset(Lb, 0, sizeof(Lb));
    if (a >= 0) {
      pose_jac_to_local(w.poses + 7 * a, 6, Ja, La);
      if (is_const(a)) std::memset(La, 0, sizeof(La));
      Piece pc[2] = {{pose_off(a), 6, La}, {pose_off(b), 6, Lb}};
      accumulate(L, 6, r, pc, 2, -1, nullptr);
    } else {
      Piece pc[1] = {{pose_off(b), 6, Lb}};
      accumulate(L, 6, r, pc, 1, -1, nullptr);
    }
  }
  L.cost = cost;
}

// dense lower Cholesky solve of S x = b in place (S d x d row-major, symmetric positive definite). returns false on breakdown
inline bool chol_solve(std::vector<double>& S, std::vector<double>& b, int d) {
  for (int j = 0; j < d; ++j) {
    double v = S[(size_t)j * d + j];
    for (int k = 0; k < j; ++k) v -= S[(size_t)j * d + k] * S[(size_t)j * d + k];
    if (!(v > 0.0)) return false;
    const double l = std::sqrt(v);
    S[(size_t)j * d + j] = l;
#pragma omp parallel for schedule(static) if (d - j > 256)
    for (int i = j + 1; i < d; ++i) {
      double s = S[(size_t)i * d + j];
      for (int k = 0; k < j; ++k) s -= S[(size_t)i * d + k] * S[(size_t)j * d + k];
      S[(size_t)i * d + j] = s / l;
    }
  }
  for (int i = 0; i < d; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= S[(size_t)i * d + k] * b[k]; b[i] = s / S[(size_t)i * d + i]; }
  for (int i = d - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < d; ++k) s -= S[(size_t)k * d + i] * b[k]; b[i] = s / S[(size_t)i * d + i]; }
  return true;
}

struct LmStep {
  double cost_before, cost_after, model_cost_change, rho;
  double gradient_max_norm, step_norm, x_norm;
  bool accepted, solved;
  std::vector<double> S, rhs;   // reduced system actually solved (damped), for parity taps
  std::vector<double> dx_c, dx_l;
};

inline void apply_step(const Window& w, const std::vector<double>& dc, const std::vector<double>& dl, double* poses, double* vel,
                       double* ba, double* bg, double* inv_depth) {
  for (int k = 0; k < w.n_kf; ++k) {
    const double* p = w.poses + 7 * k; double* o = poses + 7 * k;
    eigen_quat_plus(p, &dc[pose_off(k)], o);
    for (int c = 0; c < 3; ++c) o[4 + c] = p[4 + c] + dc[pose_off(k) + 3 + c];
    const int vo = vbb_off(w, k);
    for (int c = 0; c < 3; ++c) { vel[3 * k + c] = w.vel[3 * k + c] + dc[vo + c]; ba[3 * k + c] = w.ba[3 * k + c] + dc[vo + 3 + c]; bg[3 * k + c] = w.bg[3 * k + c] + dc[vo + 6 + c]; }
  }
  for (int l = 0; l < w.n_lm; ++l) inv_depth[l] = w.inv_depth[l] + dl[l];
}

// Jacobi scaling state of one solve: H0 = diag(J^T J) at iteration 0 (camera part d, landmark part n_lm), frozen once taken.
// (unscaled_clamp: NOT Ceres — the clamp taken on the unscaled diagonal, max(H_jj, 1e-6): what this file did before the scaling was restated;
// kept so that a test can show a window on which the two differ)
struct JacobiScale { std::vector<double> c, l; bool frozen = false; bool unscaled_clamp = false; };
// (lm_damping: robust.h)

// The trial step of one LM iteration at trust-region radius `radius`: linearise, damp, Schur-eliminate, solve, back-substitute, model
// cost change, candidate state and its cost.  Nothing is committed.  cand_* receive the candidate (sizes 7/3/3/3 n_kf, n_lm).
// `js`: the solve's Jacobi scaling (taken from THIS linearisation if not frozen yet); null = a one-iteration solve (H0 = H).
inline void lm_trial_step(const Window& w, double huber_a, double radius, LmStep& out, std::vector<double>& np, std::vector<double>& nv,
                          std::vector<double>& nba, std::vector<double>& nbg, std::vector<double>& nd, JacobiScale* js = nullptr) {
  Linearization L;
  window_linearize(w, huber_a, L);
  const int d = L.d, dp = L.dp, nl = w.n_lm;
  const double mu = radius;
  JacobiScale local;
  if (!js) js = &local;
  if (!js->frozen) {
    js->c.resize(d); js->l.resize(nl);
    for (int i = 0; i < d; ++i) js->c[i] = L.B[(size_t)i * d + i];
    for (int l = 0; l < nl; ++l) js->l[l] = L.C[l];
    js->frozen = true;
  }
  std::vector<double> Dc(d), Dl(nl), Cd(nl);
  auto damp = [&](double h, double h0) { return js->unscaled_clamp ? std::fmin(std::fmax(h, 1e-6), 1e32) : lm_damping(h, h0); };
  for (int i = 0; i < d; ++i) Dc[i] = damp(L.B[(size_t)i * d + i], js->c[i]) / mu;
  for (int l = 0; l < nl; ++l) { Dl[l] = damp(L.C[l], js->l[l]) / mu; Cd[l] = L.C[l] + Dl[l]; }
  out.S = L.B;
  for (int i = 0; i < d; ++i) out.S[(size_t)i * d + i] += Dc[i];
  out.rhs.assign(d, 0.0);
  for (int i = 0; i < d; ++i) out.rhs[i] = -L.gc[i];
  for (int l = 0; l < nl; ++l) {
    const double* e = &L.E[(size_t)l * dp];
    const double ic = 1.0 / Cd[l];
    int nzi[512]; int nz = 0;
    for (int i = 0; i < dp && nz < 512; ++i) if (e[i] != 0.0) nzi[nz++] = i;
    for (int a = 0; a < nz; ++a) {
      const double ea = e[nzi[a]] * ic;
      out.rhs[nzi[a]] += ea * L.gr[l];
      for (int b = 0; b < nz; ++b) out.S[(size_t)nzi[a] * d + nzi[b]] -= ea * e[nzi[b]];
    }
  }
  std::vector<double> Sf = out.S, dc = out.rhs;
  out.cost_before = L.cost;
  // gradient max norm at the linearisation point: max |J^T r| over every unknown (tangent coordinates; constant poses contribute 0)
  out.gradient_max_norm = 0.0;
  for (int i = 0; i < d; ++i) out.gradient_max_norm = std::fmax(out.gradient_max_norm, std::fabs(L.gc[i]));
  for (int l = 0; l < nl; ++l) out.gradient_max_norm = std::fmax(out.gradient_max_norm, std::fabs(L.gr[l]));
  out.solved = chol_solve(Sf, dc, d);
  out.accepted = false; out.cost_after = L.cost; out.model_cost_change = 0; out.rho = 0; out.step_norm = 0; out.x_norm = 0;
  if (!out.solved) return;
  std::vector<double> dl(nl);
  for (int l = 0; l < nl; ++l) {
    const double* e = &L.E[(size_t)l * dp];
    double s = -L.gr[l];
    for (int i = 0; i < dp; ++i) if (e[i] != 0.0) s -= e[i] * dc[i];
    dl[l] = s / Cd[l];
  }
  // model cost change = -dx^T (g + H dx / 2), H undamped
  double m = 0.0;
  for (int i = 0; i < d; ++i) {
    double hb = 0; for (int j = 0; j < d; ++j) hb += L.B[(size_t)i * d + j] * dc[j];
    m += dc[i] * (L.gc[i] + 0.5 * hb);
  }
  for (int l = 0; l < nl; ++l) {
    const double* e = &L.E[(size_t)l * dp];
    double ed = 0; for (int i = 0; i < dp; ++i) if (e[i] != 0.0) ed += e[i] * dc[i];
    m += dl[l] * (L.gr[l] + 0.5 * L.C[l] * dl[l]) + dl[l] * ed;   // cross term counted once (E dc . dl), twice halves
  }
  out.model_cost_change = -m;
  np.assign(7 * w.n_kf, 0.0); nv.assign(3 * w.n_kf, 0.0); nba.assign(3 * w.n_kf, 0.0); nbg.assign(3 * w.n_kf, 0.0); nd.assign(nl, 0.0);
  apply_step(w, dc, dl, np.data(), nv.data(), nba.data(), nbg.data(), nd.data());
  out.cost_after = window_cost(w, huber_a, np.data(), nv.data(), nba.data(), nbg.data(), nd.data());
  out.rho = out.model_cost_change > 0 ? (out.cost_before - out.cost_after) / out.model_cost_change : -1.0;
  out.dx_c = dc; out.dx_l = dl;
  // Ceres: step_norm = |x - x_plus_delta| and x_norm = |x| over the AMBIENT parameter vector of the reduced program (constant blocks
  // are not in it).  Window state: poses of the non-constant keyframes, v / ba / bg of every keyframe, every inverse depth.
  double sn = 0.0, xn = 0.0;
  for (int k = 0; k < w.n_kf; ++k) {
    if (!(w.pose_const && w.pose_const[k]))
      for (int c = 0; c < 7; ++c) { const double a = w.poses[7 * k + c], b = np[7 * k + c]; sn += (a - b) * (a - b); xn += a * a; }
    const int cm = w.vbb_const ? w.vbb_const[k] : 0;
    for (int c = 0; c < 3; ++c) {
      sn += (w.vel[3 * k + c] - nv[3 * k + c]) * (w.vel[3 * k + c] - nv[3 * k + c]) + (w.ba[3 * k + c] - nba[3 * k + c]) * (w.ba[3 * k + c] - nba[3 * k + c]) +
            (w.bg[3 * k + c] - nbg[3 * k + c]) * (w.bg[3 * k + c] - nbg[3 * k + c]);
      xn += ((cm & 1) ? 0.0 : w.vel[3 * k + c] * w.vel[3 * k + c]) + ((cm & 2) ? 0.0 : w.ba[3 * k + c] * w.ba[3 * k + c]) + ((cm & 4) ? 0.0 : w.bg[3 * k + c] * w.bg[3 * k + c]);
    }
  }
  for (int l = 0; l < nl; ++l) { sn += (w.inv_depth[l] - nd[l]) * (w.inv_depth[l] - nd[l]); xn += w.inv_depth[l] * w.inv_depth[l]; }
  out.step_norm = std::sqrt(sn); out.x_norm = std::sqrt(xn);
}

inline void commit(Window& w, const std::vector<double>& np, const std::vector<double>& nv, const std::vector<double>& nba,
                   const std::vector<double>& nbg, const std::vector<double>& nd) {
  std::memcpy(w.poses, np.data(), np.size() * 8); std::memcpy(w.vel, nv.data(), nv.size() * 8);
  std::memcpy(w.ba, nba.data(), nba.size() * 8); std::memcpy(w.bg, nbg.data(), nbg.size() * 8);
  std::memcpy(w.inv_depth, nd.data(), nd.size() * 8);
}

// LevenbergMarquardtStrategy's three radius updates (upstream levenberg_marquardt_strategy.cc, restated)
inline void radius_step_accepted(double rho, double* radius, double* decrease_factor) {
  const double t = 2.0 * rho - 1.0;
  *radius = std::fmin(*radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16);      // max_trust_region_radius 1e16
  *decrease_factor = 2.0;
}
inline void radius_step_rejected(double* radius, double* decrease_factor) { *radius = *radius / *decrease_factor; *decrease_factor *= 2.0; }
inline void radius_step_invalid(double* radius) { *radius *= 0.5; }      // StepIsInvalid: decrease_factor untouched

// One iteration with the radius handed in (no termination tests): the per-iteration parity point of lvf_problem_lm_iteration.
inline void lm_iteration(Window& w, double huber_a, double min_relative_decrease, double* radius, double* decrease_factor, LmStep& out, JacobiScale* js = nullptr) {
  std::vector<double> np, nv, nba, nbg, nd;
  lm_trial_step(w, huber_a, *radius, out, np, nv, nba, nbg, nd, js);
  const bool valid = out.solved && out.model_cost_change > 0.0;
  if (!valid) { radius_step_invalid(radius); return; }
  if (out.rho > min_relative_decrease) {
    out.accepted = true;
    commit(w, np, nv, nba, nbg, nd);
    radius_step_accepted(out.rho, radius, decrease_factor);
    return;
  }
  radius_step_rejected(radius, decrease_factor);
}

// ---- the whole solve: ceres::Solve's TrustRegionMinimizer loop, restated ------------------------------------------------------------
// (what adapt::Solve runs: backend.cpp:206-211, mapping.cpp:159-163.  Ceres is un-vendored; DECLARED from upstream
// trust_region_minimizer.cc, in its order:)
//   IterationZero: evaluate; gradient_max_norm <= gradient_tolerance -> CONVERGENCE.
//   loop, at the top (FinalizeIterationAndCheckIfMinimizerCanContinue): iterations >= max_num_iterations -> NO_CONVERGENCE;
//     gradient_max_norm <= gradient_tolerance -> CONVERGENCE; radius < min_trust_region_radius (1e-32) -> CONVERGENCE;
//   ComputeTrustRegionStep: linear solver failure or model_cost_change <= 0 -> INVALID step: max_num_consecutive_invalid_steps (5) in a
//     row -> FAILURE, else radius *= 0.5 and the iteration counts as unsuccessful;
//   candidate = x [+] delta; step_norm <= parameter_tolerance (x_norm + parameter_tolerance) -> CONVERGENCE (candidate NOT taken);
//   candidate cost; |cost - candidate_cost| <= function_tolerance * cost -> CONVERGENCE (candidate NOT taken; tested BEFORE acceptance);
//   relative_decrease > min_relative_decrease -> accept (x = candidate, re-linearise, StepAccepted) else StepRejected.
struct SolveOptions {
  int max_num_iterations; double huber_a, initial_trust_region_radius, function_tolerance, gradient_tolerance, parameter_tolerance, min_relative_decrease;
  bool unscaled_clamp = false;      // test-only: see JacobiScale
};
enum { LVO_CONVERGENCE = 0, LVO_NO_CONVERGENCE = 1, LVO_FAILURE = 2 };
enum { LVO_WHY_NONE = 0, LVO_WHY_GRADIENT = 1, LVO_WHY_PARAMETER = 2, LVO_WHY_FUNCTION = 3, LVO_WHY_MIN_RADIUS = 4, LVO_WHY_MAX_ITERATIONS = 5, LVO_WHY_INVALID_STEPS = 6 };
struct SolveSummary {
  double initial_cost, final_cost, final_radius, final_decrease_factor;
  int num_iterations, num_successful_steps, num_unsuccessful_steps, termination, why, num_trials;
};
// trace (optional): per TRIAL step [cost_before, cost_after, radius used, accepted, valid, rho] (6 doubles), at most max_num_iterations + 1 rows
inline void lm_solve(Window& w, const SolveOptions& o, SolveSummary& s, double* trace = nullptr) {
  double radius = o.initial_trust_region_radius, decrease = 2.0;
  std::memset(&s, 0, sizeof(s));
  s.termination = LVO_NO_CONVERGENCE; s.why = LVO_WHY_MAX_ITERATIONS;
  int invalid_run = 0, n_trials = 0;
  bool have_cost = false;
  double cost = 0.0;
  LmStep st;
  std::vector<double> np, nv, nba, nbg, nd;
  JacobiScale js;            // taken at the first linearisation (iteration 0), frozen for the whole solve
  js.unscaled_clamp = o.unscaled_clamp;
  for (;;) {
    // (the trial step re-linearises at x: after a rejected step that reproduces the previous linearisation — Ceres keeps it)
    lm_trial_step(w, o.huber_a, radius, st, np, nv, nba, nbg, nd, &js);
    if (!have_cost) { s.initial_cost = st.cost_before; have_cost = true; }
    cost = st.cost_before;
    if (s.num_iterations >= o.max_num_iterations) { s.termination = LVO_NO_CONVERGENCE; s.why = LVO_WHY_MAX_ITERATIONS; break; }   // (<= 0: only the initial cost)
    if (st.gradient_max_norm <= o.gradient_tolerance) { s.termination = LVO_CONVERGENCE; s.why = LVO_WHY_GRADIENT; break; }
    if (radius < 1e-32) { s.termination = LVO_CONVERGENCE; s.why = LVO_WHY_MIN_RADIUS; break; }
    // num_iterations counts what Ceres pushes to Summary::iterations beyond iteration 0: accepted, rejected and invalid steps.  A trial
    // step that ends the solve through the parameter / function tolerance returns before it is recorded.
    double* tr = trace ? trace + 6 * n_trials : nullptr;
    n_trials += 1;
    const bool valid = st.solved && st.model_cost_change > 0.0;
    if (tr) { tr[0] = st.cost_before; tr[1] = st.cost_after; tr[2] = radius; tr[3] = 0; tr[4] = valid ? 1 : 0; tr[5] = st.rho; }
    if (!valid) {
      s.num_iterations += 1; s.num_unsuccessful_steps += 1;
      if (++invalid_run >= 5) { s.termination = LVO_FAILURE; s.why = LVO_WHY_INVALID_STEPS; break; }
      radius_step_invalid(&radius);
      continue;
    }
    invalid_run = 0;
    if (st.step_norm <= o.parameter_tolerance * (st.x_norm + o.parameter_tolerance)) { s.termination = LVO_CONVERGENCE; s.why = LVO_WHY_PARAMETER; break; }
    if (std::fabs(st.cost_before - st.cost_after) <= o.function_tolerance * st.cost_before) { s.termination = LVO_CONVERGENCE; s.why = LVO_WHY_FUNCTION; break; }
    s.num_iterations += 1;
    if (st.rho > o.min_relative_decrease) {
      commit(w, np, nv, nba, nbg, nd);
      cost = st.cost_after;
      radius_step_accepted(st.rho, &radius, &decrease);
      s.num_successful_steps += 1;
      if (tr) tr[3] = 1;
    } else {
      radius_step_rejected(&radius, &decrease);
      s.num_unsuccessful_steps += 1;
    }
  }
  s.num_trials = n_trials;
  s.final_cost = cost; s.final_radius = radius; s.final_decrease_factor = decrease;
}

}  // namespace lvo
