// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  PARITY UNPINNED.
//
// robust.h — the upstream-Ceres pieces the reference plugs around its functors, restated
// from their published definitions (Ceres is un-vendored; call sites:
//   HuberLoss(1.0)  src/lvio_fusion/src/backend.cpp:98 ; HuberLoss(0.1) association.cpp:330 ;
//   TrivialLoss association.cpp:272 ;
//   ProductParameterization(EigenQuaternionParameterization, IdentityParameterization(3))
//   backend.cpp:99-101).
#pragma once
#include <cmath>

namespace lvo {

// LossFunction::Evaluate(s, rho[3]) : rho, rho', rho''.  a <= 0 selects TrivialLoss.
inline void loss_eval(double a, double s, double rho[3]) {
  if (a <= 0.0) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; }
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::fmax(2.2250738585072014e-308, a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

// Ceres Corrector: with rho'' <= 0 (always true for Huber/Trivial) residuals and Jacobians
// are both scaled by sqrt(rho').  Returns that scale.
inline double corrector_scale(const double rho[3]) { return std::sqrt(rho[1]); }

// EigenQuaternionParameterization::ComputeJacobian (4x3, row-major) for x = [x,y,z,w]
inline void eigen_quat_plus_jacobian(const double x[4], double j[12]) {
  j[0] = x[3];  j[1] = x[2];  j[2] = -x[1];
  j[3] = -x[2]; j[4] = x[3];  j[5] = x[0];
  j[6] = x[1];  j[7] = -x[0]; j[8] = x[3];
  j[9] = -x[0]; j[10] = -x[1]; j[11] = -x[2];
}
// EigenQuaternionParameterization::Plus : x_plus = q_delta (x) x, q_delta = [sin|d|/|d| d, cos|d|]
inline void eigen_quat_plus(const double x[4], const double d[3], double out[4]) {
  const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n > 0.0) {
    const double s = std::sin(n) / n;
    // Hamilton product in [w,x,y,z]: q_delta * x
    const double qd[4] = {std::cos(n), s * d[0], s * d[1], s * d[2]};
    const double xw[4] = {x[3], x[0], x[1], x[2]};
    double o[4];
    o[0] = qd[0] * xw[0] - qd[1] * xw[1] - qd[2] * xw[2] - qd[3] * xw[3];
    o[1] = qd[0] * xw[1] + qd[1] * xw[0] + qd[2] * xw[3] - qd[3] * xw[2];
    o[2] = qd[0] * xw[2] - qd[1] * xw[3] + qd[2] * xw[0] + qd[3] * xw[1];
    o[3] = qd[0] * xw[3] + qd[1] * xw[2] - qd[2] * xw[1] + qd[3] * xw[0];
    out[0] = o[1]; out[1] = o[2]; out[2] = o[3]; out[3] = o[0];
  } else {
    out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3];
  }
}
// Project an ambient (rows x 7) pose Jacobian to local (rows x 6): J_local = J * blockdiag(P(q), I3)
inline void pose_jac_to_local(const double pose[7], int rows, const double* J7, double* J6) {
  double P[12];
  eigen_quat_plus_jacobian(pose, P);
  for (int r = 0; r < rows; ++r) {
    const double* a = J7 + 7 * r; double* o = J6 + 6 * r;
    for (int c = 0; c < 3; ++c) o[c] = a[0] * P[c] + a[1] * P[3 + c] + a[2] * P[6 + c] + a[3] * P[9 + c];
    o[3] = a[4]; o[4] = a[5]; o[5] = a[6];
  }
}

// Levenberg-Marquardt damping of unknown j under Ceres' Jacobi column scaling (declared in lm.h's header), in UNSCALED terms:
// clamp(s^2 h, 1e-6, 1e32) / s^2 with s = 1 / (1 + sqrt(h0)); h = H_jj of this linearisation, h0 = H_jj at iteration 0 of the solve.
inline double lm_damping(double h, double h0) {
  const double s = 1.0 / (1.0 + std::sqrt(h0)), s2 = s * s;
  return std::fmin(std::fmax(h * s2, 1e-6), 1e32) / s2;
}

}  // namespace lvo
