// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).  PARITY PINNED (round 3): tests/test_oracle_ref.py compares this
// file BIT FOR BIT with the reference's own imu_error.hpp / preintegration.{h,cpp} / utility.h compiled unmodified into oracle/_ref
// (live where /root/reference exists, and everywhere against tests/golden/ref_v2.npz generated from it).
//
// imu.h — IMU pre-integration (mid-point) and the ImuError factor with its analytic
// Jacobians, restating
//   src/lvio_fusion/src/preintegration.cpp:12-165          (Preintegration)
//   src/lvio_fusion/include/lvio_fusion/ceres/imu_error.hpp:12-122   (ImuError::Evaluate)
//   src/lvio_fusion/include/lvio_fusion/utility.h:99-140   (q_delta, skew_symmetric, q_left, q_right)
// Eigen (un-vendored) semantics DECLARED here and, identically, in oracle/ref_shim/Eigen/Core (real Eigen's last-ulp rounding
// depends on its version / vector ISA; the shim fixes k-ascending products without FMA): Quaternion*Vector = v + w*uv + vec x uv with
// uv = 2 vec x v; Quaternion::inverse = conjugate / squaredNorm; toRotationMatrix as in
// Eigen/Geometry; 15x15 inverse() = partial-pivot LU solve against the identity;
// LLT = lower Cholesky reading the lower triangle.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace lvo {
namespace imu {

constexpr int O_T = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12, O_PR = 0, O_PT = 4;  // preintegration.cpp:12
constexpr double G[3] = {0.0, 0.0, 9.81007};                                         // preintegration.cpp:13

struct Quat { double x, y, z, w; };  // Eigen coefficient order

inline Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
inline Quat qinv(const Quat& q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  return Quat{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
}
inline void qrot(const Quat& q, const double v[3], double out[3]) {
  // uv = 2 * q.vec x v ; out = v + w*uv + q.vec x uv
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
inline void qmat(const Quat& q, double R[3][3]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz;       R[0][2] = txz + twy;
  R[1][0] = txy + twz;       R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy;       R[2][1] = tyz + twx;       R[2][2] = 1 - (txx + tyy);
}
inline Quat qnormalized(const Quat& q) {
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
}
// utility.h:99-112
inline Quat q_delta(const double theta[3]) { return Quat{theta[0] / 2.0, theta[1] / 2.0, theta[2] / 2.0, 1.0}; }
// utility.h:114-122
inline void skew(const double q[3], double S[3][3]) {
  S[0][0] = 0;     S[0][1] = -q[2]; S[0][2] = q[1];
  S[1][0] = q[2];  S[1][1] = 0;     S[1][2] = -q[0];
  S[2][0] = -q[1]; S[2][1] = q[0];  S[2][2] = 0;
}
// utility.h:124-140 ; sign=+1 -> q_left, sign=-1 -> q_right ; 4x4 in [w | vec] ordering
inline void q_lr(const Quat& q, double sign, double M[4][4]) {
  const double v[3] = {q.x, q.y, q.z};
  double S[3][3];
  skew(v, S);
  M[0][0] = q.w;
  for (int j = 0; j < 3; ++j) { M[0][1 + j] = -v[j]; M[1 + j][0] = v[j]; }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[1 + i][1 + j] = (i == j ? q.w : 0.0) + sign * S[i][j];
}

struct Noise { double acc_n, gyr_n, acc_w, gyr_w; };  // Imu::{ACC_N,GYR_N,ACC_W,GYR_W}, kitti.yaml:48-51

struct Preint {
  double sum_dt;
  double acc0[3], gyr0[3];
  double lin_ba[3], lin_bg[3];
  double dp[3];
  Quat dq;
  double dv[3];
  double jac[15][15];
  double cov[15][15];
  double noise[18];  // diagonal of the 18x18 noise matrix, preintegration.cpp:21-27

  // Preintegration ctor preintegration.cpp:15-28 + first-sample latch of Append (preintegration.h:29-41)
  void init(const double ba[3], const double bg[3], const double a0[3], const double g0[3], const Noise& n) {
    std::memset(this, 0, sizeof(*this));
    for (int i = 0; i < 3; ++i) { lin_ba[i] = ba[i]; lin_bg[i] = bg[i]; acc0[i] = a0[i]; gyr0[i] = g0[i]; }
    dq = Quat{0, 0, 0, 1};
    for (int i = 0; i < 15; ++i) jac[i][i] = 1.0;
    for (int i = 0; i < 3; ++i) {
      noise[0 + i] = n.acc_n * n.acc_n; noise[3 + i] = n.gyr_n * n.gyr_n;
      noise[6 + i] = n.acc_n * n.acc_n; noise[9 + i] = n.gyr_n * n.gyr_n;
      noise[12 + i] = n.acc_w * n.acc_w; noise[15 + i] = n.gyr_w * n.gyr_w;
    }
  }

  // MidPointIntegration + Propagate  preintegration.cpp:30-127
  void propagate(double dt, const double acc1[3], const double gyr1[3]) {
    double a0b[3], a1b[3], un_gyr[3];
    for (int i = 0; i < 3; ++i) {
      a0b[i] = acc0[i] - lin_ba[i];
      a1b[i] = acc1[i] - lin_ba[i];
      un_gyr[i] = 0.5 * (gyr0[i] + gyr1[i]) - lin_bg[i];
    }
    double un_acc_0[3], un_acc_1[3], un_acc[3];
    qrot(dq, a0b, un_acc_0);
    const Quat rq = qmul(dq, Quat{un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0});
    qrot(rq, a1b, un_acc_1);
    double rp[3], rv[3];
    for (int i = 0; i < 3; ++i) {
      un_acc[i] = 0.5 * (un_acc_0[i] + un_acc_1[i]);
      rp[i] = dp[i] + dv[i] * dt + 0.5 * un_acc[i] * dt * dt;
      rv[i] = dv[i] + un_acc[i] * dt;
    }
    // --- jacobian / covariance propagation  :49-98
    double Rw[3][3], Ra0[3][3], Ra1[3][3], R0[3][3], R1[3][3];
    skew(un_gyr, Rw); skew(a0b, Ra0); skew(a1b, Ra1);
    qmat(dq, R0); qmat(rq, R1);
    double ImRw[3][3];  // I - R_w_x*dt
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ImRw[i][j] = (i == j ? 1.0 : 0.0) - Rw[i][j] * dt;
    double R0a0[3][3], R1a1[3][3], R1a1I[3][3];
    mm3(R0, Ra0, R0a0); mm3(R1, Ra1, R1a1); mm3(R1a1, ImRw, R1a1I);

    static thread_local double F[15][15], V[15][18];
    std::memset(F, 0, sizeof(F)); std::memset(V, 0, sizeof(V));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double I = (i == j) ? 1.0 : 0.0;
        F[0 + i][0 + j] = I;
        F[0 + i][3 + j] = -0.25 * R0a0[i][j] * dt * dt + -0.25 * R1a1I[i][j] * dt * dt;
        F[0 + i][6 + j] = I * dt;
        F[0 + i][9 + j] = -0.25 * (R0[i][j] + R1[i][j]) * dt * dt;
        F[0 + i][12 + j] = -0.25 * R1a1[i][j] * dt * dt * -dt;
        F[3 + i][3 + j] = ImRw[i][j];
        F[3 + i][12 + j] = -1.0 * I * dt;
        F[6 + i][3 + j] = -0.5 * R0a0[i][j] * dt + -0.5 * R1a1I[i][j] * dt;
        F[6 + i][6 + j] = I;
        F[6 + i][9 + j] = -0.5 * (R0[i][j] + R1[i][j]) * dt;
        F[6 + i][12 + j] = -0.5 * R1a1[i][j] * dt * -dt;
        F[9 + i][9 + j] = I;
        F[12 + i][12 + j] = I;

        V[0 + i][0 + j] = 0.25 * R0[i][j] * dt * dt;
        V[0 + i][3 + j] = 0.25 * -R1a1[i][j] * dt * dt * 0.5 * dt;
        V[0 + i][6 + j] = 0.25 * R1[i][j] * dt * dt;
        V[0 + i][9 + j] = V[0 + i][3 + j];
        V[3 + i][3 + j] = 0.5 * I * dt;
        V[3 + i][9 + j] = 0.5 * I * dt;
        V[6 + i][0 + j] = 0.5 * R0[i][j] * dt;
        V[6 + i][3 + j] = 0.5 * -R1a1[i][j] * dt * 0.5 * dt;
        V[6 + i][6 + j] = 0.5 * R1[i][j] * dt;
        V[6 + i][9 + j] = V[6 + i][3 + j];
        V[9 + i][12 + j] = I * dt;
        V[12 + i][15 + j] = I * dt;
      }
    double nj[15][15], FC[15][15], nc[15][15];
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0, t = 0;
        for (int k = 0; k < 15; ++k) { s += F[i][k] * jac[k][j]; t += F[i][k] * cov[k][j]; }
        nj[i][j] = s; FC[i][j] = t;
      }
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0, t = 0;
        for (int k = 0; k < 15; ++k) s += FC[i][k] * F[j][k];
        for (int k = 0; k < 18; ++k) t += V[i][k] * noise[k] * V[j][k];
        nc[i][j] = s + t;
      }
    std::memcpy(jac, nj, sizeof(jac));
    std::memcpy(cov, nc, sizeof(cov));
    // --- Propagate tail  :116-126
    for (int i = 0; i < 3; ++i) { dp[i] = rp[i]; dv[i] = rv[i]; acc0[i] = acc1[i]; gyr0[i] = gyr1[i]; }
    dq = qnormalized(rq);
    sum_dt += dt;
  }

  static void mm3(const double A[3][3], const double B[3][3], double C[3][3]) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
  }
};

inline void mv3(const double M[3][3], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = M[i][0] * v[0] + M[i][1] * v[1] + M[i][2] * v[2];
}
inline void block3(const double J[15][15], int r, int c, double B[3][3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i][j] = J[r + i][c + j];
}

// 15x15 inverse: partial-pivot LU, then solve against identity columns (declared Eigen semantics)
inline void inverse15(const double A[15][15], double X[15][15]) {
  double LU[15][15]; int piv[15];
  std::memcpy(LU, A, sizeof(LU));
  for (int k = 0; k < 15; ++k) {
    int p = k; double best = std::fabs(LU[k][k]);
    for (int i = k + 1; i < 15; ++i) { const double a = std::fabs(LU[i][k]); if (a > best) { best = a; p = i; } }
    piv[k] = p;
    if (p != k) for (int j = 0; j < 15; ++j) { const double t = LU[k][j]; LU[k][j] = LU[p][j]; LU[p][j] = t; }
    for (int i = k + 1; i < 15; ++i) {
      LU[i][k] /= LU[k][k];
      const double l = LU[i][k];
      for (int j = k + 1; j < 15; ++j) LU[i][j] -= l * LU[k][j];
    }
  }
  for (int c = 0; c < 15; ++c) {
    double b[15];
    for (int i = 0; i < 15; ++i) b[i] = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < 15; ++k) { const double t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
    for (int i = 0; i < 15; ++i) { double s = b[i]; for (int j = 0; j < i; ++j) s -= LU[i][j] * b[j]; b[i] = s; }
    for (int i = 14; i >= 0; --i) { double s = b[i]; for (int j = i + 1; j < 15; ++j) s -= LU[i][j] * b[j]; b[i] = s / LU[i][i]; }
    for (int i = 0; i < 15; ++i) X[i][c] = b[i];
  }
}
// lower Cholesky reading the lower triangle; returns sqrt_info = L^T  (imu_error.hpp:32)
inline void sqrt_info_from_cov(const double cov[15][15], double S[15][15]) {
  double Ainv[15][15], L[15][15];
  inverse15(cov, Ainv);
  std::memset(L, 0, sizeof(L));
  for (int j = 0; j < 15; ++j) {
    double d = Ainv[j][j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    L[j][j] = std::sqrt(d);
    for (int i = j + 1; i < 15; ++i) {
      double s = Ainv[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      L[i][j] = s / L[j][j];
    }
  }
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) S[i][j] = L[j][i];
}

// Preintegration::Evaluate  preintegration.cpp:144-165  (unweighted residual)
inline void preint_residual(const Preint& P, const double Pi[3], const Quat& Qi, const double Vi[3],
                            const double Bai[3], const double Bgi[3], const double Pj[3], const Quat& Qj,
                            const double Vj[3], const double Baj[3], const double Bgj[3], double r[15]) {
  double dp_dba[3][3], dp_dbg[3][3], dq_dbg[3][3], dv_dba[3][3], dv_dbg[3][3];
  block3(P.jac, O_T, O_BA, dp_dba); block3(P.jac, O_T, O_BG, dp_dbg); block3(P.jac, O_R, O_BG, dq_dbg);
  block3(P.jac, O_V, O_BA, dv_dba); block3(P.jac, O_V, O_BG, dv_dbg);
  double dba[3], dbg[3];
  for (int i = 0; i < 3; ++i) { dba[i] = Bai[i] - P.lin_ba[i]; dbg[i] = Bgi[i] - P.lin_bg[i]; }
  double th[3]; mv3(dq_dbg, dbg, th);
  const Quat cq = qmul(P.dq, q_delta(th));
  double a[3], b[3], cv[3], cp[3];
  mv3(dv_dba, dba, a); mv3(dv_dbg, dbg, b);
  for (int i = 0; i < 3; ++i) cv[i] = P.dv[i] + a[i] + b[i];
  mv3(dp_dba, dba, a); mv3(dp_dbg, dbg, b);
  for (int i = 0; i < 3; ++i) cp[i] = P.dp[i] + a[i] + b[i];
  const Quat Qi_inv = qinv(Qi);
  double t[3], o[3];
  for (int i = 0; i < 3; ++i) t[i] = 0.5 * G[i] * P.sum_dt * P.sum_dt + Pj[i] - Pi[i] - Vi[i] * P.sum_dt;
  qrot(Qi_inv, t, o);
  for (int i = 0; i < 3; ++i) r[O_T + i] = o[i] - cp[i];
  const Quat e = qmul(qinv(cq), qmul(Qi_inv, Qj));
  r[O_R + 0] = 2 * e.x; r[O_R + 1] = 2 * e.y; r[O_R + 2] = 2 * e.z;
  for (int i = 0; i < 3; ++i) t[i] = G[i] * P.sum_dt + Vj[i] - Vi[i];
  qrot(Qi_inv, t, o);
  for (int i = 0; i < 3; ++i) r[O_V + i] = o[i] - cv[i];
  for (int i = 0; i < 3; ++i) { r[O_BA + i] = Baj[i] - Bai[i]; r[O_BG + i] = Bgj[i] - Bgi[i]; }
}

// ImuError::Evaluate  imu_error.hpp:17-113.
// params: [pose_i(7), v_i(3), ba_i(3), bg_i(3), pose_j(7), v_j(3), ba_j(3), bg_j(3)]
// J[k] row-major 15 x size_k, may be null individually; J itself may be null.
inline void imu_error_evaluate(const Preint& P, const double* const* prm, double* res, double** J) {
  const Quat Qi{prm[0][0], prm[0][1], prm[0][2], prm[0][3]};
  const double* Pi = prm[0] + 4; const double* Vi = prm[1]; const double* Bai = prm[2]; const double* Bgi = prm[3];
  const Quat Qj{prm[4][0], prm[4][1], prm[4][2], prm[4][3]};
  const double* Pj = prm[4] + 4; const double* Vj = prm[5]; const double* Baj = prm[6]; const double* Bgj = prm[7];
  double r0[15];
  preint_residual(P, Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, r0);
  double S[15][15];
  sqrt_info_from_cov(P.cov, S);
  for (int i = 0; i < 15; ++i) { double s = 0; for (int k = 0; k < 15; ++k) s += S[i][k] * r0[k]; res[i] = s; }
  if (!J) return;
  const double sum_dt = P.sum_dt;
  double dp_dba[3][3], dp_dbg[3][3], dq_dbg[3][3], dv_dba[3][3], dv_dbg[3][3];
  block3(P.jac, O_T, O_BA, dp_dba); block3(P.jac, O_T, O_BG, dp_dbg); block3(P.jac, O_R, O_BG, dq_dbg);
  block3(P.jac, O_V, O_BA, dv_dba); block3(P.jac, O_V, O_BG, dv_dbg);
  const Quat Qi_inv = qinv(Qi);
  double Ri_inv[3][3]; qmat(Qi_inv, Ri_inv);
  double dbg[3]; for (int i = 0; i < 3; ++i) dbg[i] = Bgi[i] - P.lin_bg[i];
  double th[3]; mv3(dq_dbg, dbg, th);
  const Quat cq = qmul(P.dq, q_delta(th));

  auto finish = [&](double M[15][7], int cols, double* out) {  // out = sqrt_info * M, row-major 15 x cols
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < cols; ++j) { double s = 0; for (int k = 0; k < 15; ++k) s += S[i][k] * M[k][j]; out[i * cols + j] = s; }
  };
  double M[15][7];
  if (J[0]) {
    std::memset(M, 0, sizeof(M));
    double t[3], o[3], Sk[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[O_T + i][O_PT + j] = -Ri_inv[i][j];
    for (int i = 0; i < 3; ++i) t[i] = 0.5 * G[i] * sum_dt * sum_dt + Pj[i] - Pi[i] - Vi[i] * sum_dt;
    qrot(Qi_inv, t, o); skew(o, Sk);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[O_T + i][O_PR + j] = Sk[i][j];
    double L[4][4], Rr[4][4];
    q_lr(qmul(qinv(Qj), Qi), +1.0, L); q_lr(cq, -1.0, Rr);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += L[1 + i][k] * Rr[k][1 + j];
        M[O_R + i][O_PR + j] = -s;
      }
    for (int i = 0; i < 3; ++i) t[i] = G[i] * sum_dt + Vj[i] - Vi[i];
    qrot(Qi_inv, t, o); skew(o, Sk);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[O_V + i][O_PR + j] = Sk[i][j];
    finish(M, 7, J[0]);
  }
  if (J[1]) {
    std::memset(M, 0, sizeof(M));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { M[O_T + i][j] = -Ri_inv[i][j] * sum_dt; M[O_V + i][j] = -Ri_inv[i][j]; }
    finish(M, 3, J[1]);
  }
  if (J[2]) {
    std::memset(M, 0, sizeof(M));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { M[O_T + i][j] = -dp_dba[i][j]; M[O_V + i][j] = -dv_dba[i][j]; M[O_BA + i][j] = (i == j) ? -1.0 : -0.0; }
    finish(M, 3, J[2]);
  }
  if (J[3]) {
    std::memset(M, 0, sizeof(M));
    double L[4][4];
    q_lr(qmul(qmul(qinv(Qj), Qi), P.dq), +1.0, L);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0; for (int k = 0; k < 3; ++k) s += L[1 + i][1 + k] * dq_dbg[k][j];
        M[O_R + i][j] = -s;
        M[O_T + i][j] = -dp_dbg[i][j]; M[O_V + i][j] = -dv_dbg[i][j]; M[O_BG + i][j] = (i == j) ? -1.0 : -0.0;
      }
    finish(M, 3, J[3]);
  }
  if (J[4]) {
    std::memset(M, 0, sizeof(M));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[O_T + i][O_PT + j] = Ri_inv[i][j];
    double L[4][4];
    q_lr(qmul(qmul(qinv(cq), Qi_inv), Qj), +1.0, L);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[O_R + i][O_PR + j] = L[1 + i][1 + j];
    finish(M, 7, J[4]);
  }
  if (J[5]) {
    std::memset(M, 0, sizeof(M));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[O_V + i][j] = Ri_inv[i][j];
    finish(M, 3, J[5]);
  }
  if (J[6]) {
    std::memset(M, 0, sizeof(M));
    for (int i = 0; i < 3; ++i) M[O_BA + i][i] = 1.0;
    finish(M, 3, J[6]);
  }
  if (J[7]) {
    std::memset(M, 0, sizeof(M));
    for (int i = 0; i < 3; ++i) M[O_BG + i][i] = 1.0;
    finish(M, 3, J[7]);
  }
}

}  // namespace imu
}  // namespace lvo
