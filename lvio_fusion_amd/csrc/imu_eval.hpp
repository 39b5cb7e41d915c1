// imu_eval.hpp — the ImuError evaluation shared by the stand-alone kernels (imu_kernels.hip) and the solver's merged launches
// (solver_kernels.hip): constants of the flattened pre-integration, quaternion helpers, and the per-factor evaluation in phases
// (the caller places the workgroup barriers between them, so one wave per factor works inside any workgroup shape).
//
// Replaces ImuError::Evaluate include/lvio_fusion/ceres/imu_error.hpp:17-113 / Preintegration::Evaluate src/preintegration.cpp:144-165.
#pragma once
#include "lvf_internal.hpp"

namespace lvf {

constexpr int kPre = 467;     // doubles per flattened lvf_preint
constexpr int OFF_SUMDT = 0, OFF_LBA = 1, OFF_LBG = 4, OFF_DP = 7, OFF_DQ = 10, OFF_DV = 14, OFF_JAC = 17, OFF_COV = 242;
constexpr int O_T = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12, O_PR = 0, O_PT = 4;   // preintegration.cpp:12
__device__ constexpr double kG[3] = {0.0, 0.0, 9.81007};                            // preintegration.cpp:13

struct Qd { double x, y, z, w; };
__device__ __forceinline__ Qd qmul(const Qd& a, const Qd& b) {
  Qd r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
__device__ __forceinline__ Qd qinv(const Qd& q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  return Qd{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
}
__device__ __forceinline__ void qrot(const Qd& q, const double v[3], double o[3]) {
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  o[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  o[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
__device__ __forceinline__ void qmat(const Qd& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ Qd q_delta(const double th[3]) { return Qd{th[0] / 2.0, th[1] / 2.0, th[2] / 2.0, 1.0}; }
// bottom-right 3x3 of q_left(q) (sign=+1) / q_right(q) (sign=-1): w I +- skew(vec)   utility.h:124-140
__device__ __forceinline__ void q_lr_br(const Qd& q, double sign, double B[9]) {
  B[0] = q.w;           B[1] = -sign * q.z;  B[2] = sign * q.y;
  B[3] = sign * q.z;    B[4] = q.w;          B[5] = -sign * q.x;
  B[6] = -sign * q.y;   B[7] = sign * q.x;   B[8] = q.w;
}
__device__ __forceinline__ void blk3(const double* J15, int r, int c, double B[9]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = J15[15 * (r + i) + c + j];
}


// Phase A (all 64 lanes of the factor's wave): sqrt_info into LDS, the pre-weighting Jacobian cleared.  sS[225], sM[480].
template <bool WITH_J>
__device__ __forceinline__ void imu_stage(const int f, const int lane, const double* __restrict__ sqrt_info, double* sS, double* sM) {
  for (int k = lane; k < 225; k += 64) sS[k] = sqrt_info[(size_t)f * 225 + k];
  if (WITH_J) for (int k = lane; k < 480; k += 64) sM[k] = 0.0;
}

// Phase B (ONE lane): the raw residual sr0[15] and, WITH_J, the 15 x 32 pre-weighting Jacobian sM (columns: pose_i 0..6, v_i 7..9,
// ba_i 10..12, bg_i 13..15, pose_j 16..22, v_j 23..25, ba_j 26..28, bg_j 29..31).
// (imu_raw_at: the two keyframes' blocks by pointer — pose [7], v [3], ba [3], bg [3] each — wherever they live)
template <bool WITH_J>
__device__ __forceinline__ void imu_raw_at(const double* P /* the factor's flattened pre-integration: global or staged in LDS */,
                                           const double* pi, const double* pj, const double* Vi, const double* Vj,
                                           const double* Bai, const double* Baj, const double* Bgi, const double* Bgj, double* sr0, double* sM) {
    const Qd Qi{pi[0], pi[1], pi[2], pi[3]}, Qj{pj[0], pj[1], pj[2], pj[3]};
    const double* Pi = pi + 4; const double* Pj = pj + 4;
    const double T = P[OFF_SUMDT];
    const double* Jp = P + OFF_JAC;
    // the five 3x3 blocks of the pre-integration Jacobian are read where they are used (twice: here and in the Jacobian columns below)
    // instead of being held in 90 registers across the whole function: this code shares its kernel with the visual linearisation, and
    // the kernel's register allocation — hence its occupancy — is the maximum over everything inlined into it
    double dba[3], dbg[3];
    for (int k = 0; k < 3; ++k) { dba[k] = Bai[k] - P[OFF_LBA + k]; dbg[k] = Bgi[k] - P[OFF_LBG + k]; }
    double th[3], cv[3], cp[3];
    const Qd dq{P[OFF_DQ], P[OFF_DQ + 1], P[OFF_DQ + 2], P[OFF_DQ + 3]};
    {
      double blk[9], a3[3], b3[3];
      blk3(Jp, O_R, O_BG, blk); mat3_mul_vec(blk, dbg, th);
      blk3(Jp, O_V, O_BA, blk); mat3_mul_vec(blk, dba, a3); blk3(Jp, O_V, O_BG, blk); mat3_mul_vec(blk, dbg, b3);
      for (int k = 0; k < 3; ++k) cv[k] = P[OFF_DV + k] + a3[k] + b3[k];
      if (!WITH_J) asm volatile("" ::: "memory");           // residual-only callers stage P in LDS: one block at a time keeps the registers low
      blk3(Jp, O_T, O_BA, blk); mat3_mul_vec(blk, dba, a3); blk3(Jp, O_T, O_BG, blk); mat3_mul_vec(blk, dbg, b3);
      for (int k = 0; k < 3; ++k) cp[k] = P[OFF_DP + k] + a3[k] + b3[k];
    }
    asm volatile("" ::: "memory");                        // (the blocks are loaded again below, not carried)
    const Qd cq = qmul(dq, q_delta(th));
    const Qd Qi_inv = qinv(Qi);
    double tp[3], op[3], tv[3], ov[3];
    for (int k = 0; k < 3; ++k) tp[k] = 0.5 * kG[k] * T * T + Pj[k] - Pi[k] - Vi[k] * T;
    qrot(Qi_inv, tp, op);
    for (int k = 0; k < 3; ++k) tv[k] = kG[k] * T + Vj[k] - Vi[k];
    qrot(Qi_inv, tv, ov);
    const Qd e = qmul(qinv(cq), qmul(Qi_inv, Qj));
    for (int k = 0; k < 3; ++k) { sr0[O_T + k] = op[k] - cp[k]; sr0[O_V + k] = ov[k] - cv[k]; sr0[O_BA + k] = Baj[k] - Bai[k]; sr0[O_BG + k] = Bgj[k] - Bgi[k]; }
    sr0[O_R + 0] = 2 * e.x; sr0[O_R + 1] = 2 * e.y; sr0[O_R + 2] = 2 * e.z;
    if (WITH_J) {
      double Ri_inv[9];
      qmat(Qi_inv, Ri_inv);
#define MM(r, c) sM[(r) * 32 + (c)]
      // pose_i  (cols 0..6)
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) MM(O_T + a, O_PT + b) = -Ri_inv[3 * a + b];
      MM(O_T + 0, O_PR + 1) = -op[2]; MM(O_T + 0, O_PR + 2) = op[1];
      MM(O_T + 1, O_PR + 0) = op[2];  MM(O_T + 1, O_PR + 2) = -op[0];
      MM(O_T + 2, O_PR + 0) = -op[1]; MM(O_T + 2, O_PR + 1) = op[0];
      {
        // -(q_left(Qj^-1 Qi) q_right(cq)).bottomRightCorner<3,3>() of the 4x4 product:
        // corner = a_v (-b_v)^T + (a_w I + [a_v]x)(b_w I - [b_v]x)
        const Qd A = qmul(qinv(Qj), Qi);
        double LA[9], RB[9];
        q_lr_br(A, +1.0, LA); q_lr_br(cq, -1.0, RB);
        const double av[3] = {A.x, A.y, A.z}, bv[3] = {cq.x, cq.y, cq.z};
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            double s = av[a] * (-bv[b]);
            for (int k = 0; k < 3; ++k) s += LA[3 * a + k] * RB[3 * k + b];
            MM(O_R + a, O_PR + b) = -s;
          }
      }
      MM(O_V + 0, O_PR + 1) = -ov[2]; MM(O_V + 0, O_PR + 2) = ov[1];
      MM(O_V + 1, O_PR + 0) = ov[2];  MM(O_V + 1, O_PR + 2) = -ov[0];
      MM(O_V + 2, O_PR + 0) = -ov[1]; MM(O_V + 2, O_PR + 1) = ov[0];
      // v_i (cols 7..9)
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { MM(O_T + a, 7 + b) = -Ri_inv[3 * a + b] * T; MM(O_V + a, 7 + b) = -Ri_inv[3 * a + b]; }
      // ba_i (cols 10..12)
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { MM(O_T + a, 10 + b) = -Jp[15 * (O_T + a) + O_BA + b]; MM(O_V + a, 10 + b) = -Jp[15 * (O_V + a) + O_BA + b]; }
      for (int a = 0; a < 3; ++a) MM(O_BA + a, 10 + a) = -1.0;
      // bg_i (cols 13..15)
      {
        const Qd B = qmul(qmul(qinv(Qj), Qi), dq);
        double LB[9];
        q_lr_br(B, +1.0, LB);
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += LB[3 * a + k] * Jp[15 * (O_R + k) + O_BG + b];
            MM(O_R + a, 13 + b) = -s;
            MM(O_T + a, 13 + b) = -Jp[15 * (O_T + a) + O_BG + b];
            MM(O_V + a, 13 + b) = -Jp[15 * (O_V + a) + O_BG + b];
          }
        for (int a = 0; a < 3; ++a) MM(O_BG + a, 13 + a) = -1.0;
      }
      // pose_j (cols 16..22)
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) MM(O_T + a, 16 + O_PT + b) = Ri_inv[3 * a + b];
      {
        const Qd C = qmul(qmul(qinv(cq), Qi_inv), Qj);
        double LC[9];
        q_lr_br(C, +1.0, LC);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) MM(O_R + a, 16 + O_PR + b) = LC[3 * a + b];
      }
      // v_j (23..25), ba_j (26..28), bg_j (29..31)
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) MM(O_V + a, 23 + b) = Ri_inv[3 * a + b];
      for (int a = 0; a < 3; ++a) { MM(O_BA + a, 26 + a) = 1.0; MM(O_BG + a, 29 + a) = 1.0; }
#undef MM
    }
}


template <bool WITH_J>
__device__ __forceinline__ void imu_raw(const int f, const double* P, const int* __restrict__ kf_i, const int* __restrict__ kf_j,
                                        const double* __restrict__ poses, const double* __restrict__ vel, const double* __restrict__ ba,
                                        const double* __restrict__ bg, double* sr0, double* sM) {
  const int i = kf_i[f], j = kf_j[f];
  imu_raw_at<WITH_J>(P, poses + 7 * i, poses + 7 * j, vel + 3 * i, vel + 3 * j, ba + 3 * i, ba + 3 * j, bg + 3 * i, bg + 3 * j, sr0, sM);
}
// the two keyframes' states staged as rows of 16 doubles: [pose 7 | v 3 | ba 3 | bg 3], keyframe i then keyframe j
template <bool WITH_J>
__device__ __forceinline__ void imu_raw16(const double* P, const double* X, double* sr0, double* sM) {
  imu_raw_at<WITH_J>(P, X, X + 16, X + 7, X + 23, X + 10, X + 26, X + 13, X + 29, sr0, sM);
}

// Phase C: lane r < 15 returns row r of the weighted residual sqrt_info . r0 (other lanes 0)
__device__ __forceinline__ double imu_weighted_residual(const int lane, const double* sS, const double* sr0) {
  double s = 0.0;
  if (lane < 15) for (int k = 0; k < 15; ++k) s += sS[15 * lane + k] * sr0[k];
  return s;
}
// Phase D: entry e = 32 r + c of the weighted Jacobian sqrt_info . M
__device__ __forceinline__ double imu_weighted_jacobian(const int e, const double* sS, const double* sM) {
  const int r = e >> 5, c = e & 31;
  double s = 0.0;
  for (int k = 0; k < 15; ++k) s += sS[15 * r + k] * sM[32 * k + c];
  return s;
}

}  // namespace lvf
