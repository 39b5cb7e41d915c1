// prior_eval.hpp — one weak-constraint pose prior block, evaluated by one thread (PoseGraphError <6,7,7>, PoseError <6,7>, RError <4,7>:
// src/lvio_fusion/include/lvio_fusion/ceres/pose_error.hpp:10-110).  Shared by the materialising kernel (prior_kernels.hip) and the fused
// linearisation / cost kernels of the solver (solver_kernels.hip), which keep the block's outputs in the thread instead of a round trip
// through global memory and a second launch.
#pragma once
#include "lvf_internal.hpp"
#include "se3_jet.hpp"

namespace lvf {

// kf_a == -2: RError on pose kf_b with q0 = target[0..4); kf_a == -1: PoseError on pose kf_b with origin = target[0..7);
// kf_a >= 0: PoseGraphError between Twc1 = pose kf_a and Twc2 = pose kf_b with rpyxyz_ = target[0..6).
// Outputs of block i: res[6]; ja, jb [6][7] row-major (ja is all-zero for PoseError / RError blocks) — pointers to THIS block's storage.
template <bool WITH_J>
__device__ __forceinline__ void pose_prior_eval(const int a, const int b, const double* __restrict__ tg, const double w, const double v,
                                                const double* __restrict__ poses, double* __restrict__ res, double* __restrict__ ja, double* __restrict__ jb) {
  typedef DJet<14> J;
  if (a == -2) {
    // RError <4,7> (pose_error.hpp:88-110): r_k = w (q_k - q0_k) on the four quaternion components; rows 4,5 are padding
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      res[k] = k < 4 ? w * (poses[7 * b + k] - tg[k]) : 0.0;
      if (WITH_J) {
#pragma unroll
        for (int c = 0; c < 7; ++c) { ja[7 * k + c] = 0.0; jb[7 * k + c] = (k < 4 && c == k) ? w : 0.0; }
      }
    }
    return;
  }
  J rp[6];
  double scale[6], offs[6], sign;
  if (a >= 0) {
    J T1[7], T2[7], inv1[7], rel[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { T1[k] = J(poses[7 * a + k], k); T2[k] = J(poses[7 * b + k], 7 + k); }
    se3_inverse(T1, inv1);
    se3_product(inv1, T2, rel);
    se3_to_rpyxyz(rel, rp);
    scale[0] = scale[1] = scale[2] = v * w; scale[3] = w; scale[4] = scale[5] = 10 * w;
#pragma unroll
    for (int k = 0; k < 6; ++k) offs[k] = tg[k];
    sign = -1.0;   // r = scale * (target - rpyxyz)
  } else {
    J O[7], P[7], invo[7], rel[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { O[k] = J(tg[k]); P[k] = J(poses[7 * b + k], 7 + k); }
    se3_inverse(O, invo);
    se3_product(invo, P, rel);
    se3_to_rpyxyz(rel, rp);
    scale[0] = scale[1] = scale[2] = v * w; scale[3] = scale[4] = scale[5] = w;
#pragma unroll
    for (int k = 0; k < 6; ++k) offs[k] = 0.0;
    sign = 1.0;    // r = scale * rpyxyz
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    res[k] = (sign < 0.0) ? scale[k] * (offs[k] - rp[k].a) : scale[k] * rp[k].a;
    if (WITH_J) {
      const double s = sign * scale[k];
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        ja[7 * k + c] = s * rp[k].v[c];
        jb[7 * k + c] = s * rp[k].v[7 + c];
      }
    }
  }
}

}  // namespace lvf
