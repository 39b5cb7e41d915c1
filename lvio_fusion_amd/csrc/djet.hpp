// djet.hpp — forward-mode dual numbers for device code (gfx950).
//
// The hot functors (reprojection, point-to-plane, ImuError) carry hand-derived closed-form Jacobians (lvf_math.hpp).
// The COLD ones — the <=50 weak-constraint priors of a window (PoseGraphError / PoseError,
// src/lvio_fusion/include/lvio_fusion/ceres/pose_error.hpp:10-86) — go through atan2/asin of an un-normalised
// quaternion product; for those the reference's own definition "Jacobian = exact derivative of the code as written"
// (ceres::AutoDiffCostFunction) is reproduced literally with a dual number whose N partials live in registers.
// One thread evaluates one residual block; throughput is irrelevant here, exactness is the point.
#pragma once
#include <hip/hip_runtime.h>

namespace lvf {

template <int N>
struct DJet {
  double a;
  double v[N];
  __device__ __forceinline__ DJet() : a(0.0) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  __device__ __forceinline__ DJet(double x) : a(x) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  __device__ __forceinline__ DJet(double x, int k) : a(x) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (i == k) ? 1.0 : 0.0;
  }
};

template <int N>
__device__ __forceinline__ DJet<N> operator+(const DJet<N>& x, const DJet<N>& y) {
  DJet<N> r;
  r.a = x.a + y.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i];
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> operator-(const DJet<N>& x, const DJet<N>& y) {
  DJet<N> r;
  r.a = x.a - y.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i];
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> operator-(const DJet<N>& x) {
  DJet<N> r;
  r.a = -x.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = -x.v[i];
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> operator*(const DJet<N>& x, const DJet<N>& y) {
  DJet<N> r;
  r.a = x.a * y.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a;
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> operator/(const DJet<N>& x, const DJet<N>& y) {
  DJet<N> r;
  const double iy = 1.0 / y.a;
  r.a = x.a * iy;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * iy;
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> operator*(double s, const DJet<N>& y) {
  DJet<N> r;
  r.a = s * y.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = s * y.v[i];
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> jsqrt(const DJet<N>& x) {
  DJet<N> r;
  r.a = sqrt(x.a);
  const double d = 0.5 / r.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = d * x.v[i];
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> jasin(const DJet<N>& x) {
  DJet<N> r;
  r.a = asin(x.a);
  const double d = 1.0 / sqrt(1.0 - x.a * x.a);
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = d * x.v[i];
  return r;
}
template <int N>
__device__ __forceinline__ DJet<N> jatan2(const DJet<N>& y, const DJet<N>& x) {
  DJet<N> r;
  r.a = atan2(y.a, x.a);
  const double d = 1.0 / (x.a * x.a + y.a * y.a);
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = d * (x.a * y.v[i] - y.a * x.v[i]);
  return r;
}

}  // namespace lvf
