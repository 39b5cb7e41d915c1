// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
// executed by the product path (lvio_fusion_amd/, include/).  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may use it, and only as the checker.
//
// PARITY: the functor restatement (jet.h, se3_ops.h, factors.h) is PINNED to the reference's own text — the reference's
// ceres/{base,visual_error,lidar_error,pose_error}.hpp are compiled unmodified into oracle/_ref (oracle/ref_driver.cpp, stand-in
// third-party headers in oracle/ref_shim/) and tests/test_oracle_ref.py requires bit-for-bit agreement, live and against the
// committed reference outputs tests/golden/ref_v1.npz.  STILL UNPINNED (the reference ships no tests or golden vectors and these
// translation units need real Eigen / PCL / Ceres, absent here): imu.h, knn.h, cloud.h, extract.h, and the DECLARED Ceres
// solver semantics of robust.h / lm.h / icp.h — those are cross-checked by independent derivations instead (mpmath finite
// differences, closed-form known answers, scipy cKDTree, and a numpy dense normal-equation solve: tests/test_oracle_lm_numpy.py).
//
// jet.h — forward-mode dual numbers, the published algorithm behind
// ceres::AutoDiffCostFunction (every reference functor is wrapped in one, e.g.
// src/lvio_fusion/include/lvio_fusion/ceres/visual_error.hpp:68,100,130).
// A Jet<N> carries a value and the N partial derivatives w.r.t. the concatenated
// parameter blocks; evaluating the templated functor on Jets yields the exact derivative
// of the code as written (ambient coordinates).
#pragma once
#include <cmath>

namespace lvo {

template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT implicit
  Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};

#define LVO_JJ template <int N> inline Jet<N>
LVO_JJ operator+(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
LVO_JJ operator-(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
LVO_JJ operator-(const Jet<N>& x) { Jet<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
LVO_JJ operator*(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
LVO_JJ operator/(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; const double inv = 1.0 / y.a; const double q = x.a * inv; r.a = q;
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - q * y.v[i]) * inv;
  return r;
}
LVO_JJ operator+(const Jet<N>& x, double s) { Jet<N> r = x; r.a += s; return r; }
LVO_JJ operator+(double s, const Jet<N>& x) { Jet<N> r = x; r.a += s; return r; }
LVO_JJ operator-(const Jet<N>& x, double s) { Jet<N> r = x; r.a -= s; return r; }
LVO_JJ operator-(double s, const Jet<N>& x) { Jet<N> r = -x; r.a += s; return r; }
LVO_JJ operator*(const Jet<N>& x, double s) { Jet<N> r; r.a = x.a * s; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
LVO_JJ operator*(double s, const Jet<N>& x) { return x * s; }
LVO_JJ operator/(const Jet<N>& x, double s) { return x * (1.0 / s); }
LVO_JJ operator/(double s, const Jet<N>& y) {
  Jet<N> r; const double inv = 1.0 / y.a; r.a = s * inv; const double m = -s * inv * inv;
  for (int i = 0; i < N; ++i) r.v[i] = m * y.v[i];
  return r;
}
template <int N> inline Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> inline Jet<N>& operator-=(Jet<N>& x, const Jet<N>& y) { x = x - y; return x; }
template <int N> inline Jet<N>& operator*=(Jet<N>& x, const Jet<N>& y) { x = x * y; return x; }

LVO_JJ sqrt(const Jet<N>& x) { Jet<N> r; r.a = std::sqrt(x.a); const double d = 1.0 / (2.0 * r.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
LVO_JJ sin(const Jet<N>& x) { Jet<N> r; r.a = std::sin(x.a); const double d = std::cos(x.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
LVO_JJ cos(const Jet<N>& x) { Jet<N> r; r.a = std::cos(x.a); const double d = -std::sin(x.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
LVO_JJ asin(const Jet<N>& x) { Jet<N> r; r.a = std::asin(x.a); const double d = 1.0 / std::sqrt(1.0 - x.a * x.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
LVO_JJ atan2(const Jet<N>& y, const Jet<N>& x) {
  Jet<N> r; r.a = std::atan2(y.a, x.a); const double d = 1.0 / (x.a * x.a + y.a * y.a);
  for (int i = 0; i < N; ++i) r.v[i] = (x.a * y.v[i] - y.a * x.v[i]) * d;
  return r;
}
#undef LVO_JJ

// scalar overloads so that templated functors can call sqrt/sin/... unqualified on double/float
using std::sqrt; using std::sin; using std::cos; using std::asin; using std::atan2;

}  // namespace lvo
